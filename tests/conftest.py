import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CORNELL = os.path.join(GOLDEN, "scenes", "cornell-box", "scene.pbrt")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) the HIP library and the oracle; both are required by every test."""
    from tracerboy_amd import build as tb_build
    tb_build.build(verbose=False)
    import oracle_lib
    oracle_lib.build()
    return True


@pytest.fixture(scope="session")
def cornell_host(built):
    from tracerboy_amd import api
    return api.HostScene(CORNELL)


@pytest.fixture(scope="session")
def settings(built):
    from tracerboy_amd import api
    s = api.GetDefaultOutputSettings()
    s.EnableBlueNoise = 0
    s.MaxBounces = 4
    return s


@pytest.fixture(scope="session")
def gpu_tb(built):
    from tracerboy_amd import api
    tb = api.TracerBoy(0)  # raises without a GPU: -m gpu tests must not silently fall back
    yield tb
    tb.close()
