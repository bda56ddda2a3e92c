"""-m gpu: the split-role kernel (option pipeline = 4, pt_split.inc: shading waves + traversal waves over an LDS ray queue) against
the CPU oracle and against the lock-step kernel.  Rays leave the lane that shades their path and are walked by whichever traversal
lane is free; per-ray arithmetic, visit order, rand() order and the order of additions to a sample are unchanged: BIT-EXACT."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import CORNELL, GOLDEN

pytestmark = pytest.mark.gpu
TEAPOT = os.path.join(GOLDEN, "scenes", "Teapot", "scene.pbrt")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture()
def split_tb(gpu_tb):
    keys = ("pipeline", "split_trav", "split_shade", "split_ready", "split_refill", "split_frame_group", "split_stack_cap", "overlap_launches")
    gpu_tb.SetOption("pipeline", 4)
    yield gpu_tb
    gpu_tb.SetOption("pipeline", 0)
    for k, v in (("split_trav", 4), ("split_ready", 32), ("split_refill", 16), ("split_frame_group", 8), ("split_stack_cap", 0), ("overlap_launches", 1)):
        gpu_tb.SetOption(k, v)
    gpu_tb.SetOption("split_shade", 0)  # 0 = the default for the scene kind


def _oracle(tb, W, H, frames, s, **kw):
    return ol.render(tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, 0.0), W, H, frames, threads=8, **kw)


def _check(tb, W, H, F, s, jittered=True):
    tb.InvalidateHistory()
    tb.Render(W, H, F, s, 0.0)
    assert tb.GetOption("last_pipeline") == 4
    out, jit = tb.ReadAccumulation(jittered=True)
    ref = _oracle(tb, W, H, F, s, jittered=True)
    assert np.array_equal(bits(out), bits(ref["output"]))
    assert np.array_equal(bits(jit), bits(ref["jittered"]))


def test_split_cornell_bit_exact(split_tb, settings):
    split_tb.LoadScene(CORNELL)
    _check(split_tb, 160, 96, 4, settings)


@pytest.mark.parametrize("wh", [(1, 1), (7, 5), (17, 9), (200, 120), (333, 77)])
def test_split_ragged_frames_bit_exact(split_tb, settings, wh):
    split_tb.LoadScene(CORNELL)
    _check(split_tb, wh[0], wh[1], 3, settings)


def test_split_progressive_calls_and_depths(split_tb, settings):
    import copy
    split_tb.LoadScene(CORNELL)
    for depth in (0, 1, 2, 8):
        s = copy.copy(settings); s.MaxBounces = depth
        split_tb.InvalidateHistory()
        split_tb.Render(96, 64, 2, s, 0.0); split_tb.Render(96, 64, 3, s, 0.0)
        assert split_tb.GetOption("last_pipeline") == 4
        out = split_tb.ReadAccumulation()
        ref = _oracle(split_tb, 96, 64, 5, s)["output"]
        assert np.array_equal(bits(out), bits(ref)), depth


@pytest.mark.parametrize("shape", [(1, 1), (2, 6), (4, 4), (8, 8), (3, 13)])
def test_split_workgroup_shapes(split_tb, settings, shape):
    split_tb.LoadScene(CORNELL)
    split_tb.SetOption("split_trav", shape[0]); split_tb.SetOption("split_shade", shape[1])
    _check(split_tb, 160, 96, 4, settings)
    assert split_tb.GetOption("last_split_waves") == shape[0] * 100 + shape[1]


@pytest.mark.parametrize("ready,refill,fg", [(1, 1, 1), (64, 64, 3), (16, 48, 64)])
def test_split_thresholds(split_tb, settings, ready, refill, fg):
    split_tb.LoadScene(CORNELL)
    split_tb.SetOption("split_ready", ready); split_tb.SetOption("split_refill", refill); split_tb.SetOption("split_frame_group", fg)
    _check(split_tb, 120, 72, 5, settings)


def test_split_proc_scene_global_memory_and_split_stack(split_tb, settings):
    split_tb.LoadProcedural(0, 60000, 1234)
    _check(split_tb, 128, 80, 3, settings)
    split_tb.SetOption("split_stack_cap", 6)
    _check(split_tb, 128, 80, 3, settings)


def test_split_teapot_env_textures(split_tb, settings):
    split_tb.LoadScene(TEAPOT)
    _check(split_tb, 96, 64, 2, settings)


@pytest.mark.parametrize("kind", [1, 2])
def test_split_glass_scenes(split_tb, settings, kind):
    import copy
    split_tb.LoadProcedural(kind, 30000, 1234)
    s = copy.copy(settings); s.MaxBounces = 6
    _check(split_tb, 96, 64, 2, s)


def test_split_equals_lockstep_at_1080p(split_tb, settings):
    import copy
    split_tb.LoadScene(CORNELL)
    s = copy.copy(settings); s.MaxBounces = 8
    split_tb.Render(1920, 1080, 4, s, 0.0)
    assert split_tb.GetOption("last_pipeline") == 4
    a = split_tb.ReadAccumulation()
    split_tb.SetOption("pipeline", 0)
    split_tb.InvalidateHistory()
    split_tb.Render(1920, 1080, 4, s, 0.0)
    assert split_tb.GetOption("last_pipeline") == 0
    b = split_tb.ReadAccumulation()
    assert np.array_equal(bits(a), bits(b))
