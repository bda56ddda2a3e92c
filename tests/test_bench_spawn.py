"""`python bench.py --gpus N` as typed (no launcher): the parent starts the N ranks itself before touching torch / the GPU, relays
rank 0's single JSON line and exits with the launcher's code.  Here without a GPU: --selftest-cpu makes the ranks rendezvous over
gloo, gather their packed tiles of a synthetic frame to rank 0 and un-permute them (the N > 1 data path minus the render)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def run_bench(*extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True, timeout=600, env=env)


def test_self_spawn_two_ranks_relays_one_json_line(built):
    r = run_bench("--gpus", "2", "--selftest-cpu")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["assembled_ok"] is True and out["selftest"] == "cpu-gloo"


def test_single_rank_selftest_and_launcher_mismatch(built):
    r = run_bench("--gpus", "1", "--selftest-cpu")
    assert r.returncode == 0 and json.loads(r.stdout.strip())["n_gpus"] == 1
    # a launcher that started a different number of ranks than --gpus says is an error, not a silent 1-GPU run
    r = run_bench("--gpus", "4", "--selftest-cpu", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "launcher started 1 ranks" in (r.stderr + r.stdout)


def test_without_gpu_the_bench_refuses_instead_of_falling_back(built):
    import torch
    if torch.cuda.is_available():
        return
    r = run_bench("--steps", "1")
    assert r.returncode != 0 and "HIP-only" in (r.stderr + r.stdout)


def test_cli_ranks_return_the_failing_ranks_status_and_leave_nothing_behind(built):
    """tracerboy-hip --ranks N (the C++ host's own multi-GPU launcher): a rank that cannot get its device ends the others, the tool
    returns THAT rank's exit status (not the 128 + SIGTERM of the ranks it stopped), and the private rendezvous directory of the RCCL
    unique id (mkdtemp, mode 0700) is removed.  Runs the same way with no GPU (every rank fails) and with one (rank 1 has no device)."""
    import glob
    from conftest import CORNELL, ROOT
    exe = os.path.join(ROOT, "tracerboy_amd", "tracerboy-hip")
    before = set(glob.glob("/tmp/tracerboy-hip-rccl-*"))
    r = subprocess.run([exe, CORNELL, "--ranks", "2", "--spp", "1", "--width", "64", "--height", "48", "--out", "/tmp/tb_never_written.pfm"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 1, (r.returncode, r.stderr[-2000:])
    assert "tb_create failed" in r.stderr
    assert set(glob.glob("/tmp/tracerboy-hip-rccl-*")) == before
    # rank variables set by hand, incompletely: refused before anything is rendered
    env = dict(os.environ, TB_CLI_RANK="1")
    r = subprocess.run([exe, CORNELL], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 2 and "TB_CLI_WORLD" in r.stderr


def test_expected_speedup_reads_the_committed_stand_in_measurements():
    """bench.py's N > 1 line quotes what the tile split should give from the one-GPU stand-in runs under profiles/ (the driver computes the
    measured ratio itself): both forms are there for configs[1] at 2 / 4 / 8 ranks, grow with N, stay below N; other workloads get none."""
    sys.path.insert(0, ROOT)
    import bench
    last = 1.0
    for world in (2, 4, 8):
        e = bench.expected_speedup("cornell-box", 1920, 1080, 64, 8, world)
        assert e and e["source"].startswith("profiles/") and e["pipelined"]["source"].startswith("profiles/")
        for v in (e["vs_1gpu"], e["pipelined"]["vs_1gpu"]):
            assert 0.7 * world < v < world
        assert e["pipelined"]["vs_1gpu"] > last; last = e["pipelined"]["vs_1gpu"]
    assert bench.expected_speedup("proc0:870000", 1920, 1080, 128, 6, 8) is None
