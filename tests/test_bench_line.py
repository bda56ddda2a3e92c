"""bench.py's JSON line: the contract's fields and the round-5 rules about them.

CPU half: one BVH builder per workload at every N, `expected_speedup_leg` reads the committed per-rank
measurements, the N > 1 record copies -- never measures -- the CPU baseline.
GPU half: a short N = 1 run asserts that `frac` of every roofline block is achieved / peak of WORK (the SURVEY 8d
algorithmic byte rate over 8 TB/s; not a pipe's busy counter), with the pipe figures beside it and marked as committed."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_one_builder_per_workload_at_every_n():
    for scene in ("cornell-box", "proc0:870000", "proc1:700000", "proc2:2980000", bench.TEAPOT):
        picks = {bench.parse_args(["--gpus", str(n), "--scene", scene]).builder for n in (1, 2, 4, 8)}
        assert len(picks) == 1, (scene, picks)
    # the legs the driver's N = 1 and N > 1 runs share use the same tree
    for leg in bench.SCALE_LEGS:
        assert leg in bench.EXTRA_LEGS and "builder" in bench.WORKLOADS[leg]
    assert bench.parse_args(["--gpus", "1"]).legs.split(",") == list(bench.EXTRA_LEGS)
    assert bench.parse_args(["--gpus", "8"]).legs.split(",") == list(bench.SCALE_LEGS)


def test_expected_speedup_of_the_4k_legs_reads_the_committed_rank_sweep():
    """scripts/rank_imbalance.py ran every rank r of N in turn on one GPU; the expected speed-up of a leg is
    t(1) / max_r t(r of N), it grows with N and stays below N, and the imbalance it reports is max / mean."""
    f = bench._newest("rank_imbalance.json")
    assert f, "profiles/rN/rank_imbalance.json is missing"
    doc = json.load(open(f))
    for leg in bench.SCALE_LEGS:
        last = 1.0
        for world in (2, 4, 8):
            e = bench.expected_speedup_leg(leg, world)
            assert e and e["source"].startswith("profiles/")
            assert 0.45 * world < e["vs_1gpu"] < world * 1.02   # (vw-van at 8 spp per step and N = 8: 3.97x -- a 3.6-ms body under a launch's fixed 1.5 ms)
            assert e["vs_1gpu"] > last
            last = e["vs_1gpu"]
            rows = doc[leg]["world%d" % world]
            assert len(rows["per_rank_ms"]) == world
            assert abs(e["max_over_mean_rank_ms"] - max(rows["per_rank_ms"]) / (sum(rows["per_rank_ms"]) / world)) < 2e-3
    assert bench.expected_speedup_leg("teapot", 8) is None


def test_the_cpu_baseline_of_an_n_gt_1_record_is_a_marked_copy():
    cb = bench.cpu_baseline_copied()
    assert cb["measured"] is False
    if "copied_from" in cb:
        assert cb["copied_from"].startswith("profiles/") and cb["value"] > 0


def test_data_field_names_what_was_rendered():
    assert "cornell-box/scene.pbrt" in bench.data_label("cornell-box") and "synthetic" not in bench.data_label("cornell-box")
    assert bench.data_label("proc1:700000").startswith("synthetic")


def _check_roofline_block(r, hbm=True):
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "pipes", "useful_issue_frac", "vmem_spill_share"):
        assert k in r or k in ("useful_issue_frac",), (k, sorted(r))
    if hbm:
        assert r["bound"] == "hbm" and r["peak"] == bench.HBM_PEAK_GBS and r["unit"] == "GB/s"
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3            # a fraction of work, not a busy counter
        assert abs(r["achieved"] - r["algorithmic_bytes_per_sample"] * r["value"] * 1e6 / 1e9) / r["achieved"] < 0.25
    if r["pipes"]:
        assert "committed PMC" in r["pipes"]["source"]
        if "busiest_pipe" in r:
            assert r["busiest_pipe"]["source"] == "committed PMC" and 0 < r["busiest_pipe"]["busy"] <= 1.2


@pytest.mark.gpu
def test_n1_line_fields(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--legs", "teapot,vwvan",
                        "--leg-steps", "1", "--cpu-baseline-seconds", "1"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["unit"] == "Msamples/s" and out["dtype"] == "f32" and out["vs_baseline"] is None
    assert out["config"]["workload"] == "cornell-box 1920x1080 64spp depth8" and "synthetic" not in out["data"]
    assert out["roofline"]["bound"] == "valu" and out["roofline"]["avg_launch_ms"] > 0
    _check_roofline_block(out["roofline"], hbm=False)
    for leg in ("teapot", "vwvan"):
        blk = out["roofline_" + leg]
        assert blk["value"] > 0 and blk["avg_launch_ms"] > 0
        _check_roofline_block(blk)
    assert out["roofline_vwvan"]["kernel_variant"] == "vol" and out["roofline_vwvan"]["triangles"] > 600000
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] >= 1 and "measured" not in cb
    assert out["pcie_inclusive"]["value"] < out["value"] * 1.05
