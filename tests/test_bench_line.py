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
    """scripts/rank_imbalance.py ran every rank r of N in turn on one GPU, at the 32-spp step the scale legs are quoted on and at the 8-spp
    short step; the expected speed-up of a leg is t(1) / max_r t(r of N), it grows with N and stays below N, the imbalance it reports is
    max / mean, and the sweep was made with the tile size the leg deals (bench.WORKLOADS[leg]["tile"])."""
    f = bench._newest("rank_imbalance_32spp.json")
    assert f, "profiles/rN/rank_imbalance_32spp.json is missing"
    doc = json.load(open(f))
    for leg in bench.SCALE_LEGS:
        last = 1.0
        assert doc[leg].get("tile", bench.TILE) == bench.WORKLOADS[leg].get("tile", bench.TILE), leg
        for world in (2, 4, 8):
            e = bench.expected_speedup_leg(leg, world)
            assert e and e["source"].startswith("profiles/") and e["spp"] == bench.SCALE_SPP
            assert 0.6 * world < e["vs_1gpu"] < world * 1.02
            assert e["vs_1gpu"] > last
            last = e["vs_1gpu"]
            rows = doc[leg]["world%d" % world]
            assert len(rows["per_rank_ms"]) == world
            assert abs(e["max_over_mean_rank_ms"] - max(rows["per_rank_ms"]) / (sum(rows["per_rank_ms"]) / world)) < 2e-3
            short = e["at_short_steps"]
            assert short and short["spp"] == bench.SCALE_SHORT_SPP and 0.45 * world < short["vs_1gpu"] <= e["vs_1gpu"] * 1.05
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
        # `bound` names the saturated pipe (committed counters: vector-memory issue where the texture addresser is busier than the
        # fabric is full); the contract's units stay: bytes / 8 TB/s
        assert r["bound"] in ("hbm", "vmem_issue") and r["contract_bound"] == "hbm" and r["peak"] == bench.HBM_PEAK_GBS and r["unit"] == "GB/s"
        if r["bound"] == "vmem_issue":
            assert r["pipes"]["ta_busy"] > r["traffic_frac_of_peak"]
        assert "vmem_spill_stale" in r or r["vmem_spill_share"] is None
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
    # the parity gate in the line itself: the headline's frame against the oracle over the whole frame, every
    # leg's on two 8-row strips at the leg's own sample count -- and a value only where the gate passed
    par = out["parity"]
    assert par["against"] == "oracle/tb_oracle.cpp" and par["ok"] is True and par["bit_equal"] is True and par["rel_l2"] == 0.0
    assert par["pixels"] == 1920 * 1080 and par["frames"] >= 1 and par["tolerance"] == 1e-4
    if not par["timed_frame"]:      # the 1-s CPU sample of this test covers a few of the 64 frames: the timed frame itself is gated on strips
        assert par["timed_frame_strips"]["bit_equal"] is True and par["timed_frame_strips"]["frames"] == 64
    for leg in ("teapot", "vwvan"):
        lp = out["roofline_" + leg]["parity"]
        assert lp["ok"] is True and lp["bit_equal"] is True and lp["pixels"] >= 2 * 8 * 1920 and lp["frames"] == bench.WORKLOADS[leg]["spp"]


def test_parity_gate_blocks_a_value_whose_frame_is_wrong():
    """bench.gate: bit-equal and within-tolerance frames keep their value; a frame beyond 1e-4 relative L2 (whole block or any pixel),
    or with a NaN the oracle does not have, prints value: null and keeps the measured number as value_unverified."""
    import numpy as np
    rng = np.random.default_rng(0)
    cpu = rng.uniform(0.5, 2.0, (8, 64, 4)).astype(np.float32)
    same = bench.parity_compare(np, cpu.copy(), cpu)
    assert same["bit_equal"] and same["rel_l2"] == 0.0 and bench.parity_ok(same)
    close = cpu.copy(); close[3, 5, 1] = np.nextafter(close[3, 5, 1], np.float32(9))
    c = bench.parity_compare(np, close, cpu)
    assert not c["bit_equal"] and c["differing_pixels"] == 1 and 0 < c["max_pixel_rel_l2"] < 1e-6 and bench.parity_ok(c)
    far = cpu.copy(); far[0, 0, :3] *= np.float32(1.01)          # one pixel 1 % off: the whole-block L2 would pass, the per-pixel gate does not
    f = bench.parity_compare(np, far, cpu)
    assert f["rel_l2"] < 1e-3 and f["max_pixel_rel_l2"] > 1e-3 and not bench.parity_ok(f)
    nan = cpu.copy(); nan[1, 1, 0] = np.nan
    assert not bench.parity_ok(bench.parity_compare(np, nan, cpu))
    r = bench.gate(dict(f, ok=bench.parity_ok(f)), {"value": 123.0})
    assert r["value"] is None and r["value_unverified"] == 123.0 and r["parity"]["ok"] is False
    r = bench.gate(dict(same, ok=True), {"value": 123.0})
    assert r["value"] == 123.0 and "value_unverified" not in r
    assert bench.strip_rows(1080) == [360, 720] and bench.strip_rows(2160) == [720, 1440] and bench.strip_rows(5) == [0, 0]
