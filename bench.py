#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: Msamples/s (W x H x spp / s) on BASELINE.json configs[1]:
Scenes/cornell-box, 1920x1080, 64 spp, depth 8, persistent-thread HIP on MI355X.

  python bench.py --gpus N --steps K --warmup W      (N > 1 without a launcher: starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full render of the workload (W*H*spp samples).  With N > 1 the frame is cut into 64x64 tiles dealt
round-robin to the ranks (tile t -> rank t % N); each rank renders its tiles, packs them, the packed HDR
buffers are gathered to rank 0 over RCCL and rank 0 un-permutes them into the full frame in HBM -- all inside the timed
region.  The total work is fixed, so scaling is "strong".  Scene + BVH are resident in HBM before the timed region
starts; nothing is read from the host inside it.

What one JSON line carries:
  "roofline"          the timed kernel on the timed workload.  cornell-box is LDS-resident, so the resource is VALU
                      issue: frac = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x launch time x 2.4 GHz); the instruction
                      count per launch comes from the committed rocprofv3 PMC pass of this command
                      (profiles/rN/c2_pmc_summary.json), the launch time is measured live with HIP events.
  "roofline_<leg>"    N = 1 only: further workloads rendered in the same invocation (WORKLOADS below): configs[2] at its
                      full 128 spp, the 4K scenes the 8 GPUs divide (stand-ins and the reference's own vw-van, flattened
                      and two-level) and the reference's Teapot.  `frac` is the SURVEY 8d algorithmic byte rate / 8 TB/s.
  "scale_<leg>"       N > 1 only: the 8-GPU configurations (van-class, bistro-class, vw-van at 3840x2160) through the
                      same tile-split step, quoted on a 32-spp step (the configurations are 256 / 1024 spp) with the 8-spp step
                      beside it (at_short_steps), each with its own scale_breakdown and expected_speedup.
  "cpu_baseline"      N = 1 only: the scalar oracle on the host cores, bounded sample of the timed workload.
  "parity"            the gate of BASELINE.md section 2 (CPU image vs HIP image before any timing is accepted): the frame the LAST TIMED
                      STEP left in HBM against the oracle's -- the whole frame for the headline at N = 1 (the image cpu_baseline renders
                      anyway), two 8-row strips at the workload's own sample count for every roofline_<leg> / scale_<leg> and for the
                      assembled frame at N > 1.  bit_equal, rel_l2 (block) and max_pixel_rel_l2; a figure whose frame is beyond 1e-4
                      prints value: null (the number stays as value_unverified).
"""
import argparse
import glob
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
CORNELL = os.path.join(SCENES, "cornell-box", "scene.pbrt")
TEAPOT = os.path.join(SCENES, "Teapot", "scene.pbrt")
VWVAN = os.path.join(SCENES, "vw-van", "vw-van.pbrt")
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
# 256 CUs x 4 SIMDs; a wave64 VALU instruction issues over 2 cycles (guide + scripts/microbench/valu_issue.hip)
SIMDS, CLOCK_GHZ, VALU_CYCLES = 1024, 2.4, 2
VALU_PEAK_GINST = SIMDS * CLOCK_GHZ / VALU_CYCLES   # 1228.8 G wave-instructions / s
TILE = 64
BUILDERS = ("lbvh", "sah", "lbvh-gpu", "lbvh+treelets", "lbvh+treelets-gpu")
VARIANTS = ("matte", "env", "surf", "vol", "full", "sss")
PIPELINES = ("lockstep", "stream", "wavefront", "pooled", "split")

# key -> the workload.  ONE bvh builder per workload at every N (the driver divides its per-N values, so the
# tree must not change with N).  "opts" are tb_set_option()s applied before the scene is loaded.
# Round 5 (scripts/vwvan_builders.py, profiles/r5/builders.json): the host's SAH build with three reinsertion passes over the largest 3 %
# of the subtrees (options reinsertion_passes / reinsertion_share; one such pass for the 2.98 M-triangle scene, none for Teapot, whose
# tree is better without) loads in 1-3.5 s and beats the GPU-built LBVH + treelets the legs used before by 4-19 %.
SAH3 = {"reinsertion_passes": 3, "reinsertion_share": 3}   # three reinsertion passes over the largest 3 % of the subtrees
WORKLOADS = {
    "c2": dict(scene="cornell-box", builder=1, W=1920, H=1080, spp=64, depth=8),       # BASELINE configs[1], the headline
    # configs[2] class: 870 k triangles
    "c3": dict(scene="proc0:870000", builder=1, W=1920, H=1080, spp=128, depth=6, opts=dict(SAH3)),
    # configs[3] class: glass, 4K, 8 of 256 spp
    "c4": dict(scene="proc1:700000", builder=1, W=3840, H=2160, spp=8, depth=6, opts=dict(SAH3)),
    # configs[4] class: 2.98 M tris, 40 materials
    "c5": dict(scene="proc2:2980000", builder=1, W=3840, H=2160, spp=8, depth=16, opts={"reinsertion_passes": 1, "reinsertion_share": 3}),
    # the reference's Teapot: textures + env + GGX
    "teapot": dict(scene=TEAPOT, builder=1, W=1920, H=1080, spp=16, depth=8, opts={"reinsertion_passes": 0}),
    # the reference's own configs[3] scene (Scenes/vw-van minus the absent body shell, tests/golden/make_vw_van_fixture.py)
    # (tile: the N > 1 deal of this workload -- a car in the middle of a sky balances better on 32x32 tiles: 6.14x against 5.93x at N = 8, profiles/r5/rank_tiles_32spp.json)
    "vwvan": dict(scene=VWVAN, builder=1, W=3840, H=2160, spp=8, depth=6, opts=dict(SAH3, flatten_instances=1), tile=32),
    "vwvan_2level": dict(scene=VWVAN, builder=1, W=3840, H=2160, spp=8, depth=6, opts=dict(SAH3, flatten_instances=0)),
}


def builder_label(w):
    """config.bvh_builder of a workload: the builder and, for the SAH build, its reinsertion passes when the workload sets them"""
    passes = (w.get("opts") or {}).get("reinsertion_passes")
    if passes is None or w["builder"] != 1:
        return BUILDERS[w["builder"]]
    share = (w.get("opts") or {}).get("reinsertion_share", 100)
    return "%s+%d reinsertion pass%s%s" % (BUILDERS[w["builder"]], passes, "" if passes == 1 else "es", "" if share >= 100 else " over %d%%" % share)


EXTRA_LEGS = ("c3", "c4", "c5", "teapot", "vwvan", "vwvan_2level")    # N = 1: roofline_<leg>
SCALE_LEGS = ("c4", "c5", "vwvan")                                    # N > 1: scale_<leg>
# N > 1: samples per pixel of a scale leg's timed step.  The configurations are 256 / 1024 spp; a step of 32 measures them (a launch costs a
# fixed 0.8-2.7 ms more than its samples on a rank's eighth of a 4K frame -- scripts/async_rate.py, docs/experiments/r6.md -- which is a third
# of an 8-spp step and a twentieth of a 32-spp one); the 8-spp step of the N = 1 legs is timed beside it as `at_short_steps`.
SCALE_SPP, SCALE_SHORT_SPP = 32, 8


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a burst long enough to be mostly steady state (the first and the last launch of a burst have no partner in flight: 5 steps read
    # 6 990-7 020 Msamples/s where 20 give 7 130; 0.4 s of GPU time either way)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--scene", default="cornell-box")  # or proc0:<tris> / proc1:<tris> / proc2:<tris> / path.pbrt
    # 0 LBVH, 1 binned SAH + reinsertion, 2 LBVH on the GPU, 3 LBVH + treelet passes (the reference's tree), 4 the same on
    # the GPU.  Default: 1 for scene files, 4 for procedural scenes -- at every N (the 58-s SAH build of a 3 M-triangle
    # scene by eight ranks on a 16-CPU quota would dominate an N = 8 run; the timed region never sees the build)
    ap.add_argument("--builder", type=int, default=None)
    ap.add_argument("--pipeline", type=int, default=0)  # 0 lock-step bounce (fastest measured), 1 stream, 2 wavefront, 3 pooled, 4 split
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT")  # extra tb_set_option()s, before the scene is loaded
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")     # profiling runs only: skip the oracle comparison of the timed frames (the line then carries no parity block)
    ap.add_argument("--no-readback", action="store_true")   # skip the PCIe-inclusive side measurement
    ap.add_argument("--no-legs", "--no-c3", dest="no_legs", action="store_true")   # skip roofline_<leg> / scale_<leg>
    ap.add_argument("--legs", default=None)                 # comma list; default: every leg of this N
    # steps of a leg: 5, so that a burst is mostly steady state (the first and last launch of a burst have no partner in
    # flight; at 3 steps vw-van's leg read 1 817 Msamples/s where twenty steps give 1 950)
    ap.add_argument("--leg-steps", type=int, default=8)
    ap.add_argument("--async-steps", action="store_true")   # N = 1: run the N > 1 step pipeline (async render + pack + consumer)
    ap.add_argument("--sync-steps", action="store_true")    # N = 1: wait for every render before enqueuing the next
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0)
    ap.add_argument("--selftest-cpu", action="store_true")  # plumbing test without a GPU (tests/test_bench_spawn.py)
    a = ap.parse_args(argv)
    if a.builder is None:
        a.builder = 4 if a.scene.startswith("proc") else 1
    if a.legs is None:
        a.legs = ",".join(EXTRA_LEGS if a.gpus == 1 else SCALE_LEGS)
    return a


# ------------------------------------------------------------------------------------------------- self-spawn
def self_spawn(args):
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as fresh processes through
    torch.distributed.run BEFORE this process touches torch or the GPU, relay rank 0's JSON line, exit with the
    launcher's code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    elif p.returncode == 0:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 1
    return p.returncode


def selftest_cpu(args, rank, world):
    """No GPU: the ranks rendezvous over gloo, every rank packs its tiles of a synthetic frame (numpy restatement of the
    pack kernel), ONE gather moves them to rank 0, rank 0 un-permutes (tb_unpack_gathered_host) and checks the frame.
    Exercises the launch / relay / collective plumbing of the N > 1 path; measures nothing."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from tracerboy_amd import tiles
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 200, 136
    full = (np.arange(W * H * 4, dtype=np.float32).reshape(H, W, 4) * np.float32(0.25))
    packed = torch.from_numpy(tiles.pack_owned_reference(full, rank, world, TILE, TILE))
    gathered = tiles.gather_to_rank0(packed, rank, world)
    ok = True
    if rank == 0:
        ok = bool(np.array_equal(tiles.assemble(W, H, world, TILE, TILE, gathered), full))
        print(json.dumps({"metric": "Msamples/s (WxHxspp/s)", "value": None, "unit": "Msamples/s", "n_gpus": world,
                          "selftest": "cpu-gloo", "assembled_ok": ok}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


# ------------------------------------------------------------------------------------------------- committed counters
def _newest(pattern):
    """profiles/rN/<pattern> of the highest N, or None."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", pattern)),
                   key=lambda f: int(re.search(r"profiles/r(\d+)", f).group(1)))
    return files[-1] if files else None


def byte_model(st):
    """Algorithmic bytes of DESIGN.md section 'Byte model' (layout-A accounting of SURVEY.md 8d, with the reference's
    real 72-B hit-group record): traversal + attribute + material + light + accumulation."""
    return (32 * st.boxesTested + 48 * st.trianglesTested + 180 * st.hitsShaded + 84 * st.materialFetches
            + 104 * st.lightSamples + 32 * st.samples)


_TIMED_KERNEL = re.compile(r"pt_persistent<\d+u, (true|false), (true|false), (true|false)")   # <F, LDS, COUNT, GROUPS, ...


def pmc_summary(key):
    """Counters per launch of the timed path-tracing kernel from the newest committed rocprofv3 PMC passes of workload
    `key` (profiles/rN/<key>_pmc_summary.json; separate --pmc runs, scripts/profile_bench.sh).  Returns (dict of
    pass -> counters, file) or (None, None).  The timed kernel is the frame-group form (GROUPS = true) without counters
    (COUNT = false); when the primary-visibility pre-pass (pt_primary) fed it, a "launch" is the pair and its counters
    are the two kernels' sums (GRBM_GUI_ACTIVE too: they run one after the other)."""
    f = _newest(key + "_pmc_summary.json")
    if not f:
        return None, None
    doc = json.load(open(f))
    from tracerboy_amd import build as tb_build
    stale = doc.pop("_kernel_digest", None) != tb_build.kernel_digest()   # counters of other device code
    primary = [v for k, v in doc.items() if "pt_primary<" in k]
    best = None
    for name, passes in doc.items():
        m = _TIMED_KERNEL.search(name)
        if not m or m.group(2) == "true":
            continue
        passes["_stale"] = stale
        if m.group(3) == "true":
            if primary:
                passes["_with_prepass"] = True
                for tag, counters in primary[0].items():
                    if isinstance(counters, dict) and isinstance(passes.get(tag), dict):
                        for c, val in counters.items():
                            if c != "dispatches" and isinstance(val, (int, float)):
                                passes[tag][c] = passes[tag].get(c, 0) + val
            return passes, os.path.relpath(f, ROOT)
        best = passes if best is None else best
    return best, (os.path.relpath(f, ROOT) if best else None)


def derived_busy(key, passes):
    """Pipe utilisations of the timed kernel from the same committed PMC passes, with rocprof's own definitions:
    VALUBusy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * cycles), cycles = GRBM_GUI_ACTIVE / 8 XCDs; TA busy =
    TA_TA_BUSY_sum / (256 TAs * cycles) from profiles/rN/<key>_mem_counters.json (scripts/pmc_mem.sh).  These are static
    numbers read from committed files, not measured in this run: the block says so ("source")."""
    out = {}
    if passes and "sq" in passes and "lds" in passes and passes["lds"].get("GRBM_GUI_ACTIVE"):
        cyc = passes["lds"]["GRBM_GUI_ACTIVE"] / 8.0
        out["valu_busy"] = round(4.0 * passes["sq"]["SQ_ACTIVE_INST_VALU"] / (SIMDS * cyc), 3)
        out["salu_busy"] = round(passes["lds"]["SQ_INSTS_SALU"] / (256.0 * cyc), 3)   # one scalar instruction per cycle per CU
        out["lds_busy"] = round(passes["lds"]["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc), 3)
        out["wait_any"] = round(passes["sq"]["SQ_WAIT_ANY"] / passes["sq"]["SQ_WAVE_CYCLES"], 3)
        if "util" in passes:
            u = passes["util"]
            out["valu_lane_utilisation"] = round(u["SQ_THREAD_CYCLES_VALU"] / (64.0 * u["SQ_ACTIVE_INST_VALU"]), 3)
    f = _newest(key + "_mem_counters.json")
    if f:
        busy = cyc = wr = rd = 0.0
        for name, c in json.load(open(f)).items():
            m = _TIMED_KERNEL.search(name)
            timed = (m and m.group(2) == "false" and m.group(3) == "true") or \
                    (passes and passes.get("_with_prepass") and "pt_primary<" in name)
            if timed and isinstance(c, dict) and c.get("TA_TA_BUSY_sum") and c.get("GRBM_GUI_ACTIVE"):
                busy += c["TA_TA_BUSY_sum"]
                cyc += c["GRBM_GUI_ACTIVE"]
                wr += c.get("SQ_INSTS_VMEM_WR", 0.0)
                rd += c.get("SQ_INSTS_VMEM_RD", 0.0)
        if cyc:
            out["ta_busy"] = round(busy / (256.0 * cyc / 8.0), 3)
            out["vmem_rd_insts"], out["vmem_wr_insts"] = int(rd), int(wr)
    if out:
        out["source"] = "committed PMC (profiles/), not measured in this run"
    return out


def spill_share(key):
    """Share of the timed kernel's vector-memory instructions that move spilled registers: (SQ_INSTS_VMEM of the shipped
    copy - SQ_INSTS_VMEM of the same kernel compiled without an occupancy bound, i.e. without spills; same control flow,
    same real loads and stores) / SQ_INSTS_VMEM of the shipped copy.  scripts/spill_share.sh measures both builds and
    commits profiles/rN/<key>_spill_share.json."""
    f = _newest(key + "_spill_share.json")
    if not f:
        return None, None, False
    d = json.load(open(f))
    from tracerboy_amd import build as tb_build
    stale = d.get("_kernel_digest") != tb_build.kernel_digest()     # measured of other device code: marked, like pmc_stale
    return d.get("vmem_spill_share"), os.path.relpath(f, ROOT), stale


def traffic_bytes(passes):
    """HBM bytes per launch: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- the counters are in KiB and gfx950's FETCH_SIZE reads
    half the bytes of 16-B/lane loads (MI355X_MICROARCH.md, HBM section)."""
    if not passes or "fetch" not in passes or "write" not in passes:
        return None
    return int((2.0 * passes["fetch"]["FETCH_SIZE"] + passes["write"]["WRITE_SIZE"]) * 1024)


def hbm_roofline(avg_ms, frames, pixels, st, passes, src):
    """The contract's block for a scene fetched from memory: achieved = SURVEY 8d algorithmic bytes per launch / launch
    time, peak = 8 TB/s, frac = achieved / peak; `traffic` = what the fabric carried per launch (PMC)."""
    bps = byte_model(st) / max(st.samples, 1)
    achieved = bps * pixels * frames / (avg_ms * 1e-3) / 1e9
    traffic = traffic_bytes(passes)
    n = max(st.samples, 1)
    r = {"bound": "hbm", "kernel": "pt_persistent", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_launch_ms": round(avg_ms, 3),
         "frames_per_launch": int(frames), "algorithmic_bytes_per_sample": round(bps, 1),
         "boxes_per_sample": round(st.boxesTested / n, 2), "tris_per_sample": round(st.trianglesTested / n, 2),
         "rays_per_sample": round(st.rays / n, 3), "pmc_source": src}
    if traffic:
        gbs = traffic / (avg_ms * 1e-3) / 1e9
        r["traffic_GBs"] = round(gbs, 1)
        r["traffic_frac_of_peak"] = round(gbs / HBM_PEAK_GBS, 4)
    return r


def add_pipe_fields(r, key, passes):
    """Beside the contract's fraction: which issue pipe the committed counters show busiest, how much of the VALU's busy
    time does useful work (VALUBusy x lane utilisation) and what share of the vector-memory instructions are spill
    traffic.  `frac` is NOT overwritten with a busy counter."""
    pipes = derived_busy(key, passes) if key else {}
    r["pipes"] = pipes
    r["pmc_stale"] = bool(passes and passes.get("_stale"))   # true: the committed counters were taken of other kernel code
    if pipes.get("ta_busy") is not None:
        name, busy = max(("vmem_issue", pipes["ta_busy"]), ("valu", pipes.get("valu_busy", 0.0)), key=lambda kv: kv[1])
        r["busiest_pipe"] = {"name": name, "busy": busy, "source": "committed PMC",
                             "definition": "TA_TA_BUSY / 256 TAs, or 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs, per launch cycle"}
    if pipes.get("valu_busy") is not None and pipes.get("valu_lane_utilisation") is not None:
        r["useful_issue_frac"] = round(pipes["valu_busy"] * pipes["valu_lane_utilisation"], 3)
    share, src, spill_stale = spill_share(key) if key else (None, None, False)
    r["vmem_spill_share"] = share
    if src:
        r["vmem_spill_source"] = src
        r["vmem_spill_stale"] = bool(spill_stale)
    # Which resource the committed counters say limits the kernel.  `frac` stays the contract's number (SURVEY 8d algorithmic bytes /
    # launch time / 8 TB/s -- "contract_bound": "hbm"); `bound` names the pipe that is actually saturated: the texture addresser's
    # issue of vector-memory instructions wherever it is busier than the fabric is full (the caches serve the bytes).
    if r.get("bound") == "hbm":
        r["contract_bound"] = "hbm"
        ta, fabric = pipes.get("ta_busy"), r.get("traffic_frac_of_peak")
        if ta is not None and fabric is not None and ta > fabric:
            r["bound"] = "vmem_issue"
            r["bound_note"] = ("texture addresser %.2f busy against %.2f of HBM peak on the fabric (committed PMC): the kernel is bound by "
                               "the issue of vector-memory instructions, not by DRAM bandwidth; frac is still the contract's byte rate / 8 TB/s" % (ta, fabric))
    return r


def cpu_allowance():
    """What the box lets this process use: the affinity mask, the cgroup CPU quota (v2 cpu.max, v1 cfs_quota / cfs_period;
    found by walking up from this process's cgroup) and the load others put on the machine.  A quota or busy neighbours
    explain an all-threads rate far below threads x single-thread rate; os.sched_getaffinity alone shows neither."""
    out = {"affinity_cpus": len(os.sched_getaffinity(0)), "machine_cpus": os.cpu_count(), "cgroup_quota_cpus": None,
           "cgroup_source": None}
    try:
        rel = ""
        for line in open("/proc/self/cgroup"):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and (parts[1] == "" or "cpu" in parts[1].split(",")):
                rel = parts[2]
                if parts[1] == "":
                    break
        cands = []
        d = rel
        while True:
            cands += ["/sys/fs/cgroup" + d + "/cpu.max", "/sys/fs/cgroup/cpu" + d + "/cpu.cfs_quota_us",
                      "/sys/fs/cgroup/cpu,cpuacct" + d + "/cpu.cfs_quota_us"]
            if d in ("", "/"):
                break
            d = os.path.dirname(d)
        best = None
        for f in cands:
            if not os.path.exists(f):
                continue
            if f.endswith("cpu.max"):
                q, per = open(f).read().split()[:2]
                val = None if q == "max" else float(q) / float(per)
            else:
                q = float(open(f).read())
                per = float(open(os.path.join(os.path.dirname(f), "cpu.cfs_period_us")).read())
                val = None if q <= 0 else q / per
            if out["cgroup_source"] is None:
                out["cgroup_source"] = f + (" (no limit)" if val is None else "")
            if val is not None and (best is None or val < best[0]):
                best = (val, f)
        if best:
            out["cgroup_quota_cpus"] = round(best[0], 2)
            out["cgroup_source"] = best[1]
    except Exception as e:  # noqa: BLE001 -- diagnostics only
        out["cgroup_source"] = "unreadable: %s" % e
    try:
        out["loadavg_1min"] = float(open("/proc/loadavg").read().split()[0])
    except Exception:  # noqa: BLE001
        pass
    return out


# ------------------------------------------------------------------------------------------------- expected speed-ups
def expected_speedup(scene, W, H, spp, depth, world):
    """configs[1]: what the tile split should give at this N, from the one-GPU stand-in measurement of a rank's own-tiles
    render, pack and un-permute (scripts/gather_standin.py -> profiles/rN/gather_standin.json) plus one xGMI hop of its
    packed tiles; render k + 1 overlaps gather k, so a step is the render and a fraction of a millisecond of exposed tail.
    The driver computes the measured efficiency from its own per-N runs."""
    f = _newest("gather_standin.json")
    if not f or (scene, W, H, spp, depth) != ("cornell-box", 1920, 1080, 64, 8):
        return None
    d = json.load(open(f))
    one, mine, hop = d.get("c2_rank0_of_1"), d.get("c2_rank0_of_%d" % world), d.get("1080p_world%d" % world)
    if not one or not mine:
        return None
    step = mine["render_ms"] + 0.1       # exposed tail of the pipelined step (pack + un-permute + what the gather does not hide)
    out = {"vs_1gpu": round(one["render_ms"] / step, 2), "render_ms_per_rank": mine["render_ms"],
           "gather_hop_us": hop["xgmi_hop_estimate_us"] if hop else None, "source": os.path.relpath(f, ROOT),
           "note": "one GPU emulating rank 0 of N; no multi-GPU hardware was available to the build"}
    # the same step as the timed region runs it (launches overlapping, pack and a stream-ordered consumer behind each
    # render), K steps back to back on one GPU as rank 0 of N (scripts/rank_share_async.py)
    piped = _newest("rank_share_async.json")
    if piped:
        a = json.load(open(piped))
        if a.get("world1") and a.get("world%d" % world):
            out["pipelined"] = {"vs_1gpu": round(a["world1"]["ms_per_step"] / a["world%d" % world]["ms_per_step"], 2),
                                "ms_per_step_per_rank": a["world%d" % world]["ms_per_step"],
                                "source": os.path.relpath(piped, ROOT)}
    return out


def expected_speedup_leg(key, world):
    """The 4K configurations: one GPU emulated EVERY rank r of N in turn on the leg's scene with the leg's tile size
    (scripts/rank_imbalance.py --spp 32 -> profiles/rN/rank_imbalance_32spp.json: the step the leg is quoted on; without --spp ->
    rank_imbalance.json: the 8-spp short step); a step of the N-GPU job takes what its slowest rank takes, so the expected speed-up is
    t(1 GPU) / max_r t(rank r of N) and max / mean over the ranks is the tile imbalance SURVEY 8e names as the limiter."""
    def row(pattern, spp):
        f = _newest(pattern)
        if not f:
            return None
        d = json.load(open(f)).get(key)
        if not d or "world1" not in d or ("world%d" % world) not in d or d.get("tile", TILE) != WORKLOADS[key].get("tile", TILE):
            return None
        one, w = d["world1"], d["world%d" % world]
        return {"spp": spp, "vs_1gpu": round(one["max_ms"] / w["max_ms"], 2), "max_over_mean_rank_ms": w["max_over_mean"],
                "slowest_rank": w.get("slowest_rank"), "ms_per_step_slowest_rank": w["max_ms"], "tile": d.get("tile", TILE),
                "deal": d.get("deal", "round-robin"), "source": os.path.relpath(f, ROOT)}
    main, short = row("rank_imbalance_32spp.json", SCALE_SPP), row("rank_imbalance.json", SCALE_SHORT_SPP)
    if not main:
        return None
    main["at_short_steps"] = short
    main["note"] = ("one GPU emulating each rank of N in turn (async step: render + pack + stream-ordered consumer); the xGMI gather "
                    "(16.6 MB per peer at N = 8) overlaps the next render.  A launch costs a fixed 0.8-2.7 ms more than its samples on a rank's "
                    "share of a 4K frame (scripts/async_rate.py): a third of an 8-spp step, a twentieth of the 32-spp step quoted")
    return main


# ------------------------------------------------------------------------------------------------- the renderer
class Bench:
    """The context, the scene loader and the settings of one workload."""

    def __init__(self, api, device):
        self.api = api
        self.tb = api.TracerBoy(device)
        self.loaded = None

    def settings(self, depth):
        s = self.api.GetDefaultOutputSettings()
        s.EnableBlueNoise = 0       # SURVEY.md 8d "Common": pure rand() path, Time = 0, NEE on, RIS off, box filter
        s.MaxBounces = depth
        return s

    def load(self, scene, builder, opts=None):
        """Returns the seconds the load took (host parse / generation + BVH build + upload)."""
        tb = self.tb
        tb.SetOption("bvh_builder", builder)
        tb.SetOption("reinsertion_passes", -1)     # the library's own choice unless the workload says otherwise (options outlive a load)
        tb.SetOption("reinsertion_share", 100)
        for k, v in (opts or {}).items():
            tb.SetOption(k, v)
        t0 = time.time()
        if scene == "cornell-box":
            tb.LoadScene(CORNELL)
        elif scene.startswith("proc"):
            kind, tris = scene[4:].split(":")
            tb.LoadProcedural(int(kind), int(tris), 1234)
        else:
            tb.LoadScene(scene)
        self.loaded = (scene, builder, tuple(sorted((opts or {}).items())))
        return time.time() - t0

    def load_workload(self, key):
        w = WORKLOADS[key]
        return self.load(w["scene"], w["builder"], w.get("opts"))


def scene_label(scene):
    return os.path.basename(os.path.dirname(scene)) if scene.endswith(".pbrt") else scene


def data_label(scene):
    if scene.startswith("proc"):
        return "synthetic (procedural generator %s, seed 1234, built in the run)" % scene
    rel = os.path.relpath(CORNELL if scene == "cornell-box" else scene, ROOT)
    return "scene file %s (the reference's Scenes/%s as committed under tests/golden; no images or weights involved)" % (
        rel, scene_label(CORNELL if scene == "cornell-box" else scene))


def measure_kernel(tb, W, H, spp, s, runs):
    """HIP-event duration of `runs` path-tracing launches run one at a time (events recorded on the stream the kernel is
    launched on, tb_last_render_ms) + the kernels' own event counters from a 1-spp counting launch of the same code."""
    ms = []
    for _ in range(runs):
        tb.InvalidateHistory()
        tb.Render(W, H, spp, s, 0.0)
        ms.append(tb.GetOption("last_kernel_us") / 1e3)
    frames = tb.GetOption("last_kernel_frames")
    measure_kernel.guided = bool(tb.GetOption("last_plan_guided_groups"))   # did these launches get the shrinking end (calls that wait, scenes in LDS)?
    tb.SetOption("count_rays", 1)
    tb.Render(W, H, 1, s, 0.0)           # counters-on launch of the same kernels, 1 spp, outside every timed region
    st = tb.ReadbackStats().rays
    tb.SetOption("count_rays", 0)
    tb.InvalidateHistory()
    return sum(ms) / len(ms), frames, st


def settle_overlap(tb, W, H, spp, s):
    """The library finds out by itself whether back-to-back calls of a kind should share the chip (two launches in flight)
    or take turns -- it needs a few rounds of asynchronous calls to see device-bound intervals both ways (renderImpl,
    overlap trial).  matte / env: overlapped launches always pay, nothing is tried."""
    for _ in range(4):
        if tb.GetOption("overlap_trial_phase") == 2:
            break
        for _ in range(5):
            tb.InvalidateHistory()
            tb.Render(W, H, spp, s, 0.0, sync=False)
        tb.Sync()
        if tb.GetOption("last_variant") in (0, 1):
            break


class TileSplit:
    """The N > 1 step for one frame size: this rank's own-tiles render, the device-side pack of its tiles, ONE gather of the
    packed buffers to rank 0 (RCCL over xGMI) and rank 0's device-side un-permute into the full frame.

    Equal-sized slices: every rank pads to the largest owner (rank 0) so one gather per render suffices.  Two packed
    buffers: the gather of one render runs on RCCL's stream while the next render traces, and a buffer is packed again
    only after the gather that read it (two renders ago) has finished.  Rank 0 gathers into ONE contiguous
    world x capacity buffer (views per rank) and un-permutes it on the device (tb_unpack_gathered_device), ordered behind
    the gather.  Nothing in a step blocks the host: render and pack are enqueued on the library's stream, the gather on
    RCCL's, the un-permute on torch's stream behind the gather, ordered by stream waits."""

    def __init__(self, tb, torch, dist, tiles, rank, world, backend, W, H, standin_consumer=False, tile=TILE):
        self.tb, self.torch, self.dist, self.rank, self.world, self.backend = tb, torch, dist, rank, world, backend
        self.W, self.H, self.tile = W, H, tile
        tb.SetTileAssignment(rank, world, tile, tile)
        self.owned = tb.OwnedPixels(W, H)
        self.capacity = max(tiles.packed_capacity(W, H, world, tile, tile), 1)
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device="cuda")   # noqa: E731
        self.packed = [z(self.capacity, 4) for _ in range(2)]
        root = world > 1 and rank == 0
        self.gathered = z(world, self.capacity, 4) if root else None
        self.gather_list = [self.gathered[r] for r in range(world)] if root else None
        self.frame = z(H, W, 4) if root else None
        self.scratch = torch.zeros_like(self.packed[0]) if (world == 1 and standin_consumer) else None
        self.in_flight = [None, None]
        self.renders = 0
        self.lib_stream = torch.cuda.ExternalStream(tb.Stream())
        torch.cuda.synchronize()

    def _unpack(self):
        torch = self.torch
        self.tb.UnpackGatheredTo(self.gathered.data_ptr(), self.capacity, self.W, self.H, self.world, self.tile, self.tile,
                                 self.frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)

    def _host_gather(self, buf):
        """Test hook (TB_BENCH_BACKEND=gloo): the same collective through host memory, synchronously."""
        torch, dist = self.torch, self.dist
        host = buf.cpu()
        parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(host, parts, dst=0)
        if self.rank == 0:
            self.gathered.copy_(torch.stack(parts))

    def exchange(self, buf):
        torch, dist = self.torch, self.dist
        if self.world > 1 and self.backend != "nccl":
            torch.cuda.current_stream().synchronize()
            self._host_gather(buf)
            if self.rank == 0:
                self._unpack()
            return None
        if self.world > 1:
            work = dist.gather(buf, self.gather_list if self.rank == 0 else None, dst=0, async_op=True)
            if self.rank == 0:
                work.wait()       # torch's current stream waits for the gather (the host does not) ...
                self._unpack()    # ... and the un-permute runs behind it
            return work
        self.scratch.copy_(buf, non_blocking=True)   # --async-steps on one GPU: a stand-in consumer on torch's stream
        return None

    def step(self, spp, s):
        torch, tb = self.torch, self.tb
        b = self.renders & 1
        self.renders += 1
        tb.InvalidateHistory()
        tb.Render(self.W, self.H, spp, s, 0.0, sync=False)
        if self.in_flight[b] is not None:
            self.in_flight[b].wait()                                  # torch's stream waits for the gather that last read packed[b] ...
        self.lib_stream.wait_stream(torch.cuda.current_stream())      # ... and the library's stream waits for torch's
        tb.PackOwnedTo(self.packed[b].data_ptr(), sync=False)
        torch.cuda.current_stream().wait_stream(self.lib_stream)      # the gather reads what the library's stream packed
        self.in_flight[b] = self.exchange(self.packed[b])

    def breakdown(self, spp, s, barrier, reps=3):
        """Where a step's time goes, per rank: the stages run one at a time with a device sync between them, so the figures
        are each stage's own cost, not its share of the overlapped pipeline.  Per stage the MAX over ranks (the slowest
        rank is what a step waits for) and the mean; best of `reps`."""
        torch, dist, tb = self.torch, self.dist, self.tb
        stages = {"render_ms": [], "pack_ms": [], "gather_ms": [], "unpack_ms": []}
        for _ in range(reps):
            barrier()
            t = time.perf_counter()
            tb.InvalidateHistory()
            tb.Render(self.W, self.H, spp, s, 0.0)
            stages["render_ms"].append((time.perf_counter() - t) * 1e3)
            t = time.perf_counter()
            tb.PackOwnedTo(self.packed[0].data_ptr(), sync=False)
            tb.Sync()
            stages["pack_ms"].append((time.perf_counter() - t) * 1e3)
            barrier()
            t = time.perf_counter()
            if self.backend == "nccl":
                dist.gather(self.packed[0], self.gather_list if self.rank == 0 else None, dst=0)
                torch.cuda.synchronize()
            else:
                self._host_gather(self.packed[0])
                torch.cuda.synchronize()
            stages["gather_ms"].append((time.perf_counter() - t) * 1e3)
            t = time.perf_counter()
            if self.rank == 0:
                self._unpack()
                torch.cuda.synchronize()
            stages["unpack_ms"].append((time.perf_counter() - t) * 1e3)
        dev = "cuda" if self.backend == "nccl" else "cpu"
        mine = torch.tensor([min(v) for v in stages.values()], dtype=torch.float64, device=dev)
        mx = mine.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = mine.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        out = {k: round(float(mx[i]), 3) for i, k in enumerate(stages)}
        out["mean_over_ranks"] = {k: round(float(sm[i]) / self.world, 3) for i, k in enumerate(stages)}
        out["render_max_over_mean"] = round(float(mx[0]) / max(float(sm[0]) / self.world, 1e-9), 3)
        out["gather_bytes_per_rank"] = int(self.capacity) * 16
        out["owned_pixels_rank0"] = int(self.owned) if self.rank == 0 else None
        out["note"] = ("stages run one at a time after the timed region (best of %d, max over ranks); in the timed steps "
                       "render k+1 overlaps gather k" % reps)
        return out

    def assembled_equals_single_gpu(self, np, spp, s):
        """Rank 0, after the timed region: the frame it assembled from the gathered tiles equals a single-GPU render of the
        whole frame, bit for bit."""
        tb = self.tb
        assembled = self.frame.cpu().numpy()
        tb.SetTileAssignment(0, 1)
        tb.InvalidateHistory()
        tb.Render(self.W, self.H, spp, s, 0.0)
        whole = tb.ReadAccumulation()
        tb.SetTileAssignment(self.rank, self.world, self.tile, self.tile)
        return bool(np.array_equal(assembled.view(np.uint32), whole.view(np.uint32)))


def timed_steps(step, barrier, warmup, steps):
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0



# ------------------------------------------------------------------------------------------------- parity gate
NO_PARITY = False          # --no-parity (profiling runs)
PARITY_TOL = 1e-4          # BASELINE.md section 2 / north_star: relative L2 per pixel <= 1e-4 before any timing is accepted
ORACLE_NAME = "oracle/tb_oracle.cpp"


def parity_compare(np, gpu, cpu):
    """Radiance sums of the HIP path against the oracle's over the same pixels (RGBA32F, .w = the weight): bit equality, the
    relative L2 of the whole block and the largest per-pixel relative L2 (north_star words the gate per pixel)."""
    g, c = np.ascontiguousarray(gpu, np.float32), np.ascontiguousarray(cpu, np.float32)
    equal = bool(np.array_equal(g.view(np.uint32), c.view(np.uint32)))
    if equal:
        return {"bit_equal": True, "rel_l2": 0.0, "max_pixel_rel_l2": 0.0, "pixels": int(g.shape[0] * g.shape[1]), "differing_pixels": 0}
    d = g.astype(np.float64) - c.astype(np.float64)
    whole = float(np.sqrt((d ** 2).sum()) / max(np.sqrt((c.astype(np.float64) ** 2).sum()), 1e-30))
    px = np.sqrt((d ** 2).sum(axis=-1)) / np.maximum(np.sqrt((c.astype(np.float64) ** 2).sum(axis=-1)), 1e-30)
    px = np.where(np.isfinite(px), px, np.inf)       # a NaN on either side is a failure, not a zero
    if np.isnan(d).any():
        whole = float("inf")
    return {"bit_equal": False, "rel_l2": whole, "max_pixel_rel_l2": float(px.max()), "pixels": int(g.shape[0] * g.shape[1]),
            "differing_pixels": int((g.view(np.uint32) != c.view(np.uint32)).any(axis=-1).sum())}


def parity_ok(block):
    return bool(block and (block.get("bit_equal") or (block.get("rel_l2", 1.0) <= PARITY_TOL and block.get("max_pixel_rel_l2", 1.0) <= PARITY_TOL)))


def strip_rows(H, k=2):
    """k 8-row strips inside the frame: at 1/3 and 2/3 of its height (rows of whole 8x8 tiles)."""
    return [min(max(0, (H * (i + 1) // (k + 1)) // 8 * 8), max(0, H - 8)) for i in range(k)]


def parity_strips(tb, np, frame, W, H, frames, s, threads, rows=None):
    """8-row strips of `frame` (the GPU's accumulation of `frames` frames) against the oracle fed the same host scene and frame
    constants: the gate of one leg, a second or two on the box's CPUs."""
    import oracle_lib as ol
    rows = strip_rows(H) if rows is None else rows
    view, pf = tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, 0.0)
    t0 = time.perf_counter()
    g, c = [], []
    for y0 in rows:
        y1 = min(H, y0 + 8)
        ref = ol.render(view, pf, W, H, frames, y0=y0, y1=y1, threads=threads)["output"]
        g.append(frame[y0:y1]); c.append(ref[y0:y1])
    blk = parity_compare(np, np.concatenate(g), np.concatenate(c))
    blk.update({"frames": int(frames), "against": ORACLE_NAME, "compared": "8-row strips at y = %s of the %dx%d frame" % (rows, W, H),
                "tolerance": PARITY_TOL, "oracle_s": round(time.perf_counter() - t0, 2)})
    blk["ok"] = parity_ok(blk)
    return blk


def oracle_threads():
    cores = len(os.sched_getaffinity(0))
    q = cpu_allowance().get("cgroup_quota_cpus")
    return max(1, min(cores, int(q + 0.999))) if q else cores


def gate(block, r, key="value"):
    """A figure whose frame failed the gate is not a figure: value -> null, the measured number kept under value_unverified."""
    r["parity"] = block
    if not (block or {}).get("ok", parity_ok(block)):
        r[key + "_unverified"] = r.get(key)
        r[key] = None
    return r


# ------------------------------------------------------------------------------------------------- N = 1 legs
def extra_leg(b, np, torch, key, steps):
    """One more workload under the driver's clock: loaded, warmed, `steps` renders enqueued back to back and waited for
    once (like the timed region), then launches run one at a time for the roofline figures; counters from
    profiles/rN/<key>_{pmc_summary,mem_counters}.json."""
    w = WORKLOADS[key]
    tb, W, H, SPP, D = b.tb, w["W"], w["H"], w["spp"], w["depth"]
    s = b.settings(D)
    load_s = b.load_workload(key)
    info = tb.SceneInfo()
    for _ in range(2):     # warm-up: first launch of this kernel copy, both sample buffers / side streams
        tb.InvalidateHistory()
        tb.Render(W, H, SPP, s, 0.0)
    settle_overlap(tb, W, H, SPP, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tb.InvalidateHistory()
        tb.Render(W, H, SPP, s, 0.0, sync=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    variant = VARIANTS[tb.GetOption("last_variant")]
    prepass, overlapped = bool(tb.GetOption("last_primary_prepass")), bool(tb.GetOption("last_overlap"))
    two_level = bool(w.get("opts", {}).get("flatten_instances", 1) == 0)
    # the gate: the frame the last timed step left in HBM, two 8-row strips of it against the oracle at the
    # leg's own sample count
    parity = None if NO_PARITY else parity_strips(tb, np, tb.ReadAccumulation(), W, H, SPP, s, oracle_threads())
    avg, frames, st = measure_kernel(tb, W, H, SPP, s, steps)
    passes, src = pmc_summary(key)
    r = hbm_roofline(avg, frames, W * H, st, passes, src)
    r.update({"workload": "%s %dx%d %dspp depth%d%s" % (scene_label(w["scene"]), W, H, SPP, D, " two-level" if two_level else ""),
              "data": data_label(w["scene"]), "triangles": int(info.numTriangles),
              "value": round(W * H * SPP * steps / dt / 1e6, 1), "unit_value": "Msamples/s",
              "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "scene_load_s": round(load_s, 2),
              "bvh_builder": builder_label(w), "kernel_variant": variant, "primary_prepass": prepass,
              "launches_overlap": overlapped, "kernel": "pt_primary + pt_persistent" if prepass else "pt_persistent",
              "frac_note": "SURVEY 8d byte model x samples / launch time / 8 TB/s: traversal throughput in the reference's "
                           "units; the bytes are served mostly by L2 / Infinity Cache (traffic_GBs is what the fabric carries)"})
    if frames < SPP:
        r["note"] = "launches are batched by the sample-buffer budget: avg_launch_ms / frames_per_launch are per batch launch"
    if prepass:
        r["note"] = (r.get("note", "") + "; " if r.get("note") else "") + (
            "a launch is the pair primary-visibility pre-pass (pt_primary: every camera ray of the batch, one 8x8 pixel tile "
            "per wave) + lock-step kernel: avg_launch_ms spans both, the committed counters are their sums")
    add_pipe_fields(r, key, passes)
    if parity is not None:
        gate(parity, r)
    if key == "c3":
        r["what_limits_it"] = ("instruction issue at ~19 of 64 lanes: a CU's texture addresser takes ~17 cycles per wave-level "
                               "load whatever the number of active lanes (scripts/microbench/gather64.hip) and every VALU "
                               "instruction of the walk pays for 64 lanes (DESIGN.md section 6)")
        # the same workload through the compact nodes (option node_layout = 1: 32-B nodes on a 16-bit grid, within 1e-4
        # relative L2 of the bit-exact path)
        tb.SetOption("node_layout", 1)
        tb.InvalidateHistory()
        tb.Render(W, H, SPP, s, 0.0)
        if tb.GetOption("last_node_layout") == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                tb.InvalidateHistory()
                tb.Render(W, H, SPP, s, 0.0, sync=False)
            torch.cuda.synchronize()
            dtc = time.perf_counter() - t0
            r["compact_nodes"] = {"value": round(W * H * SPP * steps / dtc / 1e6, 1), "unit_value": "Msamples/s",
                                  "ms_per_step": round(dtc / steps * 1e3, 3), "option": "node_layout=1",
                                  "contract": "relative L2 <= 1e-4 against the bit-exact layout-B path "
                                              "(tests/test_compact_nodes.py), not bit equality"}
        tb.SetOption("node_layout", 0)
    return r


def pcie_leg(tb, np, W, H, SPP, s, steps):
    """The same render with the frame handed to the host (tb_read_accum: one D2H copy of the RGBA32F sums into a host array
    the caller owns, pageable memory as a ctypes/numpy caller has it).  Reported beside `value`, never as it."""
    import ctypes
    host = np.zeros((H, W, 4), np.float32)
    hp = host.ctypes.data_as(ctypes.c_void_p)
    tb.InvalidateHistory()
    tb.Render(W, H, SPP, s, 0.0)
    tb._check(tb._L.tb_read_accum(tb._ctx, hp, None))
    n = max(2, min(steps, 10))
    t = time.perf_counter()
    for _ in range(n):
        tb.InvalidateHistory()
        tb.Render(W, H, SPP, s, 0.0)
        tb._check(tb._L.tb_read_accum(tb._ctx, hp, None))
    t = (time.perf_counter() - t) / n
    return {"value": round(W * H * SPP / t / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(t * 1e3, 3),
            "readback_bytes": int(host.nbytes),
            "note": "render + tb_read_accum into pageable host memory, synchronous, %d steps" % n}


def cpu_baseline_leg(tb, args, W, H, SPP, s, np=None, timed_frame=None):
    """The scalar oracle (oracle/tb_oracle.cpp) on a bounded sample of the timed workload -- whole frames, sized to about
    --cpu-baseline-seconds -- on the threads the box lets run at once, and on one thread.  N = 1 only (the contract; at
    N > 1 the other ranks' host threads would share the quota with it: ADVICE r4).
    The image it renders is the gate of the headline (CPU image vs HIP image before any timing is accepted):
    returns (cpu_baseline block, parity block).  `timed_frame` = the accumulation the last timed step left in HBM (SPP frames); when
    the CPU sample covers fewer frames than SPP the GPU renders that many frames again for the comparison and the timed frame itself
    is gated on strips at the full sample count."""
    import oracle_lib as ol
    allowance = cpu_allowance()
    cores = len(os.sched_getaffinity(0))      # the CPUs this process may run on, not the machine's ...
    if allowance.get("cgroup_quota_cpus"):    # ... and not more threads than the cgroup lets run at once
        cores = max(1, min(cores, int(allowance["cgroup_quota_cpus"] + 0.999)))
    view = tb.HostSceneView()
    pf = tb.FrameConstants(W, H, 0, s, 0.0)
    budget = args.cpu_baseline_seconds
    t1 = time.perf_counter()
    ol.render(view, pf, W, H, 1, threads=cores)     # probe: one full 1-spp frame
    dt = time.perf_counter() - t1
    frames = int(max(1, min(SPP, budget / max(dt, 1e-3))))
    c0 = os.times()
    t1 = time.perf_counter()
    cpu_image = ol.render(view, pf, W, H, frames, threads=cores)["output"]
    dt = time.perf_counter() - t1
    c1 = os.times()
    parity = None
    if np is not None:
        if timed_frame is not None and frames == SPP:
            gpu_image, what = timed_frame, "the frame the last timed step left in HBM"
        else:
            tb.InvalidateHistory()
            tb.Render(W, H, frames, s, 0.0)
            gpu_image, what = tb.ReadAccumulation(), "a render of the CPU sample's %d frames after the timed region" % frames
        parity = parity_compare(np, gpu_image, cpu_image)
        parity.update({"frames": int(frames), "of_frames": int(SPP), "against": ORACLE_NAME, "tolerance": PARITY_TOL,
                       "compared": "the whole %dx%d frame x %d spp: %s" % (W, H, frames, what), "timed_frame": bool(frames == SPP and timed_frame is not None)})
        if timed_frame is not None and frames != SPP:    # a slow box: the timed frame itself, on strips at its full sample count
            parity["timed_frame_strips"] = parity_strips(tb, np, timed_frame, W, H, SPP, s, cores)
        parity["ok"] = parity_ok(parity) and parity_ok(parity.get("timed_frame_strips", parity))
    del cpu_image
    cpu_seconds = (c1.user - c0.user) + (c1.system - c0.system)   # cpu_seconds / dt = CPUs' worth of service
    if cores == 1:
        dt1, n1, rows1 = dt, W * H * frames, "the same run"
    else:
        est1 = dt * cores / max(frames, 1)     # estimated single-thread seconds per whole frame
        budget1 = 0.4 * budget                 # the single-thread figure is a side note: 4 of the 14 s the two samples take together
        f1 = int(max(1, min(frames, budget1 / max(est1, 1e-3))))
        if est1 <= 3.0 * budget:
            t2 = time.perf_counter()
            ol.render(view, pf, W, H, f1, threads=1)
            dt1, n1, rows1 = time.perf_counter() - t2, W * H * f1, "whole frame x %d spp" % f1
        else:                                   # a whole frame on one thread would take minutes: every k-th 8-row strip
            k = int(est1 / budget1) + 1
            strips = list(range(0, H, 8 * k))
            t2 = time.perf_counter()
            for y0 in strips:
                ol.render(view, pf, W, H, 1, y0=y0, y1=min(H, y0 + 8), threads=1)
            dt1 = time.perf_counter() - t2
            n1 = sum(W * (min(H, y0 + 8) - y0) for y0 in strips)
            rows1 = "every %d-th 8-row strip of the frame x 1 spp" % k
    affinity = len(os.sched_getaffinity(0))
    cb = {"value": round(W * H * frames / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
          "sample": "scalar C++ oracle (oracle/tb_oracle.cpp, g++ -O2), the %dx%d frame x %d spp of %d, depth %d, %d threads "
                    "over row strips (>= 4 work items per thread) (%.1f s)" % (W, H, frames, SPP, args.depth, cores, dt),
          "single_thread": round(n1 / dt1 / 1e6, 4), "single_thread_sample": rows1 + " (%.1f s)" % dt1,
          "machine_cpus": os.cpu_count(), "box": allowance}
    cb["effective_parallelism"] = round(cb["value"] / max(cb["single_thread"], 1e-9), 1)
    cb["cpus_worth_of_service"] = round(cpu_seconds / max(dt, 1e-9), 1)
    cb["unthrottled_estimate"] = {"value": round(cb["single_thread"] * affinity, 2), "unit": "Msamples/s",
                                  "note": "single-thread rate x the %d hardware threads of the box: what the same port "
                                          "would reach there without a quota" % affinity}
    cb["note"] = ("the all-threads figure is what THIS lease delivers: effective_parallelism is well below the thread count "
                  "when the box throttles (box.cgroup_quota_cpus) or shares its cores (cpus_worth_of_service << threads); "
                  "quote speed-ups against both figures")
    if args.scene == "cornell-box":
        # BASELINE.json configs[0], the reference's own CPU-runnable case, timed exactly: 512x512, 4 spp, depth 4
        import copy
        s0 = copy.copy(s)
        s0.MaxBounces = 4
        pf0 = tb.FrameConstants(512, 512, 0, s0, 0.0)
        t3 = time.perf_counter()
        ol.render(view, pf0, 512, 512, 4, threads=cores)
        dt3 = time.perf_counter() - t3
        t4 = time.perf_counter()
        ol.render(view, pf0, 512, 512, 4, threads=1)
        dt4 = time.perf_counter() - t4
        n0 = 512 * 512 * 4
        cb["configs0"] = {"workload": "cornell-box 512x512 4spp depth4", "all_threads_s": round(dt3, 4),
                          "single_thread_s": round(dt4, 3), "all_threads": round(n0 / dt3 / 1e6, 3),
                          "single_thread": round(n0 / dt4 / 1e6, 4)}
    return cb, parity


def cpu_baseline_copied():
    """N > 1: the CPU baseline is measured at N = 1 only; the record carries the newest committed N = 1 figure, marked."""
    f = _newest("bench_default.json")
    if not f:
        return {"measured": False, "note": "measured at N = 1 only (see the N = 1 line)"}
    try:
        cb = json.load(open(f)).get("cpu_baseline") or {}
    except Exception:  # noqa: BLE001
        cb = {}
    keep = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "single_thread") if k in cb}
    keep.update({"measured": False, "copied_from": os.path.relpath(f, ROOT),
                 "note": "NOT measured in this run: the CPU baseline is timed at N = 1 only (rank 0's oracle threads would "
                         "share the host quota with the other ranks); this is the committed N = 1 record's figure"})
    return keep


# ------------------------------------------------------------------------------------------------- N > 1 legs
def scale_leg(b, np, torch, dist, tiles, key, rank, world, backend, steps, barrier):
    """One of the configurations the 8 GPUs divide through the tile-split step: every rank loads the scene, renders its own
    tiles, packs, ONE gather, rank 0 un-permutes -- `steps` steps between two barriers, max over ranks."""
    w = WORKLOADS[key]
    tb, W, H, SPP, D = b.tb, w["W"], w["H"], w["spp"], w["depth"]
    s = b.settings(D)
    load_s = b.load_workload(key)
    info = tb.SceneInfo()
    # two launches in flight, always: what the library's own trial settles on for these feature sets at 4K (4K glass scenes
    # +4-5 %, vw-van +18-39 %, profiles/r4/overlap_ab*.json) and a rank's share of a frame is a smaller call still; a trial
    # would not settle within the leg's few steps.  scripts/rank_imbalance.py (expected_speedup) runs the same way.
    tb.SetOption("overlap_launches", 2)
    tile = w.get("tile", TILE)
    ts = TileSplit(tb, torch, dist, tiles, rank, world, backend, W, H, tile=tile)
    dev = "cuda" if backend == "nccl" else "cpu"
    # the short step first (8 spp, the N = 1 legs' step), then the step the leg is quoted on: the frame that stays assembled is the long step's
    short = timed_steps(lambda: ts.step(SCALE_SHORT_SPP, s), barrier, 2, steps)
    SPP = SCALE_SPP
    elapsed = timed_steps(lambda: ts.step(SPP, s), barrier, 1, steps)
    t = torch.tensor([elapsed, load_s, short], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, load_max, short = float(t[0]), float(t[1]), float(t[2])
    tb.Sync()
    r = {"workload": "%s %dx%d %dspp depth%d" % (scene_label(w["scene"]), W, H, SPP, D), "data": data_label(w["scene"]),
         "triangles": int(info.numTriangles), "value": round(W * H * SPP * steps / elapsed / 1e6, 1), "unit": "Msamples/s",
         "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "n_gpus": world, "scaling": "strong",
         "at_short_steps": {"spp": SCALE_SHORT_SPP, "value": round(W * H * SCALE_SHORT_SPP * steps / short / 1e6, 1), "unit": "Msamples/s",
                            "ms_per_step": round(short / steps * 1e3, 3),
                            "note": "the same step at the N = 1 legs' 8 spp: a rank's launch then carries its fixed cost (its drain: the longest paths of "
                                    "its last samples) over an eighth of the work"},
         "parallelism": "tiles%d" % world, "tile": tile, "bvh_builder": builder_label(w),
         "kernel_variant": VARIANTS[tb.GetOption("last_variant")], "primary_prepass": bool(tb.GetOption("last_primary_prepass")),
         "launches_overlap": bool(tb.GetOption("last_overlap")), "scene_load_s": round(load_max, 2),
         # this rank's launches handed the regions with long interior walks out first (launch_plan.h costly_first: small calls of the glass feature sets)
         "costly_regions_first": bool(tb.GetOption("last_plan_costly_first")),
         "scene_load_s_note": "max over ranks (every rank loads and builds the scene itself)"}
    r["scale_breakdown"] = ts.breakdown(SPP, s, barrier, reps=2)
    exp = expected_speedup_leg(key, world)
    if exp:
        r["expected_speedup"] = exp
    if rank == 0:
        torch.cuda.synchronize()
        assembled = ts.frame.cpu().numpy()
        r["assembled_frame_equals_single_gpu"] = ts.assembled_equals_single_gpu(np, SPP, s)
        if not NO_PARITY:
            gate(parity_strips(tb, np, assembled, W, H, SPP, s, oracle_threads()), r)   # the assembled frame of the last timed step against the oracle
    tb.SetOption("overlap_launches", 1)
    dist.barrier()
    return r


# ------------------------------------------------------------------------------------------------- main
def main():
    global NO_PARITY
    args = parse_args()
    NO_PARITY = args.no_parity
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))          # nothing above this line imports torch or touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if args.selftest_cpu:
        sys.exit(selftest_cpu(args, rank, world))

    import numpy as np
    import torch
    import torch.distributed as dist
    from tracerboy_amd import api, tiles

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path is HIP-only (no CPU fallback)")
    # Test hooks for boxes with fewer GPUs than ranks (tests/test_gpu_parity.py): TB_BENCH_SHARE_DEVICE=1 puts every rank on
    # device 0, TB_BENCH_BACKEND=gloo moves the gather through host memory (RCCL refuses two ranks on one device).
    # Everything else of the N > 1 step -- own-tiles launch, pack, one gather per render, device-side un-permute, the
    # assembled-frame check -- is the real code.
    backend = os.environ.get("TB_BENCH_BACKEND", "nccl")
    if os.environ.get("TB_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = "cuda" if backend == "nccl" else "cpu"

    W, H, SPP = args.width, args.height, args.spp
    b = Bench(api, local_rank)
    tb = b.tb
    s = b.settings(args.depth)
    tb.SetOption("pipeline", args.pipeline)
    main_opts = {}
    for kv in args.opt:
        k, v = kv.split("=")
        main_opts[k] = int(v)
    load_s = b.load(args.scene, args.builder, main_opts)
    info = tb.SceneInfo()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region -----------------------------------------------------------------------------------------
    pipelined = world > 1 or args.async_steps
    kernel_ms = []
    if pipelined:
        ts = TileSplit(tb, torch, dist, tiles, rank, world, backend, W, H, standin_consumer=args.async_steps)
        step = lambda: ts.step(SPP, s)   # noqa: E731
    else:
        ts = None
        tb.SetTileAssignment(0, 1)

        def step():
            tb.InvalidateHistory()
            if args.sync_steps:
                tb.Render(W, H, SPP, s, 0.0)     # synchronous; GPU time measured with HIP events on the library's stream
                kernel_ms.append(tb.GetOption("last_kernel_us") / 1e3)
            else:
                # the K renders are enqueued back to back (tb_render_async) and waited for once by the closing barrier
                # (torch.cuda.synchronize = device-wide), inside the timed region: the launch of step k+1 starts on the
                # other side stream while the last paths of step k drain
                tb.Render(W, H, SPP, s, 0.0, sync=False)

    for _ in range(args.warmup):
        step()
    kernel_ms.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if pipelined or not args.sync_steps:
        tb.Sync()
        kernel_ms.append(tb.GetOption("last_kernel_us") / 1e3)   # HIP events of the last render of the timed region
    # the frame the last timed step left in HBM, for the parity gate below (N = 1: the accumulation surface; N > 1: what rank 0
    # assembled from the gathered tiles)
    timed_frame = None
    if rank == 0:
        torch.cuda.synchronize()
        timed_frame = ts.frame.cpu().numpy() if world > 1 else tb.ReadAccumulation()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_step = W * H * SPP
    value = samples_per_step * args.steps / elapsed / 1e6
    result = {
        "metric": "Msamples/s (WxHxspp/s)", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": data_label(args.scene),
        "config": {"workload": "%s %dx%d %dspp depth%d" % (args.scene, W, H, SPP, args.depth),
                   "triangles": int(info.numTriangles), "bvh_builder": BUILDERS[args.builder],
                   "pipeline": PIPELINES[tb.GetOption("last_pipeline")], "tile": TILE if world > 1 else None,
                   "parallelism": "tiles%d" % world, "scene_in_lds": bool(tb.GetOption("scene_in_lds_active")),
                   "kernel_variant": VARIANTS[tb.GetOption("last_variant")], "scene_load_s": round(load_s, 3)},
    }

    if world > 1:
        tl = torch.tensor([load_s], device=dev, dtype=torch.float64)
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        result["config"]["scene_load_s"] = round(float(tl.item()), 3)
        result["config"]["scene_load_s_note"] = "max over ranks (every rank loads and builds the scene itself)"
        result["scale_breakdown"] = ts.breakdown(SPP, s, barrier)
        exp = expected_speedup(args.scene, W, H, SPP, args.depth, world)
        if exp:
            result["expected_speedup"] = exp
        result["rccl_ranks"] = int(dist.get_world_size())
        result["collective_backend"] = backend
        if rank == 0:
            result["config"]["assembled_frame_equals_single_gpu"] = ts.assembled_equals_single_gpu(np, SPP, s)
        dist.barrier()

    # ---- roofline of the dominant (only) kernel of the timed workload: pt_persistent (rank 0) -------------------------
    if rank == 0:
        lds = bool(tb.GetOption("scene_in_lds_active"))
        launch_timing = "HIP events around the path-tracing launch of the last timed step (tb_last_render_ms)"
        if world == 1 and not args.sync_steps and not args.async_steps:
            # the timed steps overlap (the next launch starts while the last paths of the previous one drain), which
            # stretches every launch's own start-to-end time: the roofline uses launches that run alone, right after
            runs = max(1, min(args.steps, 3))
            avg_ms, launch_frames, st = measure_kernel(tb, W, H, SPP, s, runs)
            launch_timing = ("HIP events around %d path-tracing launches run one at a time after the timed region "
                             "(the timed steps overlap)" % runs)
            if getattr(measure_kernel, "guided", False):
                launch_timing += ("; a launch the caller waits for gets frame groups that shrink over its end (the GUIDED copy of the kernel, option "
                                  "guided_groups): these launches and the committed counters are of that copy, the timed asynchronous steps run "
                                  "the copy with equal groups")
        else:
            avg_ms = float(np.mean(kernel_ms))
            launch_frames = tb.GetOption("last_kernel_frames")
            tb.SetOption("count_rays", 1)
            tb.Render(W, H, 1, s, 0.0)
            st = tb.ReadbackStats().rays
            tb.SetOption("count_rays", 0)
        key = {("cornell-box", 1920, 1080, 64, 8): "c2", ("proc0:870000", 1920, 1080, 128, 6): "c3",
               ("proc0:870000", 1920, 1080, 16, 6): "c3_16spp"}.get((args.scene, W, H, SPP, args.depth))
        # N > 1: the committed counters are of the single-GPU launch of the same workload (a rank's own-tiles launch runs
        # the same kernel on 1/N of the regions); they stay in the line, marked
        passes, src = pmc_summary(key) if (key and args.pipeline == 0) else (None, None)
        pixels = W * H if world == 1 else ts.owned
        hbm = hbm_roofline(avg_ms, launch_frames, pixels, st, passes, src)
        hbm["launch_timing"] = launch_timing
        if world > 1:
            hbm["per_rank"] = {"owned_pixels_rank0": int(ts.owned), "avg_launch_ms_rank0": round(avg_ms, 3),
                               "note": "rank 0's own-tiles launch; counters (traffic, pipes) are per launch of the "
                                       "SINGLE-GPU workload, from " + str(src)}
            if hbm.get("traffic"):   # a per-launch byte count of another launch shape is not this launch's traffic
                hbm["traffic"] = None
                hbm.pop("traffic_GBs", None)
                hbm.pop("traffic_frac_of_peak", None)
        pmc_stale = bool(passes and passes.get("_stale"))
        insts = passes.get("lds", {}).get("SQ_INSTS_VALU") if (passes and not pmc_stale) else None
        if lds:
            # LDS-resident scene: the algorithmic bytes never leave the CU; the resource the kernel can saturate is VALU issue
            roof = {"bound": "valu", "kernel": "pt_persistent", "unit": "Gwave-instr/s", "peak": round(VALU_PEAK_GINST, 1),
                    "peak_model": "%d SIMDs x %.1f GHz / %d cycles per wave64 VALU instruction" % (SIMDS, CLOCK_GHZ, VALU_CYCLES),
                    "avg_launch_ms": round(avg_ms, 3), "launch_timing": launch_timing, "traffic": hbm["traffic"],
                    "pmc_source": src}
            if insts:
                ach = insts / (avg_ms * 1e-3) / 1e9
                roof.update({"achieved": round(ach, 1), "frac": round(ach / VALU_PEAK_GINST, 4),
                             "valu_insts_per_launch": int(insts),
                             "frac_note": "counts every VALU instruction at the 2-cycle rate of v_fma/v_mul/v_add; the "
                                          "kernel's mix (v_pk_fma 3.3, min/max/cndmask 3.2-3.5, v_cmp 4, rcp/sqrt 6.2 "
                                          "cycles: profiles/r2/valu_issue.txt) keeps the VALU pipe busy pipes.valu_busy "
                                          "of the time"})
            else:
                roof.update({"achieved": None, "frac": None,
                             "note": ("the committed PMC passes (%s) were taken of other kernel code (kernel digest "
                                      "differs): re-run scripts/profile_bench.sh" % src) if pmc_stale
                             else "no committed PMC pass for this workload: instruction count unknown"})
            add_pipe_fields(roof, key, passes)
            if world > 1:
                roof["per_rank"] = hbm["per_rank"]
            keys = ("achieved", "unit", "algorithmic_bytes_per_sample", "boxes_per_sample", "tris_per_sample", "rays_per_sample")
            roof["algorithmic"] = {k: hbm[k] for k in keys}
            roof["algorithmic"]["note"] = ("SURVEY 8d byte model; served by the LDS scene image, not HBM (achieved / 8 TB/s "
                                           "= %.2f says nothing about HBM)" % (hbm["achieved"] / HBM_PEAK_GBS))
            result["roofline"] = roof
        else:
            hbm["note"] = "BVH fetched from L2 / Infinity Cache / HBM"
            add_pipe_fields(hbm, key, passes)
            result["roofline"] = hbm

    # ---- further workloads: N = 1 roofline_<leg> (rank 0 is the only rank), N > 1 scale_<leg> (every rank takes part) ----
    legs = [] if (args.no_legs or args.scene != "cornell-box" or args.pipeline != 0) else \
        [x for x in args.legs.split(",") if x in WORKLOADS]
    if world == 1:
        for leg in legs:
            result["roofline_" + leg] = extra_leg(b, np, torch, leg, args.leg_steps)
    else:
        for leg in [x for x in legs if x in SCALE_LEGS]:
            r = scale_leg(b, np, torch, dist, tiles, leg, rank, world, backend, args.leg_steps, barrier)
            result["scale_" + leg] = r
    if legs:   # back to the timed workload for the side measurements below
        tb.SetOption("flatten_instances", 1)
        b.load(args.scene, args.builder, main_opts)
        tb.SetTileAssignment(0, 1)

    if rank == 0:
        if world == 1 and not args.no_readback:
            result["pcie_inclusive"] = pcie_leg(tb, np, W, H, SPP, s, args.steps)
        parity = None
        if not args.no_cpu_baseline:
            if world == 1:
                result["cpu_baseline"], parity = cpu_baseline_leg(tb, args, W, H, SPP, s, np, timed_frame)
            else:
                result["cpu_baseline"] = cpu_baseline_copied()
        if parity is None and not args.no_parity:
            # no CPU-baseline leg in this run (N > 1, or --no-cpu-baseline): the timed frame is gated on strips
            parity = parity_strips(tb, np, timed_frame, W, H, SPP, s, oracle_threads())
        if parity is not None:
            gate(parity, result)
        print(json.dumps(result))
    tb.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
