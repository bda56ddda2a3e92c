#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: Msamples/s (W x H x spp / s) on BASELINE.json
configs[1]: Scenes/cornell-box, 1920x1080, 64 spp, depth 8, persistent-thread HIP on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full render of the workload (W*H*spp samples).  With N > 1 the frame is cut into 64x64 tiles dealt
round-robin to the ranks (tile t -> rank t % N); each rank renders its tiles, packs them, the packed HDR
buffers are gathered to rank 0 over RCCL and rank 0 un-permutes them into the full frame in HBM -- all inside the timed
region.  The total work is fixed, so scaling is "strong".  Scene + BVH are resident in HBM before the timed region
starts; nothing is read from the host inside it.

The JSON line carries two roofline objects:
  "roofline"     the timed kernel on the timed workload.  cornell-box is LDS-resident, so the resource is VALU issue:
                 frac = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x launch time x 2.4 GHz); the instruction count per launch is a
                 property of the workload (same seeds, same control flow) and comes from the committed rocprofv3 PMC pass of this
                 command (profiles/rN/c2_pmc_summary.json), the launch time is measured live with HIP events.
  "roofline_c3"  BASELINE.json configs[2] (870 k-triangle dragon-class scene, 1920x1080 x 128 spp, depth 6) rendered in the
                 same invocation: the HBM roofline SURVEY.md 8d asks for (algorithmic bytes / launch time / 8 TB/s) with the PMC
                 traffic of that launch shape beside it.  Skipped with --no-c3 and at N > 1.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CORNELL = os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box", "scene.pbrt")
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
SIMDS, CLOCK_GHZ, VALU_CYCLES = 1024, 2.4, 2   # 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles (guide + scripts/microbench/valu_issue.hip)
VALU_PEAK_GINST = SIMDS * CLOCK_GHZ / VALU_CYCLES   # 1228.8 G wave-instructions / s
TILE = 64


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--scene", default="cornell-box")  # or proc0:<tris> / proc1:<tris> / proc2:<tris> / path.pbrt
    ap.add_argument("--builder", type=int, default=None)  # default: 1 -- except at --gpus > 1 on procedural scenes, where it is 4 (eight ranks each running the 58-s SAH build on a 16-CPU quota would dominate the run's wall clock; the timed region never sees the build) -- 0 LBVH, 1 binned SAH + reinsertion, 2 LBVH on the GPU, 3 LBVH + treelet passes (the reference's tree), 4 the same on the GPU
    ap.add_argument("--pipeline", type=int, default=0)  # 0 = lock-step bounce (fastest measured), 1 = streaming, 2 = wavefront queues, 3 = pooled
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT")  # extra tb_set_option()s, applied before the scene is loaded
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-readback", action="store_true")   # skip the PCIe-inclusive side measurement
    ap.add_argument("--no-c3", action="store_true")         # skip the further roofline objects (configs[2] at 128 spp, the 4K scenes, Teapot)
    ap.add_argument("--legs", default="c3,c4,c5,teapot")     # which of them the default run renders after its timed region (N = 1, cornell-box, pipeline 0)
    ap.add_argument("--async-steps", action="store_true")  # run the N > 1 step pipeline (async render + pack + stream-ordered consumer) on one GPU
    ap.add_argument("--sync-steps", action="store_true")   # N = 1: wait for every render before enqueuing the next (default: enqueue the K steps, wait once)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0)
    ap.add_argument("--selftest-cpu", action="store_true")  # plumbing test without a GPU: spawn -> gloo rendezvous -> tile gather -> assemble -> one JSON line (tests/)
    a = ap.parse_args(argv)
    if a.builder is None:
        a.builder = 4 if (a.gpus > 1 and a.scene.startswith("proc")) else 1
    return a


# --------------------------------------------------------------------------------------------- self-spawn
def self_spawn(args):
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as fresh processes through torch.distributed.run
    BEFORE this process touches torch or the GPU, relay rank 0's JSON line, exit with the launcher's code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "1")
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
        print("bench.py: the ranks printed no result line", file=sys.stderr); return 1
    return p.returncode


def selftest_cpu(args, rank, world):
    """No GPU: the ranks rendezvous over gloo, every rank packs its tiles of a synthetic frame (numpy restatement of the pack
    kernel), ONE gather moves them to rank 0, rank 0 un-permutes (tb_unpack_gathered_host) and checks the frame.  Exercises the
    launch / relay / collective plumbing of the N > 1 path; measures nothing."""
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
        print(json.dumps({"metric": "Msamples/s (WxHxspp/s)", "value": None, "unit": "Msamples/s", "n_gpus": world, "selftest": "cpu-gloo", "assembled_ok": ok}))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    return 0 if ok else 1


# --------------------------------------------------------------------------------------------- roofline helpers
def byte_model(st):
    """Algorithmic bytes of DESIGN.md section 'Byte model' (layout-A accounting of SURVEY.md 8d, with the
    reference's real 72-B hit-group record): traversal + attribute + material + light + accumulation."""
    return (32 * st.boxesTested + 48 * st.trianglesTested + 180 * st.hitsShaded + 84 * st.materialFetches
            + 104 * st.lightSamples + 32 * st.samples)


def pmc_summary(key):
    """Counters per launch of the timed path-tracing kernel from the newest committed rocprofv3 PMC passes of workload `key`
    (profiles/rN/<key>_pmc_summary.json; separate --pmc runs of this command, scripts/profile_bench.sh).  Returns
    (dict of pass -> counters, file) or (None, None)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", key + "_pmc_summary.json")), key=lambda f: int(re.search(r"profiles/r(\d+)", f).group(1)))
    if not files:
        return None, None
    best = None
    doc = json.load(open(files[-1]))
    from tracerboy_amd import build as tb_build
    stamp = doc.pop("_kernel_digest", None)
    stale = stamp != tb_build.kernel_digest()       # counters of other device code (or unstamped, pre-round-3 files)
    # the primary-visibility pre-pass (pt_primary) runs once before every lock-step launch it feeds: a "launch" of the roofline block is
    # the pair, its counters are the two kernels' sums (GRBM_GUI_ACTIVE too: the kernels run one after the other)
    primary = [v for k, v in doc.items() if k.startswith("pt_primary<") or "::pt_primary<" in k]
    for name, passes in doc.items():
        passes["_stale"] = stale
        # pt_persistent<F, LDS, COUNT, GROUPS[, HYBRID]>: not the counters-on launch (COUNT = true), not the sample fold; the frame-group
        # kernel (GROUPS = true) is the timed one -- its one-pixel-per-lane twin only appears as the zero-frame warm launch
        m = re.search(r"pt_persistent<\d+u, (true|false), (true|false), (true|false)", name)   # <F, LDS, COUNT, GROUPS, ...: the later parameters (split stack, node layout, two-level) do not matter here
        if not m or m.group(2) == "true":
            continue
        if m.group(3) == "true":
            if primary:
                passes["_with_prepass"] = True
                for tag, counters in primary[0].items():
                    if isinstance(counters, dict) and isinstance(passes.get(tag), dict):
                        for c, val in counters.items():
                            if c != "dispatches" and isinstance(val, (int, float)): passes[tag][c] = passes[tag].get(c, 0) + val
            return passes, os.path.relpath(files[-1], ROOT)
        best = passes if best is None else best
    return best, (os.path.relpath(files[-1], ROOT) if best else None)


def derived_busy(key, passes):
    """Pipe utilisations of the timed kernel from the same committed PMC passes, with rocprof's own definitions:
    VALUBusy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * cycles), cycles = GRBM_GUI_ACTIVE / 8 XCDs; TA busy = TA_TA_BUSY_sum /
    (256 TAs * cycles) from profiles/rN/<key>_mem_counters.json (scripts/pmc_mem.sh) when that file exists."""
    import glob
    import re
    out = {}
    if passes and "sq" in passes and "lds" in passes and passes["lds"].get("GRBM_GUI_ACTIVE"):
        cyc = passes["lds"]["GRBM_GUI_ACTIVE"] / 8.0
        out["valu_busy"] = round(4.0 * passes["sq"]["SQ_ACTIVE_INST_VALU"] / (SIMDS * cyc), 3)
        out["salu_busy"] = round(passes["lds"]["SQ_INSTS_SALU"] / (256.0 * cyc), 3)       # one scalar instruction per cycle per CU (valu_issue.hip)
        out["lds_busy"] = round(passes["lds"]["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc), 3)
        out["wait_any"] = round(passes["sq"]["SQ_WAIT_ANY"] / passes["sq"]["SQ_WAVE_CYCLES"], 3)
        if "util" in passes:
            out["valu_lane_utilisation"] = round(passes["util"]["SQ_THREAD_CYCLES_VALU"] / (64.0 * passes["util"]["SQ_ACTIVE_INST_VALU"]), 3)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", key + "_mem_counters.json")), key=lambda f: int(re.search(r"profiles/r(\d+)", f).group(1)))
    if files:
        busy = cyc = 0.0
        for name, c in json.load(open(files[-1])).items():
            m = re.search(r"pt_persistent<\d+u, (true|false), (true|false), (true|false)", name)
            timed = (m and m.group(2) == "false" and m.group(3) == "true") or (passes and passes.get("_with_prepass") and "pt_primary<" in name)
            if timed and isinstance(c, dict) and c.get("TA_TA_BUSY_sum") and c.get("GRBM_GUI_ACTIVE"):
                busy += c["TA_TA_BUSY_sum"]; cyc += c["GRBM_GUI_ACTIVE"]
        if cyc: out["ta_busy"] = round(busy / (256.0 * cyc / 8.0), 3)
    return out


def traffic_bytes(passes):
    """HBM bytes per launch: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- the counters are in KiB and gfx950's FETCH_SIZE reads half the
    bytes of 16-B/lane loads (MI355X_MICROARCH.md, HBM section)."""
    if not passes or "fetch" not in passes or "write" not in passes:
        return None
    return int((2.0 * passes["fetch"]["FETCH_SIZE"] + passes["write"]["WRITE_SIZE"]) * 1024)


def cpu_allowance():
    """What the box lets this process use: the affinity mask, the cgroup CPU quota (v2 cpu.max, v1 cfs_quota / cfs_period; found by
    walking up from this process's cgroup) and the load others put on the machine.  A quota or busy neighbours explain an
    all-threads rate far below threads x single-thread rate; os.sched_getaffinity alone does not show either."""
    out = {"affinity_cpus": len(os.sched_getaffinity(0)), "machine_cpus": os.cpu_count(), "cgroup_quota_cpus": None, "cgroup_source": None}
    try:
        rel = ""
        for line in open("/proc/self/cgroup"):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and (parts[1] == "" or "cpu" in parts[1].split(",")):
                rel = parts[2]
                if parts[1] == "": break
        cands = []
        d = rel
        while True:
            cands.append("/sys/fs/cgroup" + d + "/cpu.max"); cands.append("/sys/fs/cgroup/cpu" + d + "/cpu.cfs_quota_us"); cands.append("/sys/fs/cgroup/cpu,cpuacct" + d + "/cpu.cfs_quota_us")
            if d in ("", "/"): break
            d = os.path.dirname(d)
        best = None
        for f in cands:
            if not os.path.exists(f): continue
            if f.endswith("cpu.max"):
                q, per = open(f).read().split()[:2]
                val = None if q == "max" else float(q) / float(per)
            else:
                q = float(open(f).read()); per = float(open(os.path.join(os.path.dirname(f), "cpu.cfs_period_us")).read())
                val = None if q <= 0 else q / per
            if out["cgroup_source"] is None: out["cgroup_source"] = f + (" (no limit)" if val is None else "")
            if val is not None and (best is None or val < best[0]): best = (val, f)
        if best: out["cgroup_quota_cpus"] = round(best[0], 2); out["cgroup_source"] = best[1]
    except Exception as e:  # noqa: BLE001 -- diagnostics only
        out["cgroup_source"] = "unreadable: %s" % e
    try:
        out["loadavg_1min"] = float(open("/proc/loadavg").read().split()[0])
    except Exception:  # noqa: BLE001
        pass
    return out


def measure_kernel(tb, api, np, W, H, spp, s, runs):
    """HIP-event duration of `runs` path-tracing launches run one at a time + the kernels' own event counters (1-spp launch)."""
    ms = []
    for _ in range(runs):
        tb.InvalidateHistory(); tb.Render(W, H, spp, s, 0.0); ms.append(tb.GetOption("last_kernel_us") / 1e3)
    frames = tb.GetOption("last_kernel_frames")
    tb.SetOption("count_rays", 1)
    tb.Render(W, H, 1, s, 0.0)           # counters-on launch of the same kernels, 1 spp, outside every timed region
    st = tb.ReadbackStats().rays
    tb.SetOption("count_rays", 0); tb.InvalidateHistory()
    return float(np.mean(ms)), frames, st


def hbm_roofline(avg_ms, frames, pixels, st, passes, src):
    bps = byte_model(st) / max(st.samples, 1)
    achieved = bps * pixels * frames / (avg_ms * 1e-3) / 1e9
    traffic = traffic_bytes(passes)
    r = {"bound": "hbm", "kernel": "pt_persistent", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
         "traffic": traffic, "avg_launch_ms": round(avg_ms, 3), "frames_per_launch": int(frames), "algorithmic_bytes_per_sample": round(bps, 1),
         "boxes_per_sample": round(st.boxesTested / max(st.samples, 1), 2), "tris_per_sample": round(st.trianglesTested / max(st.samples, 1), 2),
         "rays_per_sample": round(st.rays / max(st.samples, 1), 3), "pmc_source": src}
    if traffic:
        r["traffic_GBs"] = round(traffic / (avg_ms * 1e-3) / 1e9, 1); r["traffic_frac_of_peak"] = round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return r


def expected_speedup(scene, W, H, spp, depth, world):
    """What the tile split should give at this N, from the one-GPU stand-in measurement of a rank's own-tiles render, pack and un-permute
    (scripts/gather_standin.py -> profiles/rN/gather_standin.json) plus one xGMI hop of its packed tiles; render k + 1 overlaps gather k, so a
    step is the render and a fraction of a millisecond of exposed tail.  The driver computes the measured efficiency from its own per-N runs."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "gather_standin.json")), key=lambda f: int(re.search(r"profiles/r(\d+)", f).group(1)))
    if not files or (scene, W, H, spp, depth) != ("cornell-box", 1920, 1080, 64, 8): return None
    d = json.load(open(files[-1]))
    one, mine, hop = d.get("c2_rank0_of_1"), d.get("c2_rank0_of_%d" % world), d.get("1080p_world%d" % world)
    if not one or not mine: return None
    step = mine["render_ms"] + 0.1       # exposed tail of the pipelined step (pack + un-permute + what the gather does not hide)
    out = {"vs_1gpu": round(one["render_ms"] / step, 2), "render_ms_per_rank": mine["render_ms"], "gather_hop_us": hop["xgmi_hop_estimate_us"] if hop else None,
           "source": os.path.relpath(files[-1], ROOT), "note": "one GPU emulating rank 0 of N; no multi-GPU hardware was available to the build"}
    # the same step as the timed region runs it (launches overlapping, pack and a stream-ordered consumer behind each render), K steps back to
    # back on one GPU as rank 0 of N (scripts/rank_share_async.py): what is left per step when the links keep up
    piped = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "rank_share_async.json")), key=lambda f: int(re.search(r"profiles/r(\d+)", f).group(1)))
    if piped:
        a = json.load(open(piped[-1]))
        if a.get("world1") and a.get("world%d" % world):
            out["pipelined"] = {"vs_1gpu": round(a["world1"]["ms_per_step"] / a["world%d" % world]["ms_per_step"], 2), "ms_per_step_per_rank": a["world%d" % world]["ms_per_step"],
                                "source": os.path.relpath(piped[-1], ROOT)}
    return out


TEAPOT = os.path.join(ROOT, "tests", "golden", "scenes", "Teapot", "scene.pbrt")
EXTRA_LEGS = {   # key -> (scene, builder, W, H, spp, depth): the configurations the 8 GPUs divide, and the reference's own textured scene
    "c4": ("proc1:700000", 4, 3840, 2160, 8, 6),      # BASELINE configs[3] class: 0.7 M triangles with glass, 4K (8 of its 256 spp per step)
    "c5": ("proc2:2980000", 4, 3840, 2160, 8, 16),    # BASELINE configs[4] class: 2.98 M triangles, 40 materials, depth 16 (8 of its 1024 spp per step)
    "teapot": (TEAPOT, 1, 1920, 1080, 16, 8),         # /root/reference/Scenes/Teapot as committed under tests/golden: 126 k triangles, textures + env + GGX
}


def extra_leg(tb, api, np, torch, load, key, steps=3):
    """One more workload under the driver's clock: loaded, warmed, `steps` renders enqueued back to back and waited for once (like the
    timed region), then launches run one at a time for the roofline figures; counters from profiles/rN/<key>_{pmc_summary,mem_counters}.json."""
    scene, builder, W, H, SPP, D = EXTRA_LEGS[key]
    s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = D
    tb.SetOption("bvh_builder", builder)
    load_s = load(scene)
    info = tb.SceneInfo()
    for _ in range(2):                                        # warm-up: first launch of this kernel copy, both sample buffers / side streams
        tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0)
    # the library finds out by itself whether back-to-back calls of this kind should share the chip (two launches in flight) or take
    # turns -- it needs a few rounds of asynchronous calls to see device-bound intervals both ways (renderImpl, overlap trial)
    for _ in range(4):
        if tb.GetOption("overlap_trial_phase") == 2: break
        for _ in range(5):
            tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0, sync=False)
        tb.Sync()
        if tb.GetOption("last_variant") in (0, 1): break      # matte / env: overlapped launches always pay, nothing is tried
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0, sync=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    variant = ["matte", "env", "surf", "vol", "full", "sss"][tb.GetOption("last_variant")]
    prepass = bool(tb.GetOption("last_primary_prepass")); overlapped = bool(tb.GetOption("last_overlap"))
    avg, frames, st = measure_kernel(tb, api, np, W, H, SPP, s, steps)
    passes, src = pmc_summary(key)
    r = hbm_roofline(avg, frames, W * H, st, passes, src)
    r.update({"workload": "%s %dx%d %dspp depth%d" % (os.path.basename(os.path.dirname(scene)) if scene.endswith(".pbrt") else scene, W, H, SPP, D), "triangles": int(info.numTriangles),
              "value": round(W * H * SPP * steps / dt / 1e6, 1), "unit_value": "Msamples/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "scene_load_s": round(load_s, 2),
              "bvh_builder": ("lbvh", "sah", "lbvh-gpu", "lbvh+treelets", "lbvh+treelets-gpu")[builder], "kernel_variant": variant, "primary_prepass": prepass, "launches_overlap": overlapped,
              "kernel": "pt_primary + pt_persistent" if prepass else "pt_persistent", "pipes": derived_busy(key, passes), "pmc_stale": bool(passes and passes.get("_stale")),
              "algorithmic": {"achieved": None, "unit": "GB/s", "peak": HBM_PEAK_GBS, "note": "SURVEY 8d byte model x samples / launch time; served mostly by L2 / Infinity Cache (traffic_GBs is what the fabric carries)"}})
    r["algorithmic"]["achieved"] = r["achieved"]; r["algorithmic"]["frac"] = r["frac"]
    pipes = r["pipes"]
    if pipes.get("ta_busy") is not None:   # like roofline_c3: the busier of the two issue pipes the counters show
        busiest = max(("vmem_issue", pipes["ta_busy"]), ("valu", pipes.get("valu_busy", 0.0)), key=lambda kv: kv[1])
        r.update({"bound": busiest[0], "frac": busiest[1], "achieved": busiest[1], "peak": 1.0, "unit": "busy fraction of the launch (TA_TA_BUSY / 256 TAs, or 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs)"})
    return r


def main():
    args = parse_args()
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
    # Test hooks for boxes with fewer GPUs than ranks (tests/test_gpu_parity.py): TB_BENCH_SHARE_DEVICE=1 puts every rank on device 0,
    # TB_BENCH_BACKEND=gloo moves the gather through host memory (RCCL refuses two ranks on one device).  Everything else of the
    # N > 1 step -- own-tiles launch, pack, one gather per render, device-side un-permute, the assembled-frame check -- is the real code.
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

    W, H, SPP = args.width, args.height, args.spp
    s = api.GetDefaultOutputSettings()
    s.EnableBlueNoise = 0       # SURVEY.md 8d "Common": pure rand() path, Time = 0, NEE on, RIS off, box filter
    s.MaxBounces = args.depth
    tb = api.TracerBoy(local_rank)
    tb.SetOption("bvh_builder", args.builder)
    tb.SetOption("pipeline", args.pipeline)
    for kv in args.opt:
        k, v = kv.split("="); tb.SetOption(k, int(v))

    def load(scene):
        t0 = time.time()
        if scene == "cornell-box":
            tb.LoadScene(CORNELL)
        elif scene.startswith("proc"):
            kind, tris = scene[4:].split(":")
            tb.LoadProcedural(int(kind), int(tris), 1234)
        else:
            tb.LoadScene(scene)
        return time.time() - t0

    load_s = load(args.scene)
    info = tb.SceneInfo()
    tb.SetTileAssignment(rank, world, TILE, TILE)
    owned = tb.OwnedPixels(W, H)
    # equal-sized slices: every rank pads to the largest owner (rank 0) so ONE gather per render suffices.  Two packed buffers: the
    # gather of one render runs on RCCL's stream while the next render traces, and a buffer is packed again only after the gather
    # that read it (two renders ago) has finished.  Rank 0 gathers into ONE contiguous world x capacity buffer (views per rank)
    # and un-permutes it on the device into the full frame (tb_unpack_gathered_device), ordered behind the gather.
    capacity = max(tiles.packed_capacity(W, H, world, TILE, TILE), 1)
    packed = [torch.zeros((capacity, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    gathered = torch.zeros((world, capacity, 4), dtype=torch.float32, device="cuda") if (world > 1 and rank == 0) else None
    gather_list = [gathered[r] for r in range(world)] if gathered is not None else None
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") if gathered is not None else None
    scratch = torch.zeros_like(packed[0]) if (world == 1 and args.async_steps) else None
    in_flight = [None, None]
    renders = [0]
    torch.cuda.synchronize()

    kernel_ms = []

    # N > 1 (and --async-steps): nothing in a step blocks the host -- the render and the pack are enqueued on the library's
    # stream, the gather on RCCL's, the un-permute on torch's stream behind the gather, ordered by stream waits -- so host-side
    # launch gaps do not idle the GPU between renders
    pipelined = world > 1 or args.async_steps
    lib_stream = torch.cuda.ExternalStream(tb.Stream()) if pipelined else None

    def exchange(buf):
        if world > 1 and backend != "nccl":      # test hook: the same collective through host memory, synchronously
            torch.cuda.current_stream().synchronize()
            host = buf.cpu()
            parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, parts, dst=0)
            if rank == 0:
                gathered.copy_(torch.stack(parts))
                tb.UnpackGatheredTo(gathered.data_ptr(), capacity, W, H, world, TILE, TILE, frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            return None
        if world > 1:
            work = dist.gather(buf, gather_list if rank == 0 else None, dst=0, async_op=True)
            if rank == 0:
                work.wait()       # torch's current stream waits for the gather (the host does not) ...
                tb.UnpackGatheredTo(gathered.data_ptr(), capacity, W, H, world, TILE, TILE, frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)   # ... and the un-permute runs behind it
            return work
        scratch.copy_(buf, non_blocking=True)   # --async-steps on one GPU: a stand-in consumer on torch's stream
        return None

    def step():
        tb.InvalidateHistory()
        if not pipelined:
            if args.sync_steps:
                tb.Render(W, H, SPP, s, 0.0)          # synchronous; GPU time measured with HIP events on the library's stream
                kernel_ms.append(tb.GetOption("last_kernel_us") / 1e3)  # the render's (first) path-tracing launch, without the sample fold
            else:
                # the K renders are enqueued back to back (tb_render_async) and waited for once by the closing barrier
                # (torch.cuda.synchronize = device-wide), inside the timed region: the launch of step k+1 starts on the other
                # side stream while the last paths of step k drain
                tb.Render(W, H, SPP, s, 0.0, sync=False)
            return
        b = renders[0] & 1; renders[0] += 1
        tb.Render(W, H, SPP, s, 0.0, sync=False)
        if in_flight[b] is not None:
            in_flight[b].wait()                                   # torch's stream waits for the gather that last read packed[b] ...
        lib_stream.wait_stream(torch.cuda.current_stream())     # ... and the library's stream waits for torch's
        tb.PackOwnedTo(packed[b].data_ptr(), sync=False)
        torch.cuda.current_stream().wait_stream(lib_stream)     # the gather reads what the library's stream packed
        in_flight[b] = exchange(packed[b])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
        tb.Sync(); kernel_ms.append(tb.GetOption("last_kernel_us") / 1e3)   # HIP events of the last render of the timed region
    if world > 1:
        t = torch.tensor([elapsed], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())

    samples_per_step = W * H * SPP
    value = samples_per_step * args.steps / elapsed / 1e6
    result = {
        "metric": "Msamples/s (WxHxspp/s)", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s %dx%d %dspp depth%d" % (args.scene, W, H, SPP, args.depth), "triangles": int(info.numTriangles),
                   "bvh_builder": ("lbvh", "sah", "lbvh-gpu", "lbvh+treelets", "lbvh+treelets-gpu")[args.builder], "pipeline": ("lockstep", "stream", "wavefront", "pooled", "split")[tb.GetOption("last_pipeline")], "tile": TILE if world > 1 else None,
                   "parallelism": "tiles%d" % world, "scene_in_lds": bool(tb.GetOption("scene_in_lds_active")),
                   "kernel_variant": ["matte", "env", "surf", "vol", "full", "sss"][tb.GetOption("last_variant")], "scene_load_s": round(load_s, 3)},
    }

    if world > 1:
        tl = torch.tensor([load_s], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64); dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        result["config"]["scene_load_s"] = round(float(tl.item()), 3); result["config"]["scene_load_s_note"] = "max over ranks (every rank loads and builds the scene itself)"
        # ---- where a step's time goes, per rank (after the timed region, stages run one at a time with a device sync between them, so
        #      the figures are each stage's own cost, not its share of the overlapped pipeline): this rank's own-tiles render, the
        #      device-side pack of its tiles, the gather to rank 0 (RCCL over xGMI; every rank's buffer is `capacity` pixels) and rank 0's
        #      device-side un-permute.  Per stage the MAX over ranks (the slowest rank is what a step waits for) and the mean.
        stages = {"render_ms": [], "pack_ms": [], "gather_ms": [], "unpack_ms": []}
        for _ in range(3):
            barrier()
            t = time.perf_counter(); tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0); stages["render_ms"].append((time.perf_counter() - t) * 1e3)
            t = time.perf_counter(); tb.PackOwnedTo(packed[0].data_ptr(), sync=False); tb.Sync(); stages["pack_ms"].append((time.perf_counter() - t) * 1e3)
            barrier()
            t = time.perf_counter()
            if backend == "nccl":
                dist.gather(packed[0], gather_list if rank == 0 else None, dst=0); torch.cuda.synchronize()
            else:
                host = packed[0].cpu(); parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, parts, dst=0)
                if rank == 0: gathered.copy_(torch.stack(parts)); torch.cuda.synchronize()
            stages["gather_ms"].append((time.perf_counter() - t) * 1e3)
            t = time.perf_counter()
            if rank == 0:
                tb.UnpackGatheredTo(gathered.data_ptr(), capacity, W, H, world, TILE, TILE, frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
            stages["unpack_ms"].append((time.perf_counter() - t) * 1e3)
        mine = torch.tensor([min(v) for v in stages.values()], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")   # best of 3 per stage
        mx = mine.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = mine.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        result["scale_breakdown"] = {k: round(float(mx[i]), 3) for i, k in enumerate(stages)}
        result["scale_breakdown"].update({"mean_over_ranks": {k: round(float(sm[i]) / world, 3) for i, k in enumerate(stages)},
                                          "gather_bytes_per_rank": int(capacity) * 16, "owned_pixels_rank0": int(owned),
                                          "note": "stages run one at a time after the timed region (best of 3, max over ranks); in the timed steps render k+1 overlaps gather k"})
        exp = expected_speedup(args.scene, W, H, SPP, args.depth, world)
        if exp: result["expected_speedup"] = exp
        result["rccl_ranks"] = int(dist.get_world_size())
        result["collective_backend"] = backend
        # after the timed region: the frame rank 0 assembled from the gathered tiles equals a single-GPU render of the whole frame
        if rank == 0:
            assembled = frame.cpu().numpy()
            tb.SetTileAssignment(0, 1); tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0)
            whole = tb.ReadAccumulation()
            tb.SetTileAssignment(rank, world, TILE, TILE)
            result["config"]["assembled_frame_equals_single_gpu"] = bool(np.array_equal(assembled.view(np.uint32), whole.view(np.uint32)))
        dist.barrier()

    if rank == 0:
        lds = bool(tb.GetOption("scene_in_lds_active"))
        # ---- roofline of the dominant (only) kernel: pt_persistent ------------------------------------
        launch_timing = "HIP events around the path-tracing launch of the last timed step (tb_last_render_ms)"
        if world == 1 and not args.sync_steps and not args.async_steps:
            # the timed steps overlap (the next launch starts while the last paths of the previous one drain), which stretches
            # every launch's own start-to-end time: the roofline uses launches that run alone, right after the timed region
            runs = max(1, min(args.steps, 3))
            avg_ms, launch_frames, st = measure_kernel(tb, api, np, W, H, SPP, s, runs)
            launch_timing = "HIP events around %d path-tracing launches run one at a time after the timed region (the timed steps overlap)" % runs
        else:
            avg_ms = float(np.mean(kernel_ms)); launch_frames = tb.GetOption("last_kernel_frames")
            tb.SetOption("count_rays", 1); tb.Render(W, H, 1, s, 0.0); st = tb.ReadbackStats().rays; tb.SetOption("count_rays", 0)
        key = {("cornell-box", 1920, 1080, 64, 8): "c2", ("proc0:870000", 1920, 1080, 128, 6): "c3", ("proc0:870000", 1920, 1080, 16, 6): "c3_16spp"}.get((args.scene, W, H, SPP, args.depth))
        # N > 1: the committed counters are of the single-GPU launch of the same workload (a rank's own-tiles launch runs the same kernel on
        # 1/N of the regions); they stay in the line, marked, so that the N = 1 and N > 1 records carry the same fields
        passes, src = pmc_summary(key) if (key and args.pipeline == 0) else (None, None)
        pixels = W * H if world == 1 else owned
        hbm = hbm_roofline(avg_ms, launch_frames, pixels, st, passes, src)
        hbm["launch_timing"] = launch_timing
        if world > 1:
            hbm["per_rank"] = {"owned_pixels_rank0": int(owned), "avg_launch_ms_rank0": round(avg_ms, 3), "note": "rank 0's own-tiles launch; counters (traffic, pipes) are per launch of the SINGLE-GPU workload, from " + str(src)}
            if hbm.get("traffic"): hbm["traffic"] = None; hbm.pop("traffic_GBs", None); hbm.pop("traffic_frac_of_peak", None)   # a per-launch byte count of another launch shape is not this launch's traffic
        insts = passes.get("lds", {}).get("SQ_INSTS_VALU") if passes else None
        pmc_stale = bool(passes and passes.get("_stale"))
        if pmc_stale: insts = None      # an instruction count of other code says nothing about this build's issue rate
        if lds:
            # LDS-resident scene: the algorithmic bytes never leave the CU; the resource the kernel can saturate is VALU issue
            roof = {"bound": "valu", "kernel": "pt_persistent", "unit": "Gwave-instr/s", "peak": round(VALU_PEAK_GINST, 1),
                    "peak_model": "%d SIMDs x %.1f GHz / %d cycles per wave64 VALU instruction" % (SIMDS, CLOCK_GHZ, VALU_CYCLES),
                    "avg_launch_ms": round(avg_ms, 3), "launch_timing": launch_timing, "traffic": hbm["traffic"], "pmc_source": src}
            if insts:
                ach = insts / (avg_ms * 1e-3) / 1e9
                roof.update({"achieved": round(ach, 1), "frac": round(ach / VALU_PEAK_GINST, 4), "valu_insts_per_launch": int(insts),
                             "frac_note": "counts every VALU instruction at the 2-cycle rate of v_fma/v_mul/v_add; the kernel's mix (v_pk_fma 3.3, min/max/cndmask 3.2-3.5, "
                                          "v_cmp 4, f64 3.1-3.7, rcp/sqrt 6.2 cycles: profiles/r2/valu_issue.txt) keeps the VALU pipe busy pipes.valu_busy of the time"})
            else:
                roof.update({"achieved": None, "frac": None, "note": ("the committed PMC passes (%s) were taken of other kernel code (kernel digest differs): re-run scripts/profile_bench.sh" % src) if pmc_stale
                             else "no committed PMC pass for this workload: instruction count unknown"})
            roof["pipes"] = derived_busy(key, passes) if key else {}
            if world > 1: roof["per_rank"] = hbm["per_rank"]
            roof["pmc_stale"] = pmc_stale
            roof["algorithmic"] = {k: hbm[k] for k in ("achieved", "unit", "algorithmic_bytes_per_sample", "boxes_per_sample", "tris_per_sample", "rays_per_sample")}
            roof["algorithmic"]["note"] = "SURVEY 8d byte model; served by the LDS scene image, not HBM (achieved / 8 TB/s = %.2f says nothing about HBM)" % (hbm["achieved"] / HBM_PEAK_GBS)
            result["roofline"] = roof
        else:
            hbm["note"] = "BVH fetched from L2 / Infinity Cache / HBM"
            if key: hbm["pipes"] = derived_busy(key, passes)
            result["roofline"] = hbm

        # ---- second object: configs[2] at its full 128 spp, the workload whose roofline IS HBM ---------
        if world == 1 and not args.no_c3 and args.scene == "cornell-box" and args.pipeline == 0 and "c3" in args.legs.split(","):
            W3, H3, SPP3, D3 = 1920, 1080, 128, 6
            s3 = api.GetDefaultOutputSettings(); s3.EnableBlueNoise = 0; s3.MaxBounces = D3
            load3 = load("proc0:870000")
            info3 = tb.SceneInfo()
            tb.InvalidateHistory(); tb.Render(W3, H3, SPP3, s3, 0.0)      # warm-up (first launch of this kernel copy)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            for _ in range(3):
                tb.InvalidateHistory(); tb.Render(W3, H3, SPP3, s3, 0.0, sync=False)
            torch.cuda.synchronize(); dt3 = time.perf_counter() - t3
            variant3 = ["matte", "env", "surf", "vol", "full", "sss"][tb.GetOption("last_variant")]
            prepass3 = bool(tb.GetOption("last_primary_prepass"))   # of the timed renders above (measure_kernel ends with a counting launch, which never has it)
            avg3, frames3, st3 = measure_kernel(tb, api, np, W3, H3, SPP3, s3, 3)
            passes3, src3 = pmc_summary("c3")
            r3 = hbm_roofline(avg3, frames3, W3 * H3, st3, passes3, src3)
            r3.update({"workload": "proc0:870000 %dx%d %dspp depth%d" % (W3, H3, SPP3, D3), "triangles": int(info3.numTriangles), "value": round(W3 * H3 * SPP3 * 3 / dt3 / 1e6, 1),
                       "unit_value": "Msamples/s", "ms_per_step": round(dt3 / 3 * 1e3, 3), "steps": 3, "scene_load_s": round(load3, 2),
                       "kernel_variant": variant3,
                       "note": "launches of 128 frames are batched by the sample-buffer budget: avg_launch_ms / frames_per_launch are per batch launch"})
            r3["primary_prepass"] = prepass3
            if r3["primary_prepass"]:
                r3["kernel"] = "pt_primary + pt_persistent"
                r3["note"] += ("; a launch is the pair primary-visibility pre-pass (pt_primary: every camera ray of the batch, one 8x8 pixel tile per wave) + lock-step kernel "
                               "(takes the first hits from the sample slots): avg_launch_ms spans both, the committed counters are their sums")
            r3["pipes"] = derived_busy("c3", passes3)
            r3["pmc_stale"] = bool(passes3 and passes3.get("_stale"))   # true: the committed counters were taken of other kernel code
            # What the committed counters say limits it is not the fabric (traffic_frac_of_peak) but the issue of vector-memory and vector-ALU
            # instructions at ~19 of 64 lanes: the texture addresser is pipes.ta_busy busy, the VALU pipes.valu_busy.
            # `frac` is therefore the busier of the two pipes; the SURVEY 8d algorithmic rate / 8 TB/s stays under `algorithmic`.
            r3["algorithmic"] = {"achieved": r3["achieved"], "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": r3["frac"],
                                 "note": "SURVEY 8d byte model x samples / launch time; the bytes are served mostly by L2 / Infinity Cache (traffic_GBs is what the fabric carries)"}
            pipes3 = r3["pipes"]
            if pipes3.get("ta_busy") is not None:
                busiest = max(("vmem_issue", pipes3["ta_busy"]), ("valu", pipes3.get("valu_busy", 0.0)), key=lambda kv: kv[1])
                r3.update({"bound": busiest[0], "frac": busiest[1], "achieved": busiest[1], "peak": 1.0, "unit": "busy fraction of the launch (TA_TA_BUSY / 256 TAs, or 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs)"})
            r3["what_limits_it"] = ("instruction issue at ~19 of 64 lanes: a CU's texture addresser takes ~17 cycles per wave-level load whatever the number of active lanes "
                                    "(scripts/microbench/gather64.hip) and every VALU instruction of the walk pays for 64 lanes; halving the node loads (layout C) moves "
                                    "the time by 1-2 % because the conversions it adds fill the VALU instead (profiles/r3/pmcab_c3.json, DESIGN.md section 6)")
            # the same workload through the compact nodes (option node_layout = 1: 32-B nodes on a 16-bit grid, within 1e-4 rel. L2 of the bit-exact path)
            tb.SetOption("node_layout", 1)
            tb.InvalidateHistory(); tb.Render(W3, H3, SPP3, s3, 0.0)
            if tb.GetOption("last_node_layout") == 1:
                torch.cuda.synchronize(); t3c = time.perf_counter()
                for _ in range(3):
                    tb.InvalidateHistory(); tb.Render(W3, H3, SPP3, s3, 0.0, sync=False)
                torch.cuda.synchronize(); dt3c = time.perf_counter() - t3c
                ms3c = []
                for _ in range(3):
                    tb.InvalidateHistory(); tb.Render(W3, H3, SPP3, s3, 0.0); ms3c.append(tb.GetOption("last_kernel_us") / 1e3)
                r3["compact_nodes"] = {"value": round(W3 * H3 * SPP3 * 3 / dt3c / 1e6, 1), "unit_value": "Msamples/s", "ms_per_step": round(dt3c / 3 * 1e3, 3),
                                       "avg_launch_ms": round(float(np.mean(ms3c)), 3), "option": "node_layout=1",
                                       "contract": "relative L2 <= 1e-4 against the bit-exact layout-B path (tests/test_compact_nodes.py), not bit equality"}
            tb.SetOption("node_layout", 0)
            result["roofline_c3"] = r3
            tb.SetOption("bvh_builder", args.builder); load(args.scene)   # back to the timed workload for the CPU baseline below
        # ---- the configurations the 8 GPUs divide (4K glass scenes) and the reference's own Teapot, each a few steps under the driver's clock
        if world == 1 and not args.no_c3 and args.scene == "cornell-box" and args.pipeline == 0:
            for leg in [x for x in args.legs.split(",") if x in EXTRA_LEGS]:
                result["roofline_" + leg] = extra_leg(tb, api, np, torch, load, leg)
            tb.SetOption("bvh_builder", args.builder); load(args.scene)

        # ---- CPU baseline: the scalar oracle on a bounded sample of the same workload ------------------
        # ---- the same render with the frame handed to the host (tb_read_accum: one D2H copy of the RGBA32F sums into a host
        #      array the caller owns, pageable memory as a ctypes/numpy caller has it).  Reported beside `value`, never as it.
        if world == 1 and not args.no_readback:
            import numpy as np
            host = np.zeros((H, W, 4), np.float32)
            hp = host.ctypes.data_as(__import__("ctypes").c_void_p)
            tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0); tb._check(tb._L.tb_read_accum(tb._ctx, hp, None))
            n_rb = max(2, min(args.steps, 10)); t_rb = time.perf_counter()
            for _ in range(n_rb):
                tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0); tb._check(tb._L.tb_read_accum(tb._ctx, hp, None))
            t_rb = (time.perf_counter() - t_rb) / n_rb
            result["pcie_inclusive"] = {"value": round(samples_per_step / t_rb / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(t_rb * 1e3, 3),
                                        "readback_bytes": int(host.nbytes), "note": "render + tb_read_accum into pageable host memory, synchronous, %d steps" % n_rb}

        if not args.no_cpu_baseline:   # rank 0 (the other ranks wait at the closing barrier): the same leg at every N, so that the records carry the same fields
            import oracle_lib as ol
            allowance = cpu_allowance()
            cores = len(os.sched_getaffinity(0))      # the CPUs this process may run on, not the machine's ...
            if allowance.get("cgroup_quota_cpus"):    # ... and not more threads than the cgroup lets run at once: 256 threads on a 16-CPU quota
                cores = max(1, min(cores, int(allowance["cgroup_quota_cpus"] + 0.999)))   # time-slice each other to 8.8x; 16 threads reach the quota
            view = tb.HostSceneView(); pf = tb.FrameConstants(W, H, 0, s, 0.0)
            # probe the all-threads rate on one full 1-spp frame, size the sample (whole frames) to ~cpu_baseline_seconds; the
            # single-thread figure is measured on whole frames too (the same rows), sized to about the same time
            t1 = time.perf_counter(); ol.render(view, pf, W, H, 1, threads=cores); dt = time.perf_counter() - t1
            frames = int(max(1, min(SPP, args.cpu_baseline_seconds / max(dt, 1e-3))))
            c0 = os.times(); t1 = time.perf_counter(); ol.render(view, pf, W, H, frames, threads=cores); dt = time.perf_counter() - t1; c1 = os.times()
            cpu_seconds = (c1.user - c0.user) + (c1.system - c0.system)    # CPU time the threads actually got: cpu_seconds / dt = CPUs' worth of service
            if cores == 1:
                dt1, n1, rows1 = dt, W * H * frames, "the same run"
            else:
                est1 = dt * cores / max(frames, 1)     # estimated single-thread seconds per whole frame
                f1 = int(max(1, min(frames, args.cpu_baseline_seconds / max(est1, 1e-3))))
                if est1 <= 3.0 * args.cpu_baseline_seconds:
                    t2 = time.perf_counter(); ol.render(view, pf, W, H, f1, threads=1); dt1 = time.perf_counter() - t2; n1 = W * H * f1; rows1 = "whole frame x %d spp" % f1
                else:                                    # a whole frame on one thread would take minutes: every k-th 8-row strip of the frame instead
                    k = int(est1 / args.cpu_baseline_seconds) + 1
                    strips = list(range(0, H, 8 * k)); t2 = time.perf_counter()
                    for y0 in strips: ol.render(view, pf, W, H, 1, y0=y0, y1=min(H, y0 + 8), threads=1)
                    dt1 = time.perf_counter() - t2; n1 = sum(W * (min(H, y0 + 8) - y0) for y0 in strips); rows1 = "every %d-th 8-row strip of the frame x 1 spp" % k
            result["cpu_baseline"] = {"value": round(W * H * frames / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                                      "sample": "scalar C++ oracle (oracle/tb_oracle.cpp, g++ -O2), the %dx%d frame x %d spp of %d, depth %d, %d threads over 8-row strips (%.1f s)"
                                                % (W, H, frames, SPP, args.depth, cores, dt),
                                      "single_thread": round(n1 / dt1 / 1e6, 4), "single_thread_sample": rows1 + " (%.1f s)" % dt1,
                                      "machine_cpus": os.cpu_count()}
            cb = result["cpu_baseline"]
            cb["box"] = allowance
            cb["effective_parallelism"] = round(cb["value"] / max(cb["single_thread"], 1e-9), 1)        # all-threads rate / single-thread rate
            cb["cpus_worth_of_service"] = round(cpu_seconds / max(dt, 1e-9), 1)                        # process CPU time / wall time of the all-threads run
            cb["unthrottled_estimate"] = {"value": round(cb["single_thread"] * len(os.sched_getaffinity(0)), 2), "unit": "Msamples/s",
                                          "note": "single-thread rate x the %d hardware threads of the box: what the same port would reach there without a quota" % len(os.sched_getaffinity(0))}
            cb["sample"] = cb["sample"].replace("over 8-row strips", "over row strips (>= 4 work items per thread)")
            cb["note"] = ("the all-threads figure is what THIS lease delivers: effective_parallelism is well below the thread count when the box throttles "
                          "(cgroup quota in box.cgroup_quota_cpus) or shares its cores (cpus_worth_of_service << threads); quote speed-ups against both figures")
            if args.scene == "cornell-box":
                # BASELINE.json configs[0], the reference's own CPU-runnable case, timed exactly: 512x512, 4 spp, depth 4
                import copy
                s0 = copy.copy(s); s0.MaxBounces = 4
                pf0 = tb.FrameConstants(512, 512, 0, s0, 0.0)
                t3 = time.perf_counter(); ol.render(view, pf0, 512, 512, 4, threads=cores); dt3 = time.perf_counter() - t3
                t4 = time.perf_counter(); ol.render(view, pf0, 512, 512, 4, threads=1); dt4 = time.perf_counter() - t4
                result["cpu_baseline"]["configs0"] = {"workload": "cornell-box 512x512 4spp depth4", "all_threads_s": round(dt3, 4), "single_thread_s": round(dt4, 3),
                                                      "all_threads": round(512 * 512 * 4 / dt3 / 1e6, 3), "single_thread": round(512 * 512 * 4 / dt4 / 1e6, 4)}
        print(json.dumps(result))
    tb.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
