#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: Msamples/s (W x H x spp / s) on BASELINE.json
configs[1]: Scenes/cornell-box, 1920x1080, 64 spp, depth 8, wavefront HIP on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full render of the workload (W*H*spp samples).  With N > 1 the frame is cut into
64x64 tiles dealt round-robin to the ranks (tile t -> rank t % N); each rank renders its
tiles, packs them and the packed HDR buffers are gathered to rank 0 over RCCL inside the timed region.
The total work is fixed, so scaling is "strong".  Scene + BVH are resident in HBM before the timed
region starts; nothing is read from the host inside it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CORNELL = os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box", "scene.pbrt")
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def byte_model(st):
    """Algorithmic bytes of DESIGN.md section 'Byte model' (layout-A accounting of SURVEY.md 8d, with the
    reference's real 72-B hit-group record): traversal + attribute + material + light + accumulation."""
    return (32 * st.boxesTested + 48 * st.trianglesTested + 180 * st.hitsShaded + 84 * st.materialFetches
            + 104 * st.lightSamples + 32 * st.samples)


def pmc_traffic_bytes(args, world):
    """HBM bytes per launch of the timed kernel from the committed rocprofv3 PMC passes of this exact command
    (profiles/<round>/<key>_pmc_summary.json, separate --pmc FETCH_SIZE / WRITE_SIZE runs, scripts/profile_r1.sh):
    (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- the counters are in KiB and gfx950's FETCH_SIZE reads half the bytes of
    16-B/lane loads (MI355X_MICROARCH.md, HBM).  None when no profile of this workload is committed."""
    key = {("cornell-box", 1920, 1080, 64, 8): "c2", ("proc0:870000", 1920, 1080, 16, 6): "c3"}.get(
        (args.scene, args.width, args.height, args.spp, args.depth))
    if key is None or world != 1 or args.pipeline != 0:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", key + "_pmc_summary.json")))
    if not files:
        return None
    import re
    best = None
    for name, passes in json.load(open(files[-1])).items():
        # pt_persistent<F, LDS, COUNT, GROUPS>: not the counters-on launch (COUNT = true), not the sample fold; the frame-group
        # kernel (GROUPS = true) is the timed one -- its one-pixel-per-lane twin only appears as the zero-frame warm launch
        m = re.search(r"pt_persistent<\d+u, (true|false), (true|false), (true|false)(?:, (?:true|false))?>", name)  # <F, LDS, COUNT, GROUPS[, HYBRID]>
        if not m or m.group(2) == "true" or "fetch" not in passes or "write" not in passes:
            continue
        traffic = int((2.0 * passes["fetch"]["FETCH_SIZE"] + passes["write"]["WRITE_SIZE"]) * 1024)
        if m.group(3) == "true":
            return traffic
        best = traffic if best is None else best
    return best
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--scene", default="cornell-box")  # or proc0:<tris> / proc1:<tris> / proc2:<tris> / path.pbrt
    ap.add_argument("--builder", type=int, default=1)  # 0 LBVH, 1 binned SAH + reinsertion, 2 LBVH on the GPU, 3 LBVH + treelet passes (the reference's tree), 4 the same on the GPU
    ap.add_argument("--pipeline", type=int, default=0)  # 0 = lock-step bounce (fastest measured), 1 = streaming (resumable BVH walk)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT")  # extra tb_set_option()s, applied before the scene is loaded
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--async-steps", action="store_true")  # run the N > 1 step pipeline (async render + pack + stream-ordered consumer) on one GPU
    ap.add_argument("--sync-steps", action="store_true")   # N = 1: wait for every render before enqueuing the next (default: enqueue the K steps, wait once)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from tracerboy_amd import api

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed launcher (WORLD_SIZE is 1)" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path is HIP-only (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    W, H, SPP = args.width, args.height, args.spp
    s = api.GetDefaultOutputSettings()
    s.EnableBlueNoise = 0       # SURVEY.md 8d "Common": pure rand() path, Time = 0, NEE on, RIS off, box filter
    s.MaxBounces = args.depth
    tb = api.TracerBoy(local_rank)
    tb.SetOption("bvh_builder", args.builder)
    tb.SetOption("pipeline", args.pipeline)
    for kv in args.opt:
        k, v = kv.split("="); tb.SetOption(k, int(v))
    t0 = time.time()
    if args.scene == "cornell-box":
        tb.LoadScene(CORNELL)
    elif args.scene.startswith("proc"):
        kind, tris = args.scene[4:].split(":")
        tb.LoadProcedural(int(kind), int(tris), 1234)
    else:
        tb.LoadScene(args.scene)
    load_s = time.time() - t0
    info = tb.SceneInfo()
    TILE = 64
    tb.SetTileAssignment(rank, world, TILE, TILE)
    from tracerboy_amd import tiles
    owned = tb.OwnedPixels(W, H)
    # equal-sized slices: every rank pads to the largest owner (rank 0) so ONE gather per render suffices
    # two packed buffers: the gather of one render runs on RCCL's stream while the next render traces, and a buffer is packed
    # again only after the gather that read it (two renders ago) has finished
    packed = [torch.zeros((max(tiles.packed_capacity(W, H, world, TILE, TILE), 1), 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    gather_list = [torch.zeros_like(packed[0]) for _ in range(world)] if (world > 1 and rank == 0) else None
    scratch = torch.zeros_like(packed[0]) if (world == 1 and args.async_steps) else None
    in_flight = [None, None]
    renders = [0]
    torch.cuda.synchronize()

    kernel_ms = []

    # N > 1 (and --async-steps): nothing in a step blocks the host -- the render and the pack are enqueued on the library's
    # stream, the gather on RCCL's, ordered by stream waits -- so host-side launch gaps do not idle the GPU between renders
    pipelined = world > 1 or args.async_steps
    lib_stream = torch.cuda.ExternalStream(tb.Stream()) if pipelined else None

    def exchange(buf):
        if world > 1:
            return dist.gather(buf, gather_list if rank == 0 else None, dst=0, async_op=True)
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
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())

    samples_per_step = W * H * SPP
    value = samples_per_step * args.steps / elapsed / 1e6
    result = {
        "metric": "Msamples/s (WxHxspp/s)", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s %dx%d %dspp depth%d" % (args.scene, W, H, SPP, args.depth), "triangles": int(info.numTriangles),
                   "bvh_builder": ("lbvh", "sah", "lbvh-gpu", "lbvh+treelets", "lbvh+treelets-gpu")[args.builder], "pipeline": ("lockstep", "stream", "wavefront", "pooled")[args.pipeline], "tile": TILE if world > 1 else None,
                   "parallelism": "tiles%d" % world, "scene_in_lds": bool(tb.GetOption("scene_in_lds_active")),
                   "kernel_variant": ["matte", "env", "surf", "vol", "full"][tb.GetOption("last_variant")], "scene_load_s": round(load_s, 3)},
    }

    if rank == 0:
        # ---- roofline of the dominant (only) kernel: pt_persistent ------------------------------------
        launch_timing = "HIP events around the path-tracing launch of every timed step (tb_last_render_ms)"
        if world == 1 and not args.sync_steps and not args.async_steps:
            # the timed steps overlap (the next launch starts while the last paths of the previous one drain), which stretches
            # every launch's own start-to-end time: the roofline uses launches that run alone, right after the timed region
            kernel_ms.clear()
            for _ in range(max(1, min(args.steps, 3))):
                tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0); kernel_ms.append(tb.GetOption("last_kernel_us") / 1e3)
            launch_timing = "HIP events around %d path-tracing launches run one at a time after the timed region (the timed steps overlap)" % len(kernel_ms)
        avg_ms = float(np.mean(kernel_ms)); launch_frames = tb.GetOption("last_kernel_frames")
        tb.SetOption("count_rays", 1)
        tb.Render(W, H, 1, s, 0.0)           # counters-on launch of the same kernels, 1 spp, outside the timed region
        st = tb.ReadbackStats().rays
        tb.SetOption("count_rays", 0)
        bytes_per_sample = byte_model(st) / max(st.samples, 1)
        samples_per_launch = owned_samples = (W * H if world == 1 else owned) * launch_frames
        achieved = bytes_per_sample * samples_per_launch / (avg_ms * 1e-3) / 1e9
        result["roofline"] = {
            "bound": "hbm", "kernel": ("pt_persistent", "pt_stream", "wf_* (all stages)", "pt_pooled")[args.pipeline], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic_bytes(args, world),
            "avg_launch_ms": round(avg_ms, 3), "launch_timing": launch_timing, "algorithmic_bytes_per_sample": round(bytes_per_sample, 1),
            "boxes_per_sample": round(st.boxesTested / max(st.samples, 1), 2), "tris_per_sample": round(st.trianglesTested / max(st.samples, 1), 2),
            "rays_per_sample": round(st.rays / max(st.samples, 1), 3),
            "note": ("scene image is LDS-resident: algorithmic bytes are served by LDS, HBM only sees the sample buffer / accumulation surfaces"
                     if tb.GetOption("scene_in_lds_active") else "BVH fetched from L2/MALL/HBM"),
        }
        # ---- CPU baseline: the scalar oracle on a bounded sample of the same workload ------------------
        if not args.no_cpu_baseline:
            import oracle_lib as ol
            cores = os.cpu_count() or 1
            view = tb.HostSceneView(); pf = tb.FrameConstants(W, H, 0, s, 0.0)
            # probe the rate on one full 1-spp frame, then size the sample (whole frames) to ~cpu_baseline_seconds
            t1 = time.perf_counter(); ol.render(view, pf, W, H, 1, threads=cores); dt = time.perf_counter() - t1
            frames = int(max(1, min(SPP, args.cpu_baseline_seconds / max(dt, 1e-3))))
            t1 = time.perf_counter(); ol.render(view, pf, W, H, frames, threads=cores); dt = time.perf_counter() - t1
            rows1 = 8 * max(1, H // 8 // 32)
            t2 = time.perf_counter(); ol.render(view, pf, W, H, 1, y0=(H - rows1) // 2, y1=(H - rows1) // 2 + rows1, threads=1); dt1 = time.perf_counter() - t2
            result["cpu_baseline"] = {"value": round(W * H * frames / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                                      "sample": "scalar C++ oracle (oracle/tb_oracle.cpp, g++ -O2), the %dx%d frame x %d spp of %d, depth %d, %d threads over 8-row strips (%.1f s)"
                                                % (W, H, frames, SPP, args.depth, cores, dt),
                                      "single_thread": round(W * rows1 / dt1 / 1e6, 4)}
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
