import sys, os, time, json
sys.path.insert(0, os.getcwd())
import bench
from tracerboy_amd import api
b = bench.Bench(api, 0); tb = b.tb
for leg in sys.argv[1].split(","):
    w = bench.WORKLOADS[leg]; W, H, SPP = w["W"], w["H"], w["spp"]; s = b.settings(w["depth"]); b.load_workload(leg)
    tb.SetOption("overlap_launches", 2)
    for parts in (1, 2, 4):
        tb.SetOption("pooled_samples", W * H * max(1, SPP // parts) if parts > 1 else 256 << 20)
        for _ in range(3): tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0)
        best = 1e9
        for _ in range(5):
            tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, SPP, s, 0.0); best = min(best, time.perf_counter() - t)
        print(json.dumps({"leg": leg, "batches": parts, "sync_call_ms": round(best * 1e3, 3), "frames_per_launch": int(tb.GetOption("last_kernel_frames")), "G": int(tb.GetOption("last_plan_frame_group")), "guided": int(tb.GetOption("last_plan_guided_groups"))}), flush=True)
    tb.SetOption("pooled_samples", 256 << 20); tb.SetOption("overlap_launches", 1)
