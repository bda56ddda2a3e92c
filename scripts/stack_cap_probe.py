import sys, os, time, json
sys.path.insert(0, os.getcwd())
from tracerboy_amd import api
tb = api.TracerBoy(0)
for key, scene, builder, W, H, F, D in (("c4 lbvh", "proc1:700000", 4, 3840, 2160, 8, 6), ("c5 lbvh", "proc2:2980000", 4, 3840, 2160, 8, 16)):
    tb.SetOption("bvh_builder", builder); k, n = scene[4:].split(":"); tb.LoadProcedural(int(k), int(n), 1234)
    s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = D
    for cap in (0, 24, 20, 16, 12, 8):
        tb.SetOption("stack_lds_cap", cap)
        for _ in range(2): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
        for _ in range(3):
            for _ in range(5): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
            tb.Sync()
        best = 0
        for _ in range(3):
            t = time.perf_counter()
            for _ in range(6): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
            tb.Sync(); best = max(best, W * H * F * 6 / (time.perf_counter() - t) / 1e6)
        print(json.dumps({"workload": key, "stack_lds_cap": cap, "msamples": round(best, 1), "overflow_entries": tb.GetOption("last_plan_stack_overflow"), "variant": tb.GetOption("last_variant"), "overlap": tb.GetOption("last_overlap")}), flush=True)
    tb.SetOption("stack_lds_cap", 0)
