import copy, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tracerboy_amd import api
tb = api.TracerBoy()
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 6
tb.SetOption("bvh_builder", 1); tb.LoadProcedural(0, 870000, 1234); tb.SetOption("bvh_builder", 0)
W, H, F = 1920, 1080, 128
for pre in (1, 0):
    tb.SetOption("primary_prepass", 2 * pre)
    for g in (0, 2, 4, 8, 16, 32):
        tb.SetOption("frame_group", g); ts = []
        for r in range(4):
            tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, F, s, 0.0); ts.append(time.perf_counter() - t)
        print("c3 128 spp (sah) prepass", pre, "frame_group", g if g else "auto", "%.1f Msamples/s" % (W * H * F / np.median(ts[1:]) / 1e6), flush=True)
