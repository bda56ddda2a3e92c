import copy, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tracerboy_amd import api
tb = api.TracerBoy()
s0 = api.GetDefaultOutputSettings(); s0.EnableBlueNoise = 0
for name, proc, depth in (("c4-class", (1, 700000, 1234), 6), ("c5-class", (2, 2980000, 1234), 16)):
    s = copy.copy(s0); s.MaxBounces = depth
    tb.SetOption("bvh_builder", 4); tb.LoadProcedural(*proc); tb.SetOption("bvh_builder", 0)
    W, H, F = 3840, 2160, 32
    tb.SetOption("primary_prepass", 2)
    for g in (0, 2, 4, 8, 16):
        tb.SetOption("frame_group", g); ts = []
        for r in range(3):
            tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, F, s, 0.0); ts.append(time.perf_counter() - t)
        print(name, "4K x 32 spp, pre-pass asked for, frame_group", g if g else "auto", "%.1f Msamples/s" % (W * H * F / np.median(ts[1:]) / 1e6), flush=True)
    tb.SetOption("frame_group", 0)
