#!/usr/bin/env python3
"""One GPU as rank 0 of N, running the step bench.py runs at N > 1 -- asynchronous render of the rank's own tiles, device-side pack,
a stream-ordered consumer of the packed tiles (a device copy stands in for the xGMI gather) -- with nothing blocking the host, K steps
enqueued back to back.  gather_standin.py times the stages one at a time; this is what the overlapped pipeline leaves per step, the
figure the 8-GPU line of the driver should approach when the links keep up.  Usage: rank_share_async.py [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tracerboy_amd import api, tiles  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W, H, SPP, TILE = 1920, 1080, 64, 64
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 8
tb = api.TracerBoy(0); tb.SetOption("bvh_builder", 1)
tb.LoadScene(os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box", "scene.pbrt"))
lib_stream = torch.cuda.ExternalStream(tb.Stream())
res = {}
for world in (1, 2, 4, 8):
    tb.SetTileAssignment(0, world, TILE, TILE)
    cap = max(tiles.packed_capacity(W, H, world, TILE, TILE), 1)
    packed = [torch.zeros((cap, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    sink = torch.zeros_like(packed[0])
    n = [0]

    def step():
        b = n[0] & 1; n[0] += 1
        tb.InvalidateHistory()
        tb.Render(W, H, SPP, s, 0.0, sync=False)
        lib_stream.wait_stream(torch.cuda.current_stream())
        tb.PackOwnedTo(packed[b].data_ptr(), sync=False)
        torch.cuda.current_stream().wait_stream(lib_stream)
        sink.copy_(packed[b], non_blocking=True)

    for _ in range(6): step()
    tb.Sync(); torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(K): step()
        tb.Sync(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / K * 1e3
        best = ms if best is None else min(best, ms)
    res["world%d" % world] = {"ms_per_step": round(best, 3), "frame_group": tb.GetOption("last_plan_frame_group"), "overlap": tb.GetOption("last_overlap")}
tb.SetTileAssignment(0, 1)
one = res["world1"]["ms_per_step"]
for world in (2, 4, 8):
    res["world%d" % world]["speedup_if_links_keep_up"] = round(one / res["world%d" % world]["ms_per_step"], 2)
print(json.dumps(res))
