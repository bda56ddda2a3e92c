"""Frame-group size sweep on one GPU (also emulating each rank of a tile split in turn):
   python scripts/frame_group_sweep.py [scene] [spp] [depth] [worlds, e.g. 1,2,4,8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import api
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 64
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
worlds = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 4, 8]
tb = api.TracerBoy(0); tb.SetOption("bvh_builder", 1)
if scene.startswith("proc"):
    kind, tris = scene[4:].split(":"); tb.LoadProcedural(int(kind), int(tris), 1234)
else:
    tb.LoadScene("tests/golden/scenes/cornell-box/scene.pbrt")
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = depth
W, H = 1920, 1080
def t(rank, world, fg):
    tb.SetTileAssignment(rank, world, 64, 64); tb.SetOption("frame_group", fg)
    tb.Render(W, H, SPP, s, 0.0); tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0)
    return tb.LastRenderMs()
full = t(0, 1, 0)
print("full frame, automatic group size: %.2f ms = %.0f Msamples/s; one pixel per lane: %.2f ms" % (full, W * H * SPP / full / 1e3, t(0, 1, -1)), flush=True)
for world in worlds:
    for fg in (0, 64, 32, 16, 8, 4, 2):
        if fg > SPP: continue
        ts = [t(r, world, fg) for r in range(0, world, max(1, world // 2))]
        print("world %d G=%d: max %.2f ms, %.0f%% of linear" % (world, fg, max(ts), 100 * full / world / max(ts)), flush=True)
