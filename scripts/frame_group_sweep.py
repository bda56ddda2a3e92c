import os, sys
sys.path.insert(0, os.getcwd())
from tracerboy_amd import api
tb = api.TracerBoy(0); tb.SetOption("bvh_builder", 1); tb.LoadScene("tests/golden/scenes/cornell-box/scene.pbrt")
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 8
W, H, SPP = 1920, 1080, 64
def t(rank, world, fg):
    tb.SetTileAssignment(rank, world, 64, 64); tb.SetOption("frame_group", fg)
    tb.Render(W, H, SPP, s, 0.0); tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0)
    return tb.LastRenderMs()
full = t(0, 1, 0)
print("full frame classic %.2f ms" % full, flush=True)
for fg in (64, 32, 16, 8, 4, 2):
    print("world 1 G=%d: %.2f ms" % (fg, t(0, 1, fg)), flush=True)
for world in (2, 4, 8):
    for fg in (64, 32, 16, 8, 4):
        ts = [t(r, world, fg) for r in range(0, world, max(1, world // 2))]
        print("world %d G=%d: max %.2f ideal %.2f eff %.0f%%" % (world, fg, max(ts), full / world, 100 * full / world / max(ts)), flush=True)
