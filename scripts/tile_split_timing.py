"""Per-rank render time of the multi-GPU tile split, measured on ONE GPU by rendering each rank's tiles in turn:
   python scripts/tile_split_timing.py   (efficiency = full-frame time / world / slowest rank)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import api
tb = api.TracerBoy(0); tb.SetOption("bvh_builder", 1); tb.LoadScene("tests/golden/scenes/cornell-box/scene.pbrt")
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 8
W, H, SPP = 1920, 1080, 64
def t(rank, world, tile=64):
    tb.SetTileAssignment(rank, world, tile, tile)
    tb.Render(W, H, SPP, s, 0.0); tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0)
    return tb.LastRenderMs()
full = t(0, 1)
print("full frame %.2f ms" % full)
for world in (2, 4, 8):
    ts = [t(r, world) for r in range(world)]
    print("world %d: per-rank ms %s  max %.2f  ideal %.2f  efficiency %.0f%%" % (world, " ".join("%.2f" % x for x in ts), max(ts), full / world, 100 * full / world / max(ts)))
