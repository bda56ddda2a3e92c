#!/usr/bin/env python3
"""Times the real-time chain (tb_render_realtime + tb_post_process) per displayed frame.
   python scripts/realtime_bench.py [--width 1920 --height 1080 --frames 30 --depth 3]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--frames", type=int, default=30)
ap.add_argument("--depth", type=int, default=3)
a = ap.parse_args()
tb = api.TracerBoy(0)
tb.SetOption("bvh_builder", 1)
tb.LoadScene(os.path.join(ROOT, "tests/golden/scenes/cornell-box/scene.pbrt"))
s = api.GetDefaultOutputSettings(); s.MaxBounces = a.depth
dn = api.GetDefaultDenoiserSettings()
ps = api.GetDefaultPostProcessSettings()
for _ in range(3):
    tb.RenderRealTime(a.width, a.height, s, dn, 0.0)
t0 = time.perf_counter()
for _ in range(a.frames):
    tb.RenderRealTime(a.width, a.height, s, dn, 0.0)
t1 = time.perf_counter()
tb.PostProcess(ps)
print("real-time chain %dx%d depth %d: %.2f ms per displayed frame (%.1f fps), path tracing alone %.2f ms" % (
    a.width, a.height, a.depth, (t1 - t0) / a.frames * 1e3, a.frames / (t1 - t0), tb.LastRenderMs()))
