#!/usr/bin/env python3
"""Wall time of the first few tb_render calls after a scene load (one-off costs: code object load, queue scratch, sample buffers)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import api
tb = api.TracerBoy(0); tb.SetOption("bvh_builder", 1)
tb.LoadScene(os.path.join(ROOT, "tests/golden/scenes/cornell-box/scene.pbrt"))
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 8
for i in range(6):
    tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(1920, 1080, 64, s, 0.0); dt = time.perf_counter() - t
    print("call %d: %.2f ms wall, %.2f ms on the stream" % (i, dt * 1e3, tb.LastRenderMs()))
