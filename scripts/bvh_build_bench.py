#!/usr/bin/env python3
"""Times scene load (generation + BVH build + upload) for one builder; run under rocprofv3 --kernel-trace --stats to get the
per-kernel times of the GPU builders (bvh_kernels.hip).   python scripts/bvh_build_bench.py [--scene procK:N] [--builder B]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import api  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="proc2:3000000")
ap.add_argument("--builder", type=int, default=4)
ap.add_argument("--repeat", type=int, default=2)
a = ap.parse_args()
k, n = a.scene[4:].split(":")
tb = api.TracerBoy(0)
tb.SetOption("bvh_builder", a.builder)
for i in range(a.repeat):
    t = time.time(); tb.LoadProcedural(int(k), int(n), 1234); dt = time.time() - t
    print("builder %d %s: load %.3f s, %d triangles, depth %d" % (a.builder, a.scene, dt, tb.SceneInfo().numTriangles, tb.SceneInfo().bvhMaxDepth))
