#!/usr/bin/env python3
"""sha1 of the layout-A BVH image(s) the host builds for a few scenes (builder 1, 0 and 1 reinsertion passes): a change to the builder that claims
to build THE SAME trees (e.g. the parallel top-down pass) prints the same lines.   TB_LIB=<other library> python scripts/tree_digest.py"""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import api  # noqa: E402
G = os.path.join(ROOT, "tests", "golden", "scenes")
cases = [("cornell", dict(path=os.path.join(G, "cornell-box", "scene.pbrt"))), ("teapot", dict(path=os.path.join(G, "Teapot", "scene.pbrt"))),
         ("proc0:300000", dict(procedural=(0, 300000, 1234))), ("proc1:150000", dict(procedural=(1, 150000, 7))), ("proc2:400000", dict(procedural=(2, 400000, 9))),
         ("vwvan", dict(path=os.path.join(G, "vw-van", "vw-van.pbrt"))), ("vwvan 2-level", dict(path=os.path.join(G, "vw-van", "vw-van.pbrt"), flatten_instances=False))]
for passes in ("0", "1"):
    for name, kw in cases:
        t = time.time(); hs = api.HostScene(bvh_builder=1, reinsertion_passes=int(passes), **kw); dt = time.time() - t; v = hs.view()
        h = hashlib.sha1(C.string_at(v.bvh, v.bvhBytes))
        if v.tlas: h.update(C.string_at(v.tlas, v.tlasBytes))
        print("%-14s passes %s  %s  %9d B  %.2f s" % (name, passes, h.hexdigest()[:16], v.bvhBytes, dt), flush=True)
