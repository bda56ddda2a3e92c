#!/usr/bin/env python3
"""Stress: small frame-group renders that cycle through three random streams (time seed 0 / 1 / 2; the two sample buffers alternate), so that a sample slot nobody wrote
shows as a mismatch with the oracle.  argv: reps [primary_prepass option]"""
import copy, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib as ol
from tracerboy_amd import api
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tb = api.TracerBoy()
try: tb.SetOption("primary_prepass", pre)
except api.TracerBoyError: print("no primary_prepass option in this build")
for kv in os.environ.get("TB_STRESS_OPTIONS", "").split():
    k, v = kv.split("="); tb.SetOption(k, int(v)); print("option", k, v)
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 16
W, H, F = 200, 120, 9
def bits(a): return np.ascontiguousarray(a).view(np.uint32)
total = 0
for kind, tris, seed in ((1, 30000, 7), (2, 40000, 9), (0, 30000, 5)):
    tb.LoadProcedural(kind, tris, seed)
    refs = [ol.render(tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, float(t)), W, H, F, threads=8)["output"] for t in (0, 1, 2)]   # three streams against two alternating buffers: a buffer's previous content always differs
    bad = 0
    for rep in range(reps):
        t = rep % 3
        tb.InvalidateHistory(); tb.Render(W, H, F, s, float(t))
        out = tb.ReadAccumulation()
        if not np.array_equal(bits(out), bits(refs[t])):
            bad += 1
            wrong = (bits(out) != bits(refs[t])).any(-1)
            if bad <= 3 and tb.GetOption("debug_fg_samples_ptr"):
                # which of the launch's samples are wrong in the device's sample buffer itself?  (per-frame oracle renders are the samples)
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so"); buf = np.zeros((F, H, W, 4), np.float32)
                hip.hipMemcpy(ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(tb.GetOption("debug_fg_samples_ptr")), ctypes.c_size_t(buf.nbytes), 2)
                per = np.stack([ol.render(tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, float(t)), W, H, 1, first_frame=f, threads=8)["output"] for f in range(F)])
                if F > 1: per[1:] -= 0  # (each call starts from a zero accumulator: the frame's own sample)
                sb = (np.abs(buf) .view(np.uint32) != np.abs(per).view(np.uint32)).any(-1)
                prev = np.stack([ol.render(tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, float((t + 1) % 3)), W, H, 1, first_frame=f, threads=8)["output"] for f in range(F)])
                stale = (np.abs(buf).view(np.uint32) == np.abs(prev).view(np.uint32)).all(-1) & sb
                lanes = sorted(set((int(y) % 16) * 16 + int(x) % 16 for _, y, x in zip(*np.nonzero(sb))))
                print("   of the wrong samples, equal to what the buffer held two renders ago (never written):", int(stale.sum()), "; positions in the region (y*16+x), first 24:", lanes[:24], flush=True)
                cap = tb.GetOption("debug_slot_log_cap")
                if cap:
                    log = np.zeros((4096, cap), np.uint64); hip.hipMemcpy(ctypes.c_void_p(log.ctypes.data), ctypes.c_void_p(tb.GetOption("debug_slot_log_ptr")), ctypes.c_size_t(log.nbytes), 2)
                    ys, xs = np.nonzero(sb.any(0)); ry, rx = int(ys[0]) // 16, int(xs[0]) // 16; fr = int(np.nonzero(sb)[0][0])
                    rows = [(r, sl) for r in range(4096) for sl in range(cap) if log[r, sl] and not (int(log[r, sl]) >> 39) & 1 and (int(log[r, sl]) & 0xfff) == rx and ((int(log[r, sl]) >> 12) & 0xfff) == ry and ((int(log[r, sl]) >> 24) & 0x7fff) == fr]
                    print("   the lost item (region y %d x %d, frame %d) was bound by (workgroup, slot):" % (ry, rx, fr), rows, flush=True)
                    for r, sl in rows[:2]:
                        print("      workgroup %d bound:" % r, [("slot %d" % k, "nothing left" if (int(log[r, k]) >> 39) & 1 else "y%d x%d f%d" % ((int(log[r, k]) >> 12) & 0xfff, int(log[r, k]) & 0xfff, (int(log[r, k]) >> 24) & 0x7fff)) for k in range(cap) if log[r, k]], flush=True)
                    used = int((log[:, 0] != 0).sum()); print("      workgroups with a bound slot 0:", used, flush=True)
                print("   sample buffer: wrong samples", int(sb.sum()), "in frames", sorted(set(np.nonzero(sb)[0].tolist())), "pixels wrong in buffer AND picture", int((sb.any(0) & wrong).sum()), "picture only", int((~sb.any(0) & wrong).sum()), flush=True)
            if bad <= 3:
                ys, xs = np.nonzero(wrong); reg = {}
                for y, x in zip(ys // 16, xs // 16): reg[(int(y), int(x))] = reg.get((int(y), int(x)), 0) + 1
                print("   regions (y, x): count", sorted(reg.items())[:10], flush=True)
            if bad <= 3: print("kind", kind, "rep", rep, "wrong pixels", int(wrong.sum()), "equal to the other stream's picture there:", bool(np.array_equal(bits(out)[wrong], bits(refs[(t + 1) % 3])[wrong])), flush=True)
    print("kind", kind, "bad", bad, "of", reps, flush=True); total += bad
print("total bad", total)
