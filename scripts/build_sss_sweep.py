#!/usr/bin/env python3
"""Build copies of libtracerboy_hip.so that differ only in the occupancy the `sss` feature set's second kernel copy is held to
(TB_SSS_WAVES = 3..6 waves per SIMD; pt_variant_sss4.hip and the host's kVariants table read the macro):
tracerboy_amd/_sweep/libtracerboy_hip_sss<W>.so.  Runs here (hipcc cross-compiles); scripts/sss_occupancy_sweep.sh swaps them in on
the GPU box.  Reuses the other objects of the normal build (python -m tracerboy_amd.build first)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import build as b

b.build(verbose=False)
out_dir = os.path.join(b.ROOT, "_sweep"); os.makedirs(out_dir, exist_ok=True)
srcs = b.HOST_SRCS + b.KERNEL_SRCS
objs = {s: os.path.join(b.OBJ, s.replace("/", "_") + ".o") for s in srcs}
for w in ([int(x) for x in sys.argv[1:]] or [3, 4, 5, 6]):
    mine = dict(objs)
    for s in ("kernels/pt_variant_sss4.hip", "host/context.cpp"):
        o = os.path.join(out_dir, "w%d_%s.o" % (w, s.replace("/", "_")))
        cmd = [b.HIPCC] + b.COMMON + ["-DTB_SSS_WAVES=%d" % w] + (b.DEVICE if s.endswith(".hip") else ["-x", "hip", "--offload-arch=" + b.ARCH]) + ["-c", os.path.join(b.CSRC, s), "-o", o]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        mine[s] = o
    lib = os.path.join(out_dir, "libtracerboy_hip_sss%d.so" % w)
    subprocess.run([b.HIPCC, "-shared", "-fPIC", "--offload-arch=" + b.ARCH, "-o", lib] + [mine[s] for s in srcs], check=True)
    print("built", lib)
