#!/usr/bin/env python3
"""Build an experimental copy of libtracerboy_hip.so that differs from the tree's in a few translation units compiled with extra
flags: tracerboy_amd/_sweep/libtracerboy_hip_<TAG>.so (git-ignored; travels to the GPU box with the snapshot; TB_LIB=<path> selects
it, tracerboy_amd/api.py).  The other objects are the normal build's (python -m tracerboy_amd.build runs first).

   python scripts/build_variant.py TAG --flags "-DTB_SSS_WAVES=2" --tus kernels/pt_variant_sss4.hip host/context.cpp
   python scripts/build_variant.py TAG --flags "-DFOO" --tus kernels        # every kernel TU
Runs here (hipcc cross-compiles gfx950 without a GPU)."""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import build as b  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--flags", default="")
ap.add_argument("--tus", nargs="+", required=True)
ap.add_argument("--default-scheduler", action="store_true")   # drop the build's -amdgpu-sched-strategy (LLVM's own default)
a = ap.parse_args()

b.build(verbose=False)
out_dir = os.path.join(b.ROOT, "_sweep")
os.makedirs(out_dir, exist_ok=True)
srcs = b.HOST_SRCS + b.KERNEL_SRCS
objs = {s: os.path.join(b.OBJ, s.replace("/", "_") + ".o") for s in srcs}
want = []
for t in a.tus:
    want += [s for s in srcs if s == t or s.startswith(t + "/") or (t == "kernels" and s.startswith("kernels/"))]
if not want:
    raise SystemExit("no translation unit matches %s" % a.tus)


def compile_one(s):
    o = os.path.join(out_dir, "%s_%s.o" % (a.tag, s.replace("/", "_")))
    dev = list(b.device_flags(s)) if s.endswith(".hip") else ["-x", "hip", "--offload-arch=" + b.ARCH]
    if "-amdgpu-sched-strategy=" in a.flags or a.default_scheduler:   # an -mllvm option may occur once: the variant's scheduler replaces the build's
        dev = [f for i, f in enumerate(dev) if not f.startswith("-amdgpu-sched-strategy=") and not
               (f == "-mllvm" and i + 1 < len(dev) and dev[i + 1].startswith("-amdgpu-sched-strategy="))]
    cmd = [b.HIPCC] + b.COMMON + a.flags.split() + dev + ["-c", os.path.join(b.CSRC, s), "-o", o]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("compile failed: %s\n%s" % (" ".join(cmd), r.stderr[-3000:]))
    return s, o


with cf.ThreadPoolExecutor(max_workers=6) as ex:
    for s, o in ex.map(compile_one, sorted(set(want))):
        objs[s] = o
lib = os.path.join(out_dir, "libtracerboy_hip_%s.so" % a.tag)
subprocess.run([b.HIPCC, "-shared", "-fPIC", "--offload-arch=" + b.ARCH, "-o", lib] + [objs[s] for s in srcs], check=True)
print("built", lib)
