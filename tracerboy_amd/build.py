"""Build libtracerboy_hip.so (hipcc, gfx950) in-tree.

    python -m tracerboy_amd.build [--force]

Every translation unit is compiled with -ffp-contract=off: the fp32 arithmetic contract of
include/tb_math.h (bit-identical results on host and device) depends on it.  Kernel variants are
separate TUs so they compile in parallel.
"""
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(ROOT)
CSRC = os.path.join(ROOT, "csrc")
OBJ = os.path.join(ROOT, "_build")
LIB = os.path.join(ROOT, "libtracerboy_hip.so")
CLI = os.path.join(ROOT, "tracerboy-hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = os.environ.get("TB_ARCH", "gfx950")  # e.g. gfx950:xnack- for experiments

COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
          "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-I" + os.path.join(REPO, "include")] + os.environ.get("TB_EXTRA_FLAGS", "").split()
# Kernel TUs: no SLP vectorisation (the packed fp32 math is written out where it pays -- v_pk_fma_f32 in the box test; what the SLP
# pass adds on top are v_pk_add / v_pk_mul pairs with v_mov shuffles around them, at 3.3 cycles against 2 x 2.0) and the max-ILP
# machine scheduler.  Measured on MI355X (scripts/ab_flags.sh): cornell-box 6 894 -> 7 054, 870 k scene 4 106 -> 4 247, bistro-class 4K
# 1 171 -> 1 213 Msamples/s; neither changes a result bit (-ffp-contract=off pins the arithmetic, tests/ -m gpu).
DEVICE = ["--offload-arch=" + ARCH, "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
# The machine scheduler is chosen PER TRANSLATION UNIT where a same-box A/B says so (round 5, scripts/ab_variants.sh,
# profiles/r5/ab_sched*.json; bit-identical pictures either way -- the scheduler only orders instructions).  max-memory-clause groups a
# block's loads instead of spreading them for ILP: fewer values live at once, so the copies held to an occupancy spill less --
#   vol4 (vw-van, 4 waves per SIMD): spill stores per launch 89 M -> 61 M, 4K flattened 1 845 -> 1 958 Msamples/s (+6 %), two-level +3.7 %;
#   surf (Teapot): +1.6 % at 3 waves per SIMD, and with it 4 waves beat 3 for the first time (2 913 -> 3 116, +7 %: pt_variant_surf.hip);
# and where it loses it stays off: sss4 (van- / bistro-class 4K) -3 % / -4 % although its spill stores fall 17 %, env5 (870 k scene) -12 %,
# matte5 (cornell-box) +-0.  iterative-minreg: spill stores -10 % ... -34 %, every scene 10-13 % slower.
SCHED_MEMORY_CLAUSE = ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]
TU_SCHEDULER = {"kernels/pt_variant_vol4.hip": SCHED_MEMORY_CLAUSE, "kernels/pt_variant_surf.hip": SCHED_MEMORY_CLAUSE}


def device_flags(src):
    """DEVICE with the translation unit's own scheduler in place of the default one (an -mllvm option may occur once)."""
    if src not in TU_SCHEDULER:
        return DEVICE
    base = [f for i, f in enumerate(DEVICE) if not f.startswith("-amdgpu-sched-strategy=")
            and not (f == "-mllvm" and i + 1 < len(DEVICE) and DEVICE[i + 1].startswith("-amdgpu-sched-strategy="))]
    return base + TU_SCHEDULER[src]

HOST_SRCS = ["host/pbrt_loader.cpp", "host/pbf_loader.cpp", "host/host_scene.cpp", "host/images.cpp", "host/image_decode.cpp", "host/image_formats.cpp", "host/bvh_build.cpp", "host/procedural.cpp", "host/context.cpp", "host/context_scene.cpp", "host/context_render.cpp", "host/pbrt_dump.cpp"]
KERNEL_SRCS = ["kernels/pt_kernels.hip", "kernels/post_kernels.hip", "kernels/bvh_kernels.hip", "kernels/rt_kernels.hip", "kernels/pt_variant_matte.hip", "kernels/pt_variant_matte5.hip", "kernels/pt_variant_env.hip", "kernels/pt_variant_env5.hip", "kernels/pt_variant_surf.hip",
               "kernels/pt_variant_sss.hip", "kernels/pt_variant_sss4.hip",
               "kernels/pt_variant_vol.hip", "kernels/pt_variant_vol4.hip", "kernels/pt_variant_full.hip",
               "kernels/pt_split_matte.hip", "kernels/pt_split_env.hip", "kernels/pt_split_surf.hip", "kernels/pt_split_sss.hip"]


def _deps_digest():
    h = hashlib.sha1()
    for base in (CSRC, os.path.join(REPO, "include")):
        for dp, _, fs in sorted(os.walk(base)):
            for f in sorted(fs):
                if f.endswith((".h", ".hpp", ".inc", ".cpp", ".hip")):
                    with open(os.path.join(dp, f), "rb") as fh:
                        h.update(f.encode()); h.update(fh.read())
    # (not the include path: the checkout sits elsewhere on the GPU box, and a stamp that names it made every process there rebuild the library)
    h.update(" ".join([f for f in COMMON + DEVICE if not f.startswith("-I")] + [k + " ".join(v) for k, v in sorted(TU_SCHEDULER.items())]).encode())
    return h.hexdigest()


def kernel_digest():
    """Digest of what decides the device code: the kernel sources, the shared headers and the device flags.  scripts/profile_bench.sh
    stamps the PMC summaries it writes with it; bench.py refuses to derive a roofline fraction from counters taken of other code."""
    h = hashlib.sha1()
    for base in (os.path.join(CSRC, "kernels"), os.path.join(REPO, "include")):
        for dp, _, fs in sorted(os.walk(base)):
            for f in sorted(fs):
                if f == "tracerboy_hip.h":      # the C ABI: no kernel includes it (a comment edit there used to mark every committed counter stale)
                    continue
                if f.endswith((".h", ".hpp", ".inc", ".hip")):
                    with open(os.path.join(dp, f), "rb") as fh:
                        h.update(f.encode()); h.update(fh.read())
    flags = COMMON + DEVICE + [k + " ".join(v) for k, v in sorted(TU_SCHEDULER.items())]
    h.update(" ".join([f for f in flags if not f.startswith("-I")]).encode())   # not the include path: the checkout sits elsewhere on the GPU box
    return h.hexdigest()[:16]


def _compile(src):
    obj = os.path.join(OBJ, src.replace("/", "_") + ".o")
    cmd = [HIPCC] + COMMON + (device_flags(src) if src.endswith(".hip") else ["-x", "hip", "--offload-arch=" + ARCH]) + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("compile failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "stamp")
    digest = _deps_digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == digest:
        return LIB
    srcs = HOST_SRCS + KERNEL_SRCS
    with cf.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    cli_src = os.path.join(CSRC, "host", "cli.cpp")
    if os.path.exists(cli_src):
        cmd = [HIPCC] + COMMON + ["-x", "hip", "--offload-arch=" + ARCH, cli_src, "-o", CLI, "-L" + ROOT, "-ltracerboy_hip", "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("cli build failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(digest)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
