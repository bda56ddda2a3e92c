#!/usr/bin/env python3
"""List registers / LDS / scratch per kernel of one variant translation unit (compile-only, no GPU needed).
   python scripts/kernel_resources.py env        # -> pt_variant_env.hip
Occupancy on gfx950: 512 VGPRs per SIMD lane (arch + acc) -> waves/SIMD = floor(512 / vgprs rounded up to 8), max 8."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "matte"
    src = os.path.join(ROOT, "tracerboy_amd", "csrc", "kernels", "pt_variant_%s.hip" % name)
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
               "--offload-arch=gfx950", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-S", "--cuda-device-only", "-o", out, src]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):]
    for block in meta.split("  - .agpr_count:")[1:]:
        f = {k: v for k, v in re.findall(r"\.(\w+):\s+(\S+)", block)}
        agpr = int(block.split()[0])
        nm = subprocess.run(["c++filt", f["name"]], capture_output=True, text=True).stdout.strip()
        nm = re.sub(r"\(anonymous namespace\)::", "", nm); nm = re.sub(r"\(.*", "", nm).replace("void ", "")
        v = int(f["vgpr_count"]); waves = min(8, 512 // max(8, (v + 7) // 8 * 8))
        print("%-42s vgpr %3d (agpr %3d) sgpr %3d scratch %4s B spills %s  -> %d waves/SIMD by registers" % (
            nm, v, agpr, int(f["sgpr_count"]), f["private_segment_fixed_size"], f.get("vgpr_spill_count", "?"), waves))


if __name__ == "__main__":
    main()
