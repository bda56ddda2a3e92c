#!/usr/bin/env python3
"""Static half of the instruction mix of a lock-step kernel: VALU instructions of ONE kernel of a variant TU, bucketed by
the phase of the path loop they belong to.  CPU only (hipcc cross-compiles gfx950).

The TU is compiled with the build's flags plus -g; every instruction address of the kernel is handed to llvm-symbolizer --inlines, whose
inline chain says which step function (path_begin, traverse, path_on_closest, path_scatter ...) the instruction was inlined from and which
primitive (tb_sin, box_test2, tri_test ...) it belongs to.  Instructions of the kernel body itself are split by the source line ranges of
pt_persistent.inc (sample end / next sample / loop).  IEEE divisions are counted by their v_div_fixup_f32.

    python scripts/isa_phase_mix.py matte5 'pt_persistentILj0ELb1ELb0ELb1ELb0ELb0ELb0ELb0ELb0E' [out.json]
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import build as b  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin/"
# outermost step function -> phase; checked from the outside of the inline chain inwards
PHASE_FUNCS = [("path_begin", "path_begin"), ("path_on_closest", "path_on_closest"), ("path_apply_shadow", "shadow"), ("path_on_shadow", "shadow"),
               ("path_scatter", "path_scatter"), ("path_on_sss", "path_on_sss"), ("sss_travel", "path_on_sss"), ("traverse", "traverse"),
               ("fg_bind_next", "bookkeeping"), ("fg_resolve_slow", "bookkeeping"), ("claim_work_item", "bookkeeping")]
# innermost primitive -> sub-bucket
PRIMS = ["tb_sin", "tb_cos", "tb_pow", "tb_exp2", "tb_log2", "tb_exp", "tb_log", "tb_acos", "tb_asin", "tb_atan2", "tb_atan", "tb_sqrt", "hash13", "rnd", "box_test2", "box_test",
         "tri_test", "ray_prepare", "ray_divide", "ray_assemble", "ray_axes", "load_node", "load_tri", "fetch_surface", "get_material", "one_light_sample", "sample_light",
         "reorient", "ggx", "tb3_normalize", "next_sample", "resolve", "draw", "begin", "walk_addr", "make_refs", "block_region"]


def compile_debug(tu, out_dir):
    src = "kernels/pt_variant_%s.hip" % tu
    obj = os.path.join(out_dir, "k.o")
    cmd = [b.HIPCC] + b.COMMON + b.device_flags(src) + ["--cuda-device-only", "-g", "-c", os.path.join(b.CSRC, src), "-o", obj]
    subprocess.run(cmd, check=True, capture_output=True)
    co = os.path.join(out_dir, "k.co")
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj, "--targets=hipv4-amdgcn-amd-amdhsa--" + b.ARCH.split(":")[0],
                    "--output=" + co], check=True, capture_output=True)
    return co


def disassemble(co, kernel_substr):
    txt = subprocess.run([LLVM + "llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout.splitlines()
    start = [i for i, l in enumerate(txt) if re.match(r"^[0-9a-f]{16} <.*%s.*>:$" % re.escape(kernel_substr), l)]
    if len(start) != 1:
        raise SystemExit("kernel pattern matches %d functions" % len(start))
    ins = []
    for l in txt[start[0] + 1:]:
        if re.match(r"^[0-9a-f]{16} <", l):
            break
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]{12}):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return ins


def symbolize(co, addrs):
    p = subprocess.run([LLVM + "llvm-symbolizer", "--obj=" + co, "--inlines", "--functions=short", "--demangle"], input="\n".join("0x%x" % a for a in addrs) + "\n",
                       capture_output=True, text=True, check=True)
    chains, cur = [], []
    for l in p.stdout.splitlines():
        if not l.strip():
            if cur:
                chains.append(cur)
            cur = []
            continue
        cur.append(l.strip())
    if cur:
        chains.append(cur)
    out = []
    for c in chains:      # pairs of lines: function, file:line:col -- innermost first
        frames = [(c[i], c[i + 1]) for i in range(0, len(c) - 1, 2)]
        out.append(frames)
    return out


def classify(frames):
    names = [re.sub(r"<.*", "", f[0]) for f in frames]           # innermost first
    outer_first = names[::-1]
    phase = None
    for n in outer_first:
        for key, ph in PHASE_FUNCS:
            if n == key or n.startswith(key):
                phase = ph
                break
        if phase:
            break
    if phase is None:
        # the kernel's own body (or a lambda of it): split by the source line of the outermost frame that lies in pt_persistent.inc
        line = None
        for f in frames[::-1]:
            m = re.search(r"pt_persistent\.inc:(\d+)", f[1])
            if m:
                line = int(m.group(1))
                break
        lam = [n for n in names if n.startswith("operator()")]
        phase = ("kernel_body", line)
    prim = None
    for n in names:
        for pnm in PRIMS:
            if n == pnm or n.startswith(pnm):
                prim = pnm
                break
        if prim:
            break
    return phase, prim, names


def main():
    tu, kern = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as d:
        co = compile_debug(tu, d)
        ins = disassemble(co, kern)
        chains = symbolize(co, [a for a, _, _ in ins])
    assert len(chains) == len(ins), (len(chains), len(ins))
    src = open(os.path.join(b.CSRC, "kernels", "pt_persistent.inc")).read().splitlines()

    def body_bucket(line):
        if line is None:
            return "kernel_body:?"
        text = src[line - 1] if 0 < line <= len(src) else ""
        # ranges by markers in the source, so that the buckets follow edits
        def find(marker):
            for i, l in enumerate(src):
                if marker in l:
                    return i + 1
            return 10 ** 9
        done0, done1 = find("if (p.state == ST_DONE) {"), find("/* With the pre-pass the loop is turned by one slot")
        loop0 = find("while (alive) {")
        if done0 <= line < done1:
            return "sample_end+next_sample"
        if line < loop0:
            return "prologue"
        return "loop_glue"

    buckets = collections.OrderedDict()
    rows = []
    for (addr, mn, ops), fr in zip(ins, chains):
        phase, prim, names = classify(fr)
        if isinstance(phase, tuple):
            phase = body_bucket(phase[1])
        # inside traverse: inner step / leaf step / per-ray set-up by primitive
        if phase == "traverse":
            if prim in ("box_test2", "load_node"): phase = "traverse:inner_step"
            elif prim in ("tri_test", "load_tri"): phase = "traverse:leaf_step"
            elif prim in ("ray_prepare", "ray_divide", "ray_assemble", "ray_axes", "box_test"): phase = "traverse:ray_setup"
            else: phase = "traverse:walk_glue"
        kind = "valu" if mn.startswith("v_") else ("salu" if mn.startswith("s_") else ("lds" if mn.startswith("ds_") else "vmem" if re.match(r"(global|buffer|scratch|flat)_", mn) else "other"))
        rows.append((addr, mn, phase, prim, kind))
        bk = buckets.setdefault(phase, collections.Counter())
        bk[kind] += 1
        if mn.startswith("v_div_fixup"): bk["divisions"] += 1
        if mn.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")): bk["transcendental"] += 1
        if kind == "valu" and prim: bk["valu:" + prim] += 1
    doc = {"tu": tu, "kernel": kern, "instructions": len(ins), "buckets": {k: dict(v) for k, v in buckets.items()},
           "totals": dict(sum((collections.Counter({k2: v2 for k2, v2 in v.items() if ":" not in k2}) for v in buckets.values()), collections.Counter()))}
    print(json.dumps(doc, indent=1))
    if len(sys.argv) > 3:
        json.dump(doc, open(sys.argv[3], "w"), indent=1)
        with open(sys.argv[3].replace(".json", ".rows.txt"), "w") as fh:
            for r in rows:
                fh.write("%x %s %s %s %s\n" % r)


if __name__ == "__main__":
    main()
