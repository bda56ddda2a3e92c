#!/usr/bin/env python3
"""Where a kernel's register spills sit: static scratch loads / stores of every pt_persistent / pt_primary kernel of an assembly listing, and how
many of them are inside the walk loops (loops that fetch nodes with four global_load_dwordx4) -- those run once per step, the rest once per bounce.
   hipcc ... -S -o k.s pt_variant_sss4.hip;  python scripts/isa_spill_map.py k.s [substring of the demangled kernel name]
tests/test_isa_walk_loops.py uses spill_map() to keep the walk loops of the kernels held to an occupancy free of scratch."""
import re, subprocess, sys


def spill_map(text, want=""):
    """[{name, instr, scratch_ld, scratch_st, walk_loops: [(label, instr, ld, st), ...]}] for the kernels whose demangled name contains `want`"""
    out = []
    for m in re.finditer(r"^(_ZN\S*pt_(?:persistent|primary)\S*):.*$", text, re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"\(.*", "", name).replace("void ", "")
        if want not in name: continue
        body = text[m.start():text.index(".end_amdhsa_kernel", m.start())]
        blocks, cur = [], {"label": "entry", "ins": [], "depth": 0, "header": None}
        for line in body.split("\n"):
            lab = re.match(r"^(\.LBB\d+_\d+):", line)
            if lab:
                # LLVM's own loop annotation of the block ("in Loop: Header=BBn_m Depth=d"; inner headers carry "Parent Loop ... Depth=d" lines
                # below the label, the deepest of which is the block's depth + 1 -- read from the label line and the comment lines that follow)
                hdr = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", line)
                blocks.append(cur); cur = {"label": lab.group(1), "ins": [], "depth": max([int(d) for d in re.findall(r"Depth=(\d+)", line)] or [0]),
                                           "header": (hdr.group(1), int(hdr.group(2))) if hdr else None}
            elif line.lstrip().startswith(";") and "Depth=" in line and not cur["ins"]:
                cur["depth"] = max(cur["depth"], max(int(d) for d in re.findall(r"Depth=(\d+)", line)))
            else:
                s = line.strip()
                if s and re.match(r"^[a-z]", s): cur["ins"].append(s)
        blocks.append(cur)
        index = {b["label"]: k for k, b in enumerate(blocks)}
        loops = set()
        for k, b in enumerate(blocks):
            for ins in b["ins"]:
                br = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", ins)
                if br and br.group(1) in index and index[br.group(1)] <= k: loops.add((index[br.group(1)], k))
        count = lambda prefix, a=0, b=None: sum(sum(1 for i in blocks[k]["ins"] if i.startswith(prefix)) for k in range(a, len(blocks) if b is None else b + 1))
        # node and triangle fetches: 16-B loads, or 12-B ones where a kernel reads nothing of a triangle record's fourth words (it shades from the
        # triangles' shading records and never sees the indices kept there)
        wide = lambda a, b: count("global_load_dwordx4", a, b) + count("global_load_dwordx3", a, b)
        size = lambda a, b: sum(len(blocks[k]["ins"]) for k in range(a, b + 1))
        walk = [(a, b) for a, b in loops if wide(a, b) >= 4 and size(a, b) < 600]
        outer = [l for l in walk if not any(o != l and o[0] <= l[0] and l[1] <= o[1] for o in walk)]   # a walk = the inner-node loop nested in the while-while loop
        # Scratch accesses INSIDE A WALK, by LLVM's own loop annotation of the listing: the kernel's outermost loop is the path loop (depth 1); a
        # depth-2 loop whose extent (its own blocks and everything nested between them) fetches nodes -- four or more global_load_dwordx4 -- is a
        # walk.  Independent of which backward branches the heuristic above takes for loops (a structurised `if (feeler)` block jumps backwards
        # too), and blind to the other depth-2 loops (the sample-number loop of frame-group mode stores the lane's frame there, once per sample).
        extents = {}
        for k, b in enumerate(blocks):
            if b["header"] and b["header"][1] == 2: lo, hi = extents.get(b["header"][0], (k, k)); extents[b["header"][0]] = (min(lo, k), max(hi, k))
        walks = [(lo, hi) for lo, hi in extents.values() if wide(lo, hi) >= 4]
        deep = lambda prefix: sum(count(prefix, lo, hi) for lo, hi in walks)
        out.append({"name": name, "instr": size(0, len(blocks) - 1), "scratch_ld": count("scratch_load"), "scratch_st": count("scratch_store"),
                    "deep_ld": deep("scratch_load"), "deep_st": deep("scratch_store"), "walks": len(walks),
                    "walk_loops": [(blocks[a]["label"], size(a, b), count("scratch_load", a, b), count("scratch_store", a, b)) for a, b in sorted(outer)]})
    return out


if __name__ == "__main__":
    for k in spill_map(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else ""):
        print("%-70s instr %5d  scratch ld %3d st %3d  inside its %d walks: ld %d st %d  walk loops (label, instr, ld, st): %s" % (
            k["name"], k["instr"], k["scratch_ld"], k["scratch_st"], k["walks"], k["deep_ld"], k["deep_st"], k["walk_loops"]))
