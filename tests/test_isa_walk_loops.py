"""The kernels held to an occupancy (sss at 6, vol at 4 waves per SIMD) keep part of their state in scratch; what must not happen again is a scratch
access INSIDE a walk loop -- round 4 found the ray origin and the stack addresses reloaded there at every step (whole-kernel values the allocator
spilled whole), 7-15 % of the 4K glass scenes' time (walk_owns / walk_lane).  Compile-only: the listing of the two translation
units, mapped by scripts/isa_spill_map.py."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "scripts"))


def _listing(tmp_path, unit):
    from tracerboy_amd import build as b
    out = str(tmp_path / (unit + ".s"))
    src = "kernels/pt_variant_%s.hip" % unit
    cmd = [b.HIPCC] + b.COMMON + list(b.device_flags(src)) + ["--cuda-device-only", "-S", "-o", out, os.path.join(b.CSRC, src)]   # the unit's own scheduler (build.py TU_SCHEDULER)
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


# <F, SCENE_LDS, COUNT, GROUPS, HYBRID, NODEC, TWOLEVEL, PRIMARY, FIRST>.  Every one-level kernel is held to the rule; the two-level walks keep the world ray's slab
# constants for the way back out of an instance and may reload them there (once per instance left, not per step)
@pytest.mark.parametrize("unit", ["sss4", "vol4"])
def test_walk_loops_of_the_occupancy_copies_touch_no_scratch(tmp_path, unit):
    from isa_spill_map import spill_map
    kernels = [k for k in spill_map(_listing(tmp_path, unit)) if "pt_persistent" in k["name"] and k["walk_loops"]]
    one_level = [k for k in kernels if k["name"].rstrip(">").split(", ")[6] == "false"]
    assert len(one_level) >= 8 and len(kernels) > len(one_level)
    # "inside a walk" = at loop depth >= 2 by LLVM's own annotation of the listing (depth 1 is the path loop): independent of which backward
    # branches the mapper takes for loops (a structurised `if (feeler)` block jumps backwards too)
    for k in kernels:
        assert k["walks"] >= 1, k["name"]                    # the mapper found the walks it is asked about
    for k in one_level:
        assert k["deep_ld"] == 0 and k["deep_st"] == 0, (k["name"], k["deep_ld"], k["deep_st"], k["walk_loops"])
    for k in kernels:   # the two-level walks: nowhere a store, at most a handful of reloads (the world ray on the way out of an instance)
        assert k["deep_st"] == 0 and k["deep_ld"] <= 8, (k["name"], k["deep_ld"], k["deep_st"])


# Static spill budget of the kernels the bench workloads run (scratch loads / stores in the listing, all of them outside the walk loops by
# the test above).  The numbers are the round-5 build's plus ~10 %: a change that makes the allocator spill visibly more fails here, on
# the CPU, before anybody times it.  (What the spills cost is measured, not counted: SQ_INSTS_VMEM_WR per launch, profiles/r5.)
BUDGET = {   # unit -> {template arguments after the feature mask: (loads, stores)}; round-5 final build: 362 / 229, 159 / 78, 137 / 53, 93 / 46
    "sss4": {"false, false, true, true, false, false, true, false": (400, 255)},     # van- / bistro-class 4K: groups, split stack, pre-pass
    "vol4": {"false, false, true, true, false, false, true, false": (175, 86),       # vw-van flattened
             "false, false, true, true, false, true, false, false": (150, 60)},      # vw-van two-level
    "surf": {"false, false, true, false, false, false, true, false": (103, 51)},     # Teapot: groups, pre-pass
}


@pytest.mark.parametrize("unit", sorted(BUDGET))
def test_static_spill_budget_of_the_bench_kernels(tmp_path, unit):
    from isa_spill_map import spill_map
    # template arguments after the feature mask, the trailing ones that are off (FIRST, GUIDED: `false`) cut away so that a new defaulted parameter renames nothing here
    def args_of(name):
        a = name.split(", ", 1)[1].rstrip(">").split(", ")
        return ", ".join(a[:8]) if all(x == "false" for x in a[8:]) else ", ".join(a)
    found = {args_of(k["name"]): k for k in spill_map(_listing(tmp_path, unit)) if "pt_persistent" in k["name"]}
    for args, (ld, st) in BUDGET[unit].items():
        assert args in found, (args, sorted(found))
        k = found[args]
        assert k["scratch_ld"] <= ld and k["scratch_st"] <= st, (unit, args, k["scratch_ld"], k["scratch_st"])
