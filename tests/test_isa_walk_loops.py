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
    cmd = [b.HIPCC] + b.COMMON + b.DEVICE + ["--cuda-device-only", "-S", "-o", out, os.path.join(b.CSRC, "kernels", "pt_variant_%s.hip" % unit)]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


# <F, SCENE_LDS, COUNT, GROUPS, HYBRID, NODEC, TWOLEVEL, PRIMARY>.  Every one-level kernel is held to the rule; the two-level walks keep the world ray's slab
# constants for the way back out of an instance and may reload them there (once per instance left, not per step)
@pytest.mark.parametrize("unit", ["sss4", "vol4"])
def test_walk_loops_of_the_occupancy_copies_touch_no_scratch(tmp_path, unit):
    from isa_spill_map import spill_map
    kernels = [k for k in spill_map(_listing(tmp_path, unit)) if "pt_persistent" in k["name"] and k["walk_loops"]]
    one_level = [k for k in kernels if k["name"].rstrip(">").split(", ")[6] == "false"]
    assert len(one_level) >= 8 and len(kernels) > len(one_level)
    for k in one_level:
        assert all(ld == 0 and st == 0 for _, _, ld, st in k["walk_loops"]), (k["name"], k["walk_loops"])
    for k in kernels:   # and nowhere a store, or more than a handful of reloads
        assert all(st == 0 and ld <= 8 for _, _, ld, st in k["walk_loops"]), (k["name"], k["walk_loops"])
