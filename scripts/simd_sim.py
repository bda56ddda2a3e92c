#!/usr/bin/env python3
"""SIMD-efficiency simulator for BVH-walk scheduling policies (analysis tool, no GPU needed).

Input: the oracle's ray-step log (TB_ORACLE_RAY_LOG=<file>, see oracle/tb_oracle.cpp): one line per traversal,
"x y frame kind steps", steps = sequence of I (inner node visit: both children tested) and L (one triangle tested).
The simulator replays those sequences on 64-lane waves under different schedulers and counts wave trips through the
inner-node body and the leaf body (the two hot loops of traverse(), pt_device.hpp) and the lanes active in them.

  python scripts/simd_sim.py gpurun_out/rays_cornell.log
"""
import collections
import sys

CI, CL = 85.0, 160.0   # VALU cycles per trip through the inner-node body / the leaf body (gfx950 ISA x scripts/microbench/valu_issue.hip costs)


def load(path):
    samples = collections.defaultdict(list)   # (x, y, frame) -> [(kind, steps)]
    for line in open(path):
        x, y, f, kind, steps = line.split()
        samples[(int(x), int(y), int(f))].append((kind, "" if steps == "-" else steps))
    return samples


class Wave:
    """64 lanes walking step strings under the while-while schedule with parking."""

    def __init__(self, park_min=8):
        self.s = [""] * 64; self.p = [0] * 64
        self.park_min = park_min
        self.inner_trips = self.inner_active = self.leaf_trips = self.leaf_active = 0

    def busy(self, l): return self.p[l] < len(self.s[l])
    def nbusy(self): return sum(1 for l in range(64) if self.busy(l))
    def give(self, l, steps): self.s[l] = steps; self.p[l] = 0

    def round(self):
        """one outer iteration of traverse(): inner loop until < park_min lanes still descend, then one leaf step"""
        s, p = self.s, self.p
        inloop = [l for l in range(64) if p[l] < len(s[l]) and s[l][p[l]] == "I"]
        while inloop:
            self.inner_trips += 1; self.inner_active += len(inloop)
            nxt = []
            for l in inloop:
                p[l] += 1
                if p[l] < len(s[l]) and s[l][p[l]] == "I": nxt.append(l)
            inloop = nxt
            if len(inloop) < self.park_min: break
        leaf = [l for l in range(64) if p[l] < len(s[l]) and s[l][p[l]] == "L"]
        if leaf:
            self.leaf_trips += 1; self.leaf_active += len(leaf)
            for l in leaf: p[l] += 1

    def cost(self): return self.inner_trips * CI + self.leaf_trips * CL

    def merge(self, o):
        self.inner_trips += o.inner_trips; self.inner_active += o.inner_active; self.leaf_trips += o.leaf_trips; self.leaf_active += o.leaf_active


def report(name, w, nsamples):
    it, ia, lt, la = w.inner_trips, w.inner_active, w.leaf_trips, w.leaf_active
    print("%-44s inner occ %.3f  leaf occ %.3f  cost/sample %7.1f  (ideal %.1f)" % (
        name, ia / max(it, 1) / 64, la / max(lt, 1) / 64, (it * CI + lt * CL) / nsamples, (ia * CI + la * CL) / 64 / nsamples))


def tiles(samples):
    """8x8 pixel tiles -> lane -> frames -> ray list"""
    t = collections.defaultdict(lambda: collections.defaultdict(dict))
    for (x, y, f), rays in samples.items():
        t[(x // 8, y // 8)][(y % 8) * 8 + (x % 8)][f] = rays
    return t


def lockstep(samples, park_min=8):
    """pt_persistent: a lane owns a pixel, frames in order; per iteration slot 1 = bounce ray, slot 2 = shadow feeler"""
    total = Wave(park_min)
    for tile, lanes in tiles(samples).items():
        seq = {l: [r for f in sorted(fr) for r in fr[f]] for l, fr in lanes.items()}
        pos = {l: 0 for l in seq}
        while any(pos[l] < len(seq[l]) for l in seq):
            for kind in ("E", "S"):
                w = Wave(park_min)
                for l in seq:
                    if pos[l] < len(seq[l]) and (seq[l][pos[l]][0] == kind or (kind == "E" and seq[l][pos[l]][0] == "W")):
                        w.give(l, seq[l][pos[l]][1]); pos[l] += 1
                while w.nbusy(): w.round()
                total.merge(w)
    return total


def lockstep_regen(samples, regen_thresh=1, park_min=8, costs=(1100.0, 750.0, 400.0, 250.0, 300.0)):
    """pt_persistent with its non-traversal phases priced too (wave-instructions per trip: regenerate, closest-hit
    shading, scatter, shadow slot, ray setup) and a regeneration threshold: finished lanes start their next sample only
    when at least regen_thresh lanes of the wave are waiting (or nobody is alive)."""
    c_regen, c_shade, c_scatter, c_shadow, c_setup = costs
    total = Wave(park_min); other = 0.0; regen_trips = iters = 0
    for tile, lanes in tiles(samples).items():
        todo = {l: [fr[f] for f in sorted(fr)] for l, fr in lanes.items()}   # lane -> list of samples (ray lists)
        cur = {l: None for l in todo}; pos = {l: 0 for l in todo}
        waiting = set(todo)                                                    # lanes that need a new sample
        while True:
            alive = [l for l in cur if cur[l] is not None]
            can = [l for l in waiting if todo[l]]
            if can and (len(can) >= regen_thresh or not alive):
                for l in can: cur[l] = todo[l].pop(0); pos[l] = 0; waiting.discard(l)
                regen_trips += 1; other += c_regen
                alive = [l for l in cur if cur[l] is not None]
            if not alive:
                if not any(todo[l] for l in waiting): break
                continue
            iters += 1
            for kind in ("E", "S"):
                w = Wave(park_min); n = 0
                for l in alive:
                    q = cur[l]
                    if pos[l] < len(q) and (q[pos[l]][0] == kind or (kind == "E" and q[pos[l]][0] == "W")):
                        w.give(l, q[pos[l]][1]); pos[l] += 1; n += 1
                if n:
                    other += c_setup + (c_shade if kind == "E" else c_shadow)
                    while w.nbusy(): w.round()
                    total.merge(w)
            other += c_scatter
            for l in alive:
                if pos[l] >= len(cur[l]): cur[l] = None; waiting.add(l)
    n = len(samples)
    print("%-44s traversal %7.1f  other %7.1f  total %7.1f per sample   (regen trips/64 samples %.2f, iterations %.2f)" % (
        "lock-step regen_thresh=%d" % regen_thresh, total.cost() / n, other / n, (total.cost() + other) / n, regen_trips * 64.0 / n, iters * 64.0 / n))


def lockstep_sliced(samples, cap=8, park_min=2, costs=(1100.0, 750.0, 400.0, 250.0, 300.0)):
    """pt_persistent with TIME-SLICED traversal: a traversal slot runs at most `cap` rounds (round = inner loop + one leaf step);
    lanes whose ray is not finished keep their walk state (ref, stack height, best hit; the ray constants are recomputed) and
    resume in the same slot of the next iteration, next to the new rays of the lanes that went on -- the slot is no longer as long
    as its slowest ray.  Non-traversal phases are priced as in lockstep_regen (regenerate, closest-hit shading, scatter, shadow
    slot, ray setup); a phase costs its price whenever at least one lane runs it."""
    c_regen, c_shade, c_scatter, c_shadow, c_setup = costs
    total = Wave(park_min); other = 0.0; iters = 0
    for tile, lanes in tiles(samples).items():
        todo = {l: [fr[f] for f in sorted(fr)] for l, fr in lanes.items()}
        cur = {l: None for l in todo}; pos = {l: 0 for l in todo}
        walk = {l: None for l in todo}      # lane -> [kind, steps, position] of a walk in flight
        while True:
            can = [l for l in cur if cur[l] is None and todo[l]]
            if can:
                for l in can: cur[l] = todo[l].pop(0); pos[l] = 0
                other += c_regen
            alive = [l for l in cur if cur[l] is not None]
            if not alive: break
            iters += 1
            scatter = False
            for kind in ("E", "S"):
                w = Wave(park_min); members = []
                for l in alive:
                    if walk[l] is not None:
                        if walk[l][0] == kind: w.s[l] = walk[l][1]; w.p[l] = walk[l][2]; members.append(l)
                        continue
                    q = cur[l]
                    if pos[l] < len(q) and (q[pos[l]][0] == kind or (kind == "E" and q[pos[l]][0] == "W")):
                        w.give(l, q[pos[l]][1]); walk[l] = [kind, q[pos[l]][1], 0]; pos[l] += 1; members.append(l)
                if not members: continue
                other += c_setup
                rounds = 0
                while w.nbusy() and rounds < cap: w.round(); rounds += 1
                total.merge(w)
                fin = [l for l in members if not w.busy(l)]
                for l in members:
                    if w.busy(l): walk[l][2] = w.p[l]
                    else: walk[l] = None
                if fin:
                    other += c_shade if kind == "E" else c_shadow
                    if kind == "S": scatter = True
                    else:
                        # a lane whose bounce ray is done and has no shadow feeler next scatters (or ends) in this iteration
                        for l in fin:
                            q = cur[l]
                            if not (pos[l] < len(q) and q[pos[l]][0] == "S"): scatter = True
            if scatter: other += c_scatter
            for l in alive:
                if walk[l] is None and pos[l] >= len(cur[l]): cur[l] = None
    n = len(samples)
    print("%-44s traversal %7.1f  other %7.1f  total %7.1f per sample   (iterations/64 samples %.2f, inner occ %.3f leaf occ %.3f)" % (
        "lock-step sliced cap=%d park_min=%d" % (cap, park_min), total.cost() / n, other / n, (total.cost() + other) / n, iters * 64.0 / n,
        total.inner_active / max(total.inner_trips, 1) / 64, total.leaf_active / max(total.leaf_trips, 1) / 64))


def lockstep_hoisted(samples, park_min=8, share_wave=False, refill_below=48):
    """lock-step per wave, but the bounce ray is issued together with the shadow feeler (pt_pooled's hoisting) and a
    lane walks its two rays back to back inside one traversal loop; share_wave: the wave's rays go through a wave-local
    pool with dynamic fetch instead of staying on their own lane"""
    total = Wave(park_min)
    for tile, lanes in tiles(samples).items():
        seq = {l: [r for f in sorted(fr) for r in fr[f] + [("X", "")]] for l, fr in lanes.items()}   # X = sample boundary
        pos = {l: 0 for l in seq}
        while any(pos[l] < len(seq[l]) for l in seq):
            mine = {}
            for l in seq:
                if pos[l] >= len(seq[l]): continue
                q = seq[l]; i = pos[l]
                if q[i][0] == "X": pos[l] += 1; continue   # regenerate: costs this lane one round (like the kernel's state machine)
                take = [q[i][1]]; i += 1
                if q[i - 1][0] == "S" and q[i][0] != "X": take.append(q[i][1]); i += 1
                mine[l] = take; pos[l] = i
            if not mine: continue
            w = Wave(park_min)
            if share_wave:
                pool = [r for l in mine for r in mine[l]]; qi = 0
                while True:
                    if qi < len(pool) and w.nbusy() < refill_below:
                        for l in range(64):
                            if not w.busy(l) and qi < len(pool): w.give(l, pool[qi]); qi += 1
                    if not w.nbusy():
                        if qi >= len(pool): break
                        continue
                    w.round()
            else:
                left = {l: list(v) for l, v in mine.items()}
                while True:
                    for l in left:
                        if not w.busy(l) and left[l]: w.give(l, left[l].pop(0))
                    if not w.nbusy():
                        if not any(left.values()): break
                        continue
                    w.round()
            total.merge(w)
    return total


def pooled(samples, R=1, waves_per_block=4, refill_below=48, park_min=8, hoist=True, block_tiles=None):
    """pt_pooled: a block of 4 waves (2x2 tiles), each lane R samples in flight; per round every path contributes its
    shadow feeler and (hoisted) the next bounce ray; the waves drain the round's pool with dynamic fetch."""
    total = Wave(park_min)
    T = tiles(samples)
    blocks = collections.defaultdict(list)
    for (tx, ty), lanes in T.items(): blocks[(tx // 2, ty // 2)].append(lanes)
    for b, tl in blocks.items():
        # path slots: (tile index, lane, r) -> queue of frames
        paths = []
        for lanes in tl:
            for l, fr in lanes.items():
                frames = sorted(fr)
                for r in range(R):
                    mine = [fr[f] for f in frames[r::R]]
                    paths.append({"samples": mine, "si": 0, "ri": 0})
        while True:
            pool = []
            for p in paths:
                if p["si"] >= len(p["samples"]): continue
                rays = p["samples"][p["si"]]
                # this round: the next ray, and if it is a shadow feeler also the bounce ray behind it (hoisted)
                take = 1
                if hoist and rays[p["ri"]][0] == "S" and p["ri"] + 1 < len(rays): take = 2
                elif hoist and rays[p["ri"]][0] == "E" and p["ri"] + 1 < len(rays) and rays[p["ri"] + 1][0] == "S":
                    take = 1   # the feeler of this bounce is only known after shading the hit
                for k in range(take): pool.append(rays[p["ri"] + k][1])
                p["ri"] += take
                if p["ri"] >= len(rays): p["si"] += 1; p["ri"] = 0
            if not pool: break
            ws = [Wave(park_min) for _ in range(waves_per_block)]
            clock = [0.0] * waves_per_block
            qi = 0
            while True:
                # the wave that is earliest in time acts next
                order = sorted(range(waves_per_block), key=lambda i: clock[i])
                acted = False
                for i in order:
                    w = ws[i]
                    nb = w.nbusy()
                    if qi < len(pool) and nb < refill_below:
                        for l in range(64):
                            if not w.busy(l) and qi < len(pool): w.give(l, pool[qi]); qi += 1
                        nb = w.nbusy()
                    if nb:
                        before = w.cost(); w.round(); clock[i] += w.cost() - before + 1e-3; acted = True
                        break
                if not acted: break
            for w in ws: total.merge(w)
    return total


def infinite_pool(samples, refill_below=48, park_min=8):
    """upper bound of while-while + dynamic fetch: one wave, all rays of the image in one queue"""
    pool = [r[1] for k in sorted(samples) for r in samples[k]]
    w = Wave(park_min); qi = 0
    while True:
        if qi < len(pool) and w.nbusy() < refill_below:
            for l in range(64):
                if not w.busy(l) and qi < len(pool): w.give(l, pool[qi]); qi += 1
        if not w.nbusy():
            if qi >= len(pool): break
            continue
        w.round()
    return w


def main():
    samples = load(sys.argv[1])
    n = len(samples)
    nr = sum(len(v) for v in samples.values())
    print("%d samples, %.2f rays/sample, %.2f inner + %.2f leaf steps per ray" % (
        n, nr / n, sum(r[1].count("I") for v in samples.values() for r in v) / nr, sum(r[1].count("L") for v in samples.values() for r in v) / nr))
    report("lock-step (pt_persistent) PARK_MIN=2", lockstep(samples, 2), n)
    lockstep_regen(samples, 1, park_min=2)
    for cap in (1000, 16, 8, 6, 4, 3, 2, 1):
        lockstep_sliced(samples, cap, park_min=2)
    # every scheduler DESIGN.md / docs/experiments quote (an early return used to hide the rows below)
    report("lock-step + hoisted bounce ray, own lane", lockstep_hoisted(samples), n)
    report("lock-step + hoisted, wave-local pool", lockstep_hoisted(samples, share_wave=True), n)
    for R in (1, 2, 4):
        report("pooled block R=%d" % R, pooled(samples, R), n)
    for rb in (48,):
        for pm in (8, 16):
            report("infinite pool refill<%d PARK_MIN=%d" % (rb, pm), infinite_pool(samples, rb, pm), n)


if __name__ == "__main__":
    main()
