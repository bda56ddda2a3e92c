#!/usr/bin/env python3
"""Summary of scripts/c2_instruction_mix.sh: per-phase dynamic VALU wave-instructions per sample of the lock-step kernel.
   python scripts/c2_instruction_mix.py gpurun_out/r6/mix_c2 c2 [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import bench  # noqa: E402
from pmc_aggregate import aggregate  # noqa: E402
from tracerboy_amd import build as tb_build  # noqa: E402
out, leg = sys.argv[1], sys.argv[2]
w = bench.WORKLOADS[leg]; samples = w["W"] * w["H"] * w["spp"]
PHASES = {"d1": "path_begin (camera ray, hash13 seed, 16 rand() of which 2 are read)", "d2": "both walks whole (ray set-up + inner steps + leaf steps + stack)",
          "d3": "inner-node step: the two-box test alone", "d4": "leaf step: the watertight triangle test alone", "d5": "path_on_closest (hit attributes, material, emissive, light sample, feeler set-up)",
          "d6": "path_scatter (cosine sample, reorientation, pdf, throughput, Russian roulette)", "d7": "ray set-up (GetRayData: 5 divisions, axis order; root box test)", "d8": "every rand() (tb_sin with binary64 range reduction)"}
times = {}
for l in open(os.path.join(out, "times.txt")):
    lib, js = l.split(" ", 1); times[lib] = json.loads(js)
rows = {}
for lib in ["base"] + sorted(PHASES):
    agg = aggregate("%s/pmc_%s/**/*counter_collection.csv" % (out, lib), lambda k: "pt_persistent<" in k)
    if not agg: continue
    name, c = max(agg.items(), key=lambda kv: kv[1].get("SQ_INSTS_VALU", 0))
    rows[lib] = {"kernel": name, **{k: v for k, v in c.items() if k.startswith("SQ_")}, "launches": c.get("dispatches"), **times.get(lib, {})}
base = rows["base"]
doc = {"workload": leg, "samples_per_launch": samples, "_kernel_digest": tb_build.kernel_digest(), "kernel": base["kernel"],
       "valu_wave_instructions_per_sample": round(base["SQ_INSTS_VALU"] / samples, 2), "salu_per_sample": round(base.get("SQ_INSTS_SALU", 0) / samples, 2),
       "lds_per_sample": round(base.get("SQ_INSTS_LDS", 0) / samples, 2), "lane_utilisation": round(base["SQ_THREAD_CYCLES_VALU"] / (64.0 * base["SQ_ACTIVE_INST_VALU"]), 3),
       "kernel_ms": base.get("kernel_ms"), "picture_sha1": base.get("picture_sha1"), "phases": {}}
for lib, what in PHASES.items():
    if lib not in rows: continue
    r = rows[lib]
    d = (r["SQ_INSTS_VALU"] - base["SQ_INSTS_VALU"]) / samples
    doc["phases"][lib] = {"what": what, "valu_wave_instructions_per_sample": round(d, 2), "share_of_valu": round(d / (base["SQ_INSTS_VALU"] / samples), 4),
                          "lane_instructions_per_sample": round((r["SQ_THREAD_CYCLES_VALU"] - base["SQ_THREAD_CYCLES_VALU"]) / 4.0 / samples, 1),
                          "kernel_ms_with_phase_doubled": r.get("kernel_ms"), "time_share": round((r.get("kernel_ms", 0) - base.get("kernel_ms", 0)) / base["kernel_ms"], 4) if base.get("kernel_ms") else None,
                          "same_picture": r.get("picture_sha1") == base.get("picture_sha1")}
ph = doc["phases"]
if all(k in ph for k in ("d1", "d2", "d5", "d6")):
    known = sum(ph[k]["valu_wave_instructions_per_sample"] for k in ("d1", "d2", "d5", "d6"))
    doc["rest_per_sample"] = {"valu_wave_instructions_per_sample": round(doc["valu_wave_instructions_per_sample"] - known, 2),
                              "what": "loop glue, state dispatch, sample end (1 rand + the 16-B store), drawing the next sample, judging the feeler, workgroup prologue"}
    if all(k in ph for k in ("d3", "d4", "d7")):
        doc["walk_rest_per_sample"] = {"valu_wave_instructions_per_sample": round(ph["d2"]["valu_wave_instructions_per_sample"] - sum(ph[k]["valu_wave_instructions_per_sample"] for k in ("d3", "d4", "d7")), 2),
                                       "what": "of the walks: node / triangle address arithmetic, child ordering, push / pop, parking ballots, loop control"}
print(json.dumps(doc, indent=1))
json.dump(doc, open(sys.argv[3] if len(sys.argv) > 3 else os.path.join(out, "instruction_mix.json"), "w"), indent=1)
