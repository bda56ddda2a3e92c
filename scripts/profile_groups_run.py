import sys, os, json
sys.path.insert(0, os.getcwd())
import bench
from tracerboy_amd import api
leg = sys.argv[1] if len(sys.argv) > 1 else "c2"
w = bench.WORKLOADS[leg]; W, H, SPP = w["W"], w["H"], w["spp"]
b = bench.Bench(api, 0); tb = b.tb; s = b.settings(w["depth"]); b.load_workload(leg)
tb.SetOption("overlap_launches", 0); tb.SetOption("debug_profile_groups", 1)
for _ in range(2):
    tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0)
tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0)
prof = tb.WaveProfile(); n = W * H * SPP
st = tb.ReadbackStats().rays
print(json.dumps({"leg": leg, "kernel_ms": tb.GetOption("last_kernel_us") / 1e3, "samples": n, "counted_samples": int(st.samples), "rays_per_sample": st.rays / max(st.samples, 1), "boxes_per_sample": st.boxesTested / max(st.samples, 1), "tris_per_sample": st.trianglesTested / max(st.samples, 1),
   "phases": {k: {"lane_execs_per_sample": round(v[0] / n, 4), "wave_trips_per_sample": round(v[1] / n, 5), "occupancy": round(v[2], 4)} for k, v in prof.items()}}, indent=1))
