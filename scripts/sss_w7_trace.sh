cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cp tracerboy_amd/libtracerboy_hip.so /tmp/lib_default.so
trap 'cp /tmp/lib_default.so tracerboy_amd/libtracerboy_hip.so' EXIT   # an interrupted run must not leave a sweep build in the tree
for W in 6 7; do
  cp tracerboy_amd/_sweep/libtracerboy_hip_sss$W.so tracerboy_amd/libtracerboy_hip.so
  rm -rf gpurun_out/w$W
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/w$W -o t -- python3 bench.py --scene proc1:700000 --builder 4 --width 3840 --height 2160 --spp 8 --depth 6 --legs none --no-cpu-baseline --steps 3 --sync-steps > gpurun_out/w$W.json 2> gpurun_out/w$W.err
  python3 - $W <<'PY'
import csv,glob,sys,collections
f=glob.glob("gpurun_out/w%s/**/*kernel_trace.csv"%sys.argv[1],recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows:
    if "pt_" in r["Kernel_Name"]: agg[(r["Kernel_Name"][:60], r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for k,v in agg.items(): print(sys.argv[1], k, "n",len(v),"ms max %.2f"%max(v))
PY
  tail -c 300 gpurun_out/w$W.json | head -c 10 > /dev/null
  python3 -c "
import json; d=json.loads(open('gpurun_out/w$W.json').read().strip().splitlines()[-1]); print('waves', $W, 'value', d['value'])"
done
cp /tmp/lib_default.so tracerboy_amd/libtracerboy_hip.so
