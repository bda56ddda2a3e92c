cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reciprocal or device_math or cornell or full_size or trace_closest or config_c" 2>&1 | tail -5
for rep in 1 2 3; do
 for tag in base norcp; do
  if [ $tag = base ]; then unset TB_LIB; else export TB_LIB=$PWD/tracerboy_amd/_sweep/libtracerboy_hip_$tag.so; fi
  for leg in c2 c3 c4 vwvan; do timeout 300 python scripts/async_rate.py $leg --reps 3 --steps 12 2>/dev/null | sed "s/^/$tag /" | tee -a gpurun_out/r6/rcp_ab.txt | cut -c1-200; done
 done
done
