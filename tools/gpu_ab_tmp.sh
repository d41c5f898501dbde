cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_detect.py tests/test_gpu_shapes.py tests/test_gpu_custom_cascade.py -m gpu -q -x --no-header -p no:cacheprovider -k "not c4_bench_frames or c4_bench_frames_all_128_distinct_vs_oracle[0]" > gpurun_out/pytest3.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/pytest3.log
for rep in 1 2; do
for v in product fpp0; do
  L=""; [ $v != product ] && L=$GRAFT_REPO_ROOT/alt/$v.so
  echo "== $v"
  HEADTRACKR_HIP_LIB=$L timeout 200 python tools/gpu_kernel_times.py c2 "" 3 2>/dev/null | tail -1 | cut -c1-250
  HEADTRACKR_HIP_LIB=$L timeout 200 python tools/gpu_kernel_times.py c4 "" 2 2>/dev/null | tail -1 | cut -c1-250
done; done
HEADTRACKR_HIP_LIB=$GRAFT_REPO_ROOT/alt/tltl.so timeout 300 python tools/gpu_tile_timeline.py c2 2>/dev/null | head -12
