cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in product fp0 fpc256 fpc512; do
  L=""; [ $v != product ] && L=$GRAFT_REPO_ROOT/alt/$v.so
  echo "== $v"
  HEADTRACKR_HIP_LIB=$L timeout 200 python tools/gpu_kernel_times.py c2 "" 3 2>/dev/null | tail -1 | cut -c1-250
  HEADTRACKR_HIP_LIB=$L timeout 200 python tools/gpu_kernel_times.py c4 "" 2 2>/dev/null | tail -1 | cut -c1-250
done; done
