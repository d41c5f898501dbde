#!/bin/bash
# one gpurun call: GPU tests on the default build, then bench A/B (default build and every alt/*.so) for c2 and c4
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
LIB=headtrackr_amd/libheadtrackr_hip.so
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? ($(( $(date +%s) - t0 )) s)"; tail -5 $OUT/pytest_gpu.log
summ() { python -c "
import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2], 'fps',d['value'],'ms/step',d['ms_per_step'],'kernels',d.get('kernel_ms_per_step') or d.get('kernel_ms_per_track_call'))
" $1 "$2"; }
run() {
  for wl in ${AB_WL:-c2 c4}; do
    ST=300; [ $wl = c4 ] && ST=80; [ $wl = c3 ] && ST=8
    timeout 300 python bench.py --workload $wl --steps $ST --cpu-seconds 0 --no-sub > $OUT/ab_$1_$wl.json 2>$OUT/ab_$1_$wl.err; summ $OUT/ab_$1_$wl.json "$1 $wl"
  done
}
run base
cp $LIB /tmp/base.so
for so in alt/*.so; do
  [ -f "$so" ] || continue
  n=$(basename $so .so)
  cp $so $LIB
  if [ "${AB_ALT_TESTS:-0}" = 1 ]; then timeout 600 python -m pytest tests/test_gpu_detect.py tests/test_gpu_sizes.py -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_$n.log 2>&1; echo "$n pytest exit $?"; fi
  run $n
done
cp /tmp/base.so $LIB
