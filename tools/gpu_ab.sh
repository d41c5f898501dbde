#!/bin/bash
# tools/gpu_ab.sh — A/B of build-time variants: the default library first (with the GPU test suite), then every
# alt/*.so (built here with different -D knobs) copied over it.  Prints one compact line per bench run.
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
LIB=headtrackr_amd/libheadtrackr_hip.so
summ() { python -c "
import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2], 'fps',d['value'],'ms/step',d['ms_per_step'],'kernels',d.get('kernel_ms_per_step'))
" $1 "$2"; }
run() { # name
  for wl in ${AB_WL:-c2 c4}; do
    ST=10; [ $wl = c4 ] && ST=4
    timeout 300 python bench.py --workload $wl --steps $ST --warmup 2 --cpu-seconds 0 > $OUT/ab_$1_$wl.json 2>$OUT/ab_$1_$wl.err; summ $OUT/ab_$1_$wl.json "$1 $wl"
  done
}
if [ "${AB_TESTS:-1}" = 1 ]; then
  echo "== pytest gpu (default build)"; timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu.log
fi
run base
cp $LIB /tmp/base.so
for so in alt/*.so; do
  [ -f "$so" ] || continue
  n=$(basename $so .so)
  cp $so $LIB
  if [ "${AB_ALT_TESTS:-1}" = 1 ]; then timeout 600 python -m pytest tests/test_gpu_detect.py tests/test_gpu_sizes.py -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_$n.log 2>&1; echo "$n pytest exit $?"; fi
  run $n
done
cp /tmp/base.so $LIB
