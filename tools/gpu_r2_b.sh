#!/bin/bash
# one gpurun call: LDS micro benchmark, GPU tests, bench with / without the resample fast paths
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
./tools/micro/lds_unaligned_bench > $OUT/lds_unaligned.txt 2>&1; cat $OUT/lds_unaligned.txt
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? ($(( $(date +%s) - t0 )) s)"; tail -5 $OUT/pytest_gpu.log
summ() { python -c "
import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2], 'fps',d['value'],'ms/step',d['ms_per_step'],'kernels',d.get('kernel_ms_per_step'))
" $1 "$2"; }
for wl in c2 c4; do
  ST=200; [ $wl = c4 ] && ST=60
  timeout 300 python bench.py --workload $wl --steps $ST --cpu-seconds 0 --no-sub > $OUT/rs_fast_$wl.json 2>$OUT/rs_fast_$wl.err; summ $OUT/rs_fast_$wl.json "fast $wl"
  HT_DEBUG_RS_NOFAST=1 timeout 300 python bench.py --workload $wl --steps $ST --cpu-seconds 0 --no-sub > $OUT/rs_slow_$wl.json 2>$OUT/rs_slow_$wl.err; summ $OUT/rs_slow_$wl.json "f64  $wl"
done
