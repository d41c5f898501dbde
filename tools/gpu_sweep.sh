#!/bin/bash
# tools/gpu_sweep.sh — measurement sweeps (tile-kernel stage breakdown, hand-off bias); prints compact kernel timings
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu.log
summ() { python -c "
import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2], 'fps',d['value'],'ms/step',d['ms_per_step'],'kernels',d.get('kernel_ms_per_step'))
" $1 "$2"; }
for wl in ${SWEEP_WL:-c2 c4}; do
  ST=10; [ $wl = c4 ] && ST=4
  timeout 300 python bench.py --workload $wl --steps $ST --warmup 2 --cpu-seconds 0 > $OUT/sw_$wl.json 2>$OUT/sw_$wl.err; summ $OUT/sw_$wl.json "$wl base"
  for s in ${SWEEP_STOP:-0 1 2 4}; do
    HT_DEBUG_STOP_STAGE=$s timeout 300 python bench.py --workload $wl --steps $ST --warmup 2 --cpu-seconds 0 > $OUT/sw_${wl}_stop$s.json 2>$OUT/sw_${wl}_stop$s.err; summ $OUT/sw_${wl}_stop$s.json "$wl stop=$s"
  done
  for b in ${SWEEP_BIAS:-1 2 6 12}; do
    HT_DEBUG_DEEP_BIAS=$b timeout 300 python bench.py --workload $wl --steps $ST --warmup 2 --cpu-seconds 0 > $OUT/sw_${wl}_bias$b.json 2>$OUT/sw_${wl}_bias$b.err; summ $OUT/sw_${wl}_bias$b.json "$wl bias=$b"
  done
done
