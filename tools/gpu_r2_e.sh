#!/bin/bash
# one gpurun call: for the default build and every alt/*.so (except the timeline build): full GPU test suite + bench c2 / c4; then the
# tile-kernel phase timeline from alt/tltl.so
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
run() {
  t0=$(date +%s)
  timeout 600 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_$1.log 2>&1; echo "$1 pytest exit $? ($(( $(date +%s) - t0 )) s): $(tail -1 $OUT/pytest_$1.log)"
  for wl in c2 c4; do
    ST=300; [ $wl = c4 ] && ST=80
    timeout 300 python bench.py --workload $wl --steps $ST --cpu-seconds 0 --no-sub > $OUT/ab_$1_$wl.json 2>$OUT/ab_$1_$wl.err; summ $OUT/ab_$1_$wl.json "$1 $wl"
  done
}
cp $LIB /tmp/base.so
run new
for n in ${AB_ALTS:-old mf1 mf3}; do
  [ -f alt/$n.so ] || continue
  cp alt/$n.so $LIB
  run $n
done
if [ -f alt/tltl.so ]; then
  cp alt/tltl.so $LIB
  timeout 300 python tools/gpu_tile_timeline.py c2 2>&1 | tee $OUT/tile_timeline_c2.txt
  timeout 300 python tools/gpu_tile_timeline.py c4 2>&1 | tee $OUT/tile_timeline_c4.txt
fi
cp /tmp/base.so $LIB
