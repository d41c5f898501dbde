#!/bin/bash
# split / bias / deep-kernel-version sweep
set -u
OUT=gpurun_out; mkdir -p $OUT
summ() { python -c "
import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); k=d.get('kernel_ms_per_step'); print(sys.argv[2], 'fps',d['value'],'ms',d['ms_per_step'],'tiles',k.get('scan_tiles'),'deep',k.get('scan_deep'),'sum',round(k.get('scan_tiles',0)+k.get('scan_deep',0),4))
" $1 "$2"; }
for wl in c2 c4; do
  ST=10; [ $wl = c4 ] && ST=4
  for dv in 2; do for sp in ${SW_SPLIT:-6 7 8}; do for b in ${SW_BIAS:-0 1 2 3}; do
    HT_DEBUG_DEEP_V=$dv HT_DEBUG_SPLIT=$sp HT_DEBUG_DEEP_BIAS=$b timeout 300 python bench.py --workload $wl --steps $ST --warmup 2 --cpu-seconds 0 > $OUT/s2.json 2>$OUT/s2.err; summ $OUT/s2.json "$wl deepv=$dv split=$sp bias=$b"
  done; done; done
done
