#!/bin/bash
# frames per k_resample workgroup (HT_DEBUG_RS_K), any value: packing of the launches into the 1536 workgroup slots
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('kernel_ms_per_step'))"; }
for wl in ${WLS:-c2 c4}; do
ST=300; [ $wl = c4 ] && ST=80
for k in ${KS:-0 1 2 3 4 5 6 7 8 9 10 11 12 16}; do
 HT_DEBUG_RS_K=$k python bench.py --workload $wl --steps $ST --cpu-seconds 0 --no-sub 2>/dev/null | pr "K=$k $wl"
done; done
