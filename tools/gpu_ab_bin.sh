#!/bin/bash
# A/B of the two cs_bin forms (alt/binf.so = field by field) on one box: C3 with both forms of the fused kernel forced, three steps in flight
set -u
for rep in 1 2; do
for lib in product binf; do
for o in "cs_fused_nt=1024" "cs_fused_nt=512" ""; do
  if [ $lib = binf ]; then export HEADTRACKR_HIP_LIB=$GRAFT_REPO_ROOT/alt/binf.so; else unset HEADTRACKR_HIP_LIB; fi
  timeout 200 python bench.py --workload c3 --no-sub --cpu-seconds 0 --options "$o" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$lib c3 options=$o:', j['value'], 'frames/s', j['ms_per_step'], 'ms/step', j.get('parity_exact'), j['roofline'].get('kernel_ms_per_step'), j['roofline'].get('frac'))"
done
done
done
