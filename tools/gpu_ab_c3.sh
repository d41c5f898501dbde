#!/bin/bash
# camshift parity on both forms of the fused kernel + A/B of the C3 bench (three steps in flight) on one box
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_camshift.py tests/test_gpu_shapes.py -q -x -k "camshift or c3 or fused or golden or batch_of or frame_sizes or facetrackr" -p no:cacheprovider 2>&1 | tail -4
for rep in 1 2; do
for o in "cs_fused_nt=1024" "cs_fused_nt=512" ""; do
  timeout 200 python bench.py --workload c3 --no-sub --cpu-seconds 0 --options "$o" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('c3 options=$o:', j['value'], 'frames/s', j['ms_per_step'], 'ms/step', j.get('parity_exact'), j['roofline'].get('kernel_ms_per_step'), j['roofline'].get('frac'))"
done
done
