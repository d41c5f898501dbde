#!/bin/bash
# camshift parity subset + C3 (three steps in flight, both forced forms and the automatic choice) + the C5 track step, on one box
set -u
timeout 600 python -m pytest tests/test_gpu_camshift.py tests/test_gpu_shapes.py -q -x -k "camshift or c3 or fused or golden or batch_of or frame_sizes or facetrackr" -p no:cacheprovider 2>&1 | tail -3
for o in "" "cs_fused_nt=1024" "cs_fused_nt=512"; do
  timeout 200 python bench.py --workload c3 --no-sub --cpu-seconds 0 --options "$o" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('c3 options=$o:', j['value'], 'frames/s', j['ms_per_step'], 'ms/step', j.get('parity_exact'), j['roofline'].get('kernel_ms_per_step'), j['roofline'].get('frac'))"
done
for f in 8 1; do timeout 200 python tools/gpu_cs_step.py $f 2>/dev/null | tail -1; done
