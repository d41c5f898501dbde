set -u
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_camshift.py tests/test_gpu_shapes.py -q -x -k "camshift or c3 or fused or golden or batch_of or frame_sizes or facetrackr" -p no:cacheprovider 2>&1 | tail -3
for d in 2 3 4 1; do
  timeout 200 python bench.py --workload c3 --no-sub --cpu-seconds 0 --pipeline $d 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('c3 steps in flight $d:', j['value'], 'frames/s', j['ms_per_step'], 'ms/step', j.get('parity_exact'), j['roofline'].get('kernel_ms_per_step'), j['roofline'].get('frac'))"
done
python - <<'PY'
import json
j = json.load(open("gpurun_out/bench_sub.json")) if False else None
PY
