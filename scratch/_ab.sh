set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_c5.py tests/test_gpu_camshift.py -m gpu -q --no-header -p no:cacheprovider -x > $OUT/ab_pytest.log 2>&1; echo "pytest exit $?"; grep "passed\|failed" $OUT/ab_pytest.log | tail -1
timeout 300 python bench.py --workload c5 --feeds 8 --cpu-seconds 0 --no-sub > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "bench c5 exit $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_c5.json'))
print({k:d[k] for k in ('value','ms_per_step','parity_exact','parity_detect_exact','device_ms') if k in d})
PY
