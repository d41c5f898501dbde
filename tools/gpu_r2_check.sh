#!/bin/bash
# tools/gpu_r2_check.sh — one gpurun call: GPU parity tests, smoke, the default bench line (+ optional c5), timing of each.
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? ($(( $(date +%s) - t0 )) s)"; tail -15 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
t0=$(date +%s)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $? ($(( $(date +%s) - t0 )) s)"; tail -5 $OUT/bench_default.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
    print("c2", d["value"], "fps", d["ms_per_step"], "ms", d["kernel_ms_per_step"], "frac", d["roofline"]["frac"], "path", d["path_hbm_frac"], "cpu", d["cpu_baseline"]["value"], "wall", d["bench_wall_s"])
    for k, v in d.get("sub", {}).items():
        print(k, v["value"], "fps", v["ms_per_step"], "ms", v.get("kernel_ms_per_step") or v.get("kernel_ms_per_track_call"), "frac", v["roofline"]["frac"], "cpu", (v.get("cpu_baseline") or {}).get("value"), "vs_cpu", v.get("vs_cpu") or v.get("vs_cpu_track_calls"))
except Exception as e:
    print("bench parse failed:", e)
PY
if [ "${RUN_C5:-1}" = 1 ]; then
  timeout 600 python bench.py --workload c5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 exit $?"; cut -c1-900 $OUT/bench_c5.json; tail -3 $OUT/bench_c5.err
fi
