#!/bin/bash
# tools/gpu_quick.sh — parity suite + the driver's bench command on the GPU box (a few minutes); logs into gpurun_out/
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=8 -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $? ($(( $(date +%s) - t0 )) s)"; tail -15 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
echo "bench driver-style exit $? ($(( $(date +%s) - t0 )) s, $(wc -c < $OUT/bench_driver.json) bytes)"; cat $OUT/bench_driver.json
tail -3 $OUT/bench_driver.err
