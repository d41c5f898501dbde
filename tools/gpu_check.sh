#!/bin/bash
# tools/gpu_check.sh — one gpurun call: environment probe, GPU parity tests, bench line, rocprofv3 kernel stats.
# Everything of interest lands in gpurun_out/ (merged back by gpurun).
set -u
OUT=gpurun_out
mkdir -p $OUT
{
  echo "== env"; date; nproc; node --version 2>&1; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -6
  python -c "import torch;print('torch',torch.__version__,torch.cuda.is_available(),torch.cuda.device_count())"
} > $OUT/env.log 2>&1
export TMPDIR=/tmp
echo "== pytest gpu" > $OUT/pytest_gpu.log
timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider ${PYTEST_ARGS:-} >> $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== bench c2"; timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "exit $?"; cat $OUT/bench_c2.json; tail -5 $OUT/bench_c2.err
for fl in ${AB_FLAGS:-}; do
  echo "== bench c2 flags=$fl"; timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --flags $fl > $OUT/bench_c2_f$fl.json 2> $OUT/bench_c2_f$fl.err; echo "exit $?"; cat $OUT/bench_c2_f$fl.json; tail -3 $OUT/bench_c2_f$fl.err
done
if [ "${RUN_C3:-0}" = "1" ]; then
  echo "== bench c3"; timeout 600 python bench.py --workload c3 --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "exit $?"; cat $OUT/bench_c3.json; tail -5 $OUT/bench_c3.err
fi
if [ "${RUN_C4:-1}" = "1" ]; then
  echo "== bench c4"; timeout 600 python bench.py --workload c4 --steps 5 --warmup 2 --cpu-seconds 6 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "exit $?"; cat $OUT/bench_c4.json; tail -5 $OUT/bench_c4.err
fi
if [ "${RUN_PROF:-1}" = "1" ]; then
  echo "== rocprofv3 kernel stats (c2)"
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_c2 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$OUT/prof_c2.log 2>&1
  cd $GRAFT_REPO_ROOT; ls -R $OUT/prof_c2 | head -20
  find $OUT/prof_c2 -name "*kernel_stats*" | head -1 | xargs -r head -20
  find $OUT/prof_c2 -name "*kernel_trace*" -size +2M -delete
fi
