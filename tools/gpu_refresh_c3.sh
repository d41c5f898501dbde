#!/bin/bash
# tools/gpu_refresh_c3.sh — after a change that only touches the camshift kernels: parity suite, the driver's bench line, the
# flag-less line, C3 at 2 / 3 / 4 steps in flight, rocprofv3 kernel stats + PMC passes of C3 (both forms of k_cs_track_fused)
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? ($(( $(date +%s) - t0 )) s)"; grep -E "passed|failed|camshift parity" $OUT/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench driver-style exit $? ($(( $(date +%s) - t0 )) s, $(wc -c < $OUT/bench_driver.json) bytes)"; cut -c1-200 $OUT/bench_driver.json
cp $OUT/bench_sub.json $OUT/bench_driver_sub.json 2>/dev/null
timeout 900 python bench.py --no-sub --cpu-seconds 0 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench flag-less (c2 only) exit $?"; cut -c1-200 $OUT/bench_default.json
for d in 2 3 4; do
  timeout 200 python bench.py --workload c3 --no-sub --cpu-seconds 0 --pipeline $d 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('c3 steps in flight $d:', j['value'], 'frames/s', j['ms_per_step'], 'ms/step', j.get('parity_exact'))" | tee -a $OUT/c3_depth.txt
done
rm -f $OUT/cs_step.txt; for f in 8 1; do timeout 200 python tools/gpu_cs_step.py $f 2>/dev/null | tail -1 >> $OUT/cs_step.txt; done; cat $OUT/cs_step.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_c5 -o c5 -- python $GRAFT_REPO_ROOT/bench.py --workload c5 --feeds 8 --steps 60 --warmup 2 --cpu-seconds 0 --prewarm 0 --pipeline 1 --no-sub > $GRAFT_REPO_ROOT/$OUT/prof_c5.log 2>&1
echo "rocprofv3 stats c5 exit $?"
for form in 1024 512; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_c3_$form -o c3 -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 2 --warmup 2 --cpu-seconds 0 --prewarm 0 --pipeline 1 --no-sub --options cs_fused_nt=$form > $GRAFT_REPO_ROOT/$OUT/prof_c3_$form.log 2>&1
  echo "rocprofv3 stats c3 form $form exit $?"
done
cd $GRAFT_REPO_ROOT
rm -rf $OUT/prof_c3; cp -r $OUT/prof_c3_1024 $OUT/prof_c3
find $OUT -name "*kernel_trace.csv" -size +1M -delete
PMC_WL=c3 bash tools/gpu_pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" > $OUT/pmc_c3.txt 2>&1
echo "pmc c3 done"; echo done
