#!/bin/bash
# tools/gpu_final.sh — the round's measurement pass on the GPU box: parity tests, bench lines for every workload,
# rocprofv3 kernel stats and PMC passes (SQ set, FETCH_SIZE, WRITE_SIZE separately).  tools/collect_profiles.py turns
# gpurun_out/ into the committed summaries under profiles/.
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
for wl in c2 c4 c3 c5; do
  timeout 600 python bench.py --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "bench $wl exit $?"; cut -c1-420 $OUT/bench_$wl.json
done
timeout 300 python bench.py --workload c2 --pipeline 1 --cpu-seconds 0 > $OUT/bench_c2_p1.json 2>/dev/null
cd /tmp
for wl in c2 c4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$wl -o $wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 8 --warmup 2 --cpu-seconds 0 --prewarm 0 --pipeline 1 > $GRAFT_REPO_ROOT/$OUT/prof_$wl.log 2>&1
done
cd $GRAFT_REPO_ROOT
find $OUT -name "*kernel_trace.csv" -size +1M -delete
for wl in c2 c4; do
  PMC_WL=$wl bash tools/gpu_pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" > $OUT/pmc_$wl.txt 2>&1
done
echo done
