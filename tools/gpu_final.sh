#!/bin/bash
# tools/gpu_final.sh [TAG] — the round's measurement pass on the GPU box: parity tests, the driver's bench line (all sub-records), the
# flag-less line, rocprofv3 kernel stats per workload and PMC passes (SQ sets, FETCH_SIZE, WRITE_SIZE — each in its own run, counters
# only + kernel trace), host post-processing times, shader-clock timelines of the two big kernels (instrumented builds from
# tools/build_alt.py, loaded through HEADTRACKR_HIP_LIB: the product library is never replaced).
# tools/collect_profiles.py turns gpurun_out/ into the committed summaries under profiles/.
set -u
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
t0=$(date +%s)
if [ "${RUN_TESTS:-1}" = 1 ]; then
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? ($(( $(date +%s) - t0 )) s)"; grep -E "passed|failed|camshift parity" $OUT/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
fi
t0=$(date +%s)
# the command the driver runs at round end (BENCH_rNN.json: 20 timed steps = a 5 ms block for the primary workload)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench driver-style exit $? ($(( $(date +%s) - t0 )) s, $(wc -c < $OUT/bench_driver.json) bytes)"; cut -c1-300 $OUT/bench_driver.json
cp $OUT/bench_sub.json $OUT/bench_driver_sub.json 2>/dev/null  # the full record tree of the same run
timeout 900 python bench.py --no-sub --cpu-seconds 0 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench flag-less (c2 only) exit $?"; cut -c1-200 $OUT/bench_default.json
timeout 120 python tools/gpu_host_post.py > $OUT/host_post.txt 2> $OUT/host_post.err; tail -3 $OUT/host_post.txt
for o in "" host_threads=0; do timeout 120 python tools/gpu_kernel_times.py c2 $o 2>/dev/null | tail -1 >> $OUT/kernel_times.txt; done
timeout 120 python tools/gpu_kernel_times.py c4 2>/dev/null | tail -1 >> $OUT/kernel_times.txt; cat $OUT/kernel_times.txt
for f in 8 1; do timeout 200 python tools/gpu_cs_step.py $f 2>/dev/null | tail -1 >> $OUT/cs_step.txt; done; cat $OUT/cs_step.txt
timeout 300 bash tools/gpu_c5_trace.sh cs_flags=1 > $OUT/c5_trace.txt 2>&1; tail -3 $OUT/c5_trace.txt
# single-frame / single-stream calls in turn: kernel traces (rocprofv3 --kernel-trace only) and wall clocks
{ for g in "320 240 1" "1280 720 1" "1920 1080 1" "1920 1080 8"; do timeout 200 bash tools/gpu_one_frame_trace.sh $g; done; } > $OUT/one_frame_trace.txt 2>&1
{ timeout 200 ./tools/micro/launch_chain_bench; } > $OUT/launch_chain.txt 2>&1
{ timeout 200 python tools/gpu_cs_wall.py; timeout 200 python tools/gpu_cs_wall.py cs_sync_ring=0; for g in "320 240 1" "640 480 1" "1920 1080 1"; do timeout 200 bash tools/gpu_cs_one_stream_trace.sh $g; done; } > $OUT/cs_wall.txt 2>/dev/null
if [ "${RUN_PROF:-1}" = 1 ]; then
cd /tmp
for wl in c2 c4 c3 c5; do
  ST=8; [ $wl = c3 ] && ST=2; [ $wl = c5 ] && ST=60
  FE=""; [ $wl = c5 ] && FE="--feeds 8"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$wl -o $wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl $FE --steps $ST --warmup 2 --cpu-seconds 0 --prewarm 0 --pipeline 1 --no-sub > $GRAFT_REPO_ROOT/$OUT/prof_$wl.log 2>&1
  echo "rocprofv3 stats $wl exit $?"
done
cd $GRAFT_REPO_ROOT
find $OUT -name "*kernel_trace.csv" -size +1M -delete
for wl in c2 c4 c3; do
  PMC_WL=$wl bash tools/gpu_pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" > $OUT/pmc_$wl.txt 2>&1
  echo "pmc $wl done"
done
fi
# shader-clock phase timelines of the two big kernels (alt/tltl.so = -DHT_TILE_TIMELINE, alt/rsph.so = -DHT_RS_PHASES, built by
# tools/build_alt.py from the CURRENT sources: a stale variant is refused, stderr never lands in the evidence files)
for pair in "tltl gpu_tile_timeline tile_timeline" "rsph gpu_rsb_phases rs_phases"; do
  set -- $pair
  if python tools/build_alt.py --check $1 > /dev/null; then
    for wl in c2 c4; do
      HEADTRACKR_HIP_LIB=$GRAFT_REPO_ROOT/alt/$1.so HT_RS_WAVESTAMPS=1 timeout 300 python tools/$2.py $wl > $OUT/$3_$wl.txt 2> $OUT/$3_$wl.err || { echo "$2 $wl FAILED (see $OUT/$3_$wl.err)"; rm -f $OUT/$3_$wl.txt; }
    done
  else
    echo "alt/$1.so is stale or missing: run  python tools/build_alt.py $1 ...  first; no timeline written"
  fi
done
echo done
