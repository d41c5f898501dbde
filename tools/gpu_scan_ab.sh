#!/bin/bash
# tools/gpu_scan_ab.sh [options...] — detect parity tests, then device ms per kernel + wall per step (3 in flight) for each options string, C2 and C4,
# and one PMC pass (SQ_INSTS_VALU / SALU / LDS of k_scan_tiles) per options string at C2
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_detect.py tests/test_gpu_shapes.py tests/test_gpu_custom_cascade.py tests/test_gpu_sizes.py -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest3.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest3.log
for o in "$@"; do
  oo=$o; [ "$o" = none ] && oo=""
  timeout 200 python tools/gpu_kernel_times.py c2 "$oo" 3 2>/dev/null | tail -1
  timeout 200 python tools/gpu_kernel_times.py c4 "$oo" 2 2>/dev/null | tail -1
done
if [ "${PMC:-1}" = 1 ]; then
cd /tmp
for o in "$@"; do
  oo=$o; [ "$o" = none ] && oo="fp_sparse=1"
  for wl in c2 c4; do
  rm -rf $OUT/pmcab_${o}_$wl
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmcab_${o}_$wl -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --cpu-seconds 0 --prewarm 0 --pipeline 1 --no-sub --rounds 3 --options "$oo" > $OUT/pmcab_${o}_$wl.log 2>&1
  python - <<PY
import csv,glob,collections,re
fs=glob.glob("$OUT/pmcab_${o}_$wl/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(fs[0])):
    m=re.search(r"(k_\w+)", r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:30]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_INSTS_VALU": cnt[k]+=1
for k,v in agg.items():
    if "scan" in k: print("$o $wl", k, "launches", cnt[k], {c: round(x/max(cnt[k],1)/1e6,2) for c,x in v.items()})
PY
  find $OUT/pmcab_${o}_$wl -name "*.csv" -size +1M -delete
  done
done
fi
