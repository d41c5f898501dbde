#!/bin/bash
# tools/gpu_pmc_ab.sh "LIB|OPTIONS" ...  — one rocprofv3 PMC pass (SQ_INSTS_VALU / SALU / LDS, SQ_BUSY_CYCLES; counters + kernel trace only) per
# entry and workload (WLS, default "c2 c4"): wave instructions per launch of every kernel, in millions.  LIB = "-" or a name under alt/.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for e in "$@"; do
  lib=${e%%|*}; opt=${e#*|}
  if [ "$lib" = "-" ]; then unset HEADTRACKR_HIP_LIB; else export HEADTRACKR_HIP_LIB=$GRAFT_REPO_ROOT/alt/$lib.so; fi
  for wl in ${WLS:-c2 c4}; do
    d=$OUT/pmcab_${lib}_$wl; rm -rf $d
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --cpu-seconds 0 --prewarm 0 --pipeline 1 --no-sub --rounds 3 --options "$opt" > $d.log 2>&1
    python - <<PY
import csv,glob,collections,re
fs=glob.glob("$d/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(fs[0])) if fs else []:
    m=re.search(r"(k_\w+)", r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:30]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_INSTS_VALU": cnt[k]+=1
for k,v in sorted(agg.items()):
    print("[$lib|$opt] $wl", k, "launches", cnt[k], {c: round(x/max(cnt[k],1)/1e6,3) for c,x in sorted(v.items())})
PY
    find $d -name "*.csv" -size +1M -delete
  done
done | tee $OUT/pmc_ab.txt
