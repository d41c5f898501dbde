#!/bin/bash
# tools/gpu_pmc.sh — rocprofv3 PMC passes (counters only + kernel trace, one counter set per run) for bench.py
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
WL=${PMC_WL:-c2}
# the build the counters belong to (benchlib/fingerprint.py): tools/collect_profiles.py stores it as traffic.json's _build
python -c "import json,sys; sys.path.insert(0,'$GRAFT_REPO_ROOT'); from benchlib import fingerprint as f; json.dump(f.code_objects(), open('$OUT/pmc_build_${WL}.json','w'))"
cd /tmp
i=0
for set in "${@:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_${WL}_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps $( [ $WL = c3 ] && echo 1 || echo 3 ) --warmup 1 --cpu-seconds 0 --prewarm 0 --pipeline 1 --no-sub > $OUT/pmc_${WL}_$i.log 2>&1
  echo "set $i: $set -> exit $?"
  python - <<PY
import csv,glob,collections
fs=glob.glob("$OUT/pmc_${WL}_$i/**/*counter_collection.csv", recursive=True)
if not fs: print("no counter csv"); raise SystemExit
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(fs[0])):
    import re
    m=re.search(r"(k_\w+)", r["Kernel_Name"]); k=m.group(1) if m else r["Kernel_Name"][:30]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="$(echo $set | cut -d' ' -f1)": cnt[k]+=1
for k,v in agg.items():
    n=max(cnt[k],1)
    print(k, "launches",n, {c: round(x/n) for c,x in v.items()})
PY
done
find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*counter_collection.csv" -size +4M -delete
