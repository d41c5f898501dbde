#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python -m pytest tests/test_gpu_shapes.py -m gpu -q -x --no-header -p no:cacheprovider -k c3_shape 2>&1 | grep -E "^E |passed|failed" | head -8; done
cd /tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/pmc_cs
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_cs -o p -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 2 --warmup 1 --cpu-seconds 0 --no-sub > /tmp/pmc_cs.log 2>&1
  echo "rocprof exit $?"; tail -2 /tmp/pmc_cs.log; find /tmp/pmc_cs -name "*.csv" | head -3
  python - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/pmc_cs/**/*counter_collection.csv", recursive=True)
print(fs)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
first="$set".split()[0]
for r in csv.DictReader(open(fs[0])):
    k=r["Kernel_Name"].split("(")[0][-30:]
    if "k_cs" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]==first: cnt[k]+=1
for k,v in agg.items():
    n=max(cnt[k],1); print(k,"launches",n,{c: round(x/n) for c,x in v.items()})
PY
done
