#!/bin/bash
# tools/gpu_pmc_kernel.sh WORKLOAD KERNEL_REGEX — rocprofv3 PMC passes (counters only + kernel trace) of tools/gpu_gen_times.py, per-launch averages of one kernel
WL=$1; KR=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmck_$i -o p -- python $GRAFT_REPO_ROOT/tools/gpu_gen_times.py $WL > $OUT/pmck_$i.log 2>&1
  python - <<PY
import csv,glob,collections,re
fs=glob.glob("$OUT/pmck_$i/**/*counter_collection.csv", recursive=True)
if not fs: print("set $i: no counter csv"); raise SystemExit
agg=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(fs[0])):
    if re.search(r"$KR", r["Kernel_Name"]):
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print({c: round(x/max(n[c],1)) for c,x in agg.items()}, "launches", max(n.values()) if n else 0)
PY
done
rm -rf $OUT/pmck_*
