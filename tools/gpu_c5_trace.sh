#!/bin/bash
# tools/gpu_c5_trace.sh — rocprofv3 kernel trace of the C5 loop (8 feeds) for two option sets; prints a per-kernel timeline summary
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for o in "$@"; do
  rm -rf $OUT/tr_$o
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$o -o t -- python $GRAFT_REPO_ROOT/bench.py --workload c5 --feeds 8 --steps 60 --rounds 3 --cpu-seconds 0 --options $o --no-sub > $OUT/tr_$o.log 2>&1
  echo "trace $o exit $?"
  python - <<PY
import csv,glob,collections,re
fs=glob.glob("$OUT/tr_$o/**/*kernel_trace.csv", recursive=True)
rows=[r for r in csv.DictReader(open(fs[0]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# steady-state window: the last 400 camshift kernels
cs=[r for r in rows if "k_cs_" in r["Kernel_Name"]]
tail=cs[-600:]
t0=int(tail[0]["Start_Timestamp"]); t1=int(tail[-1]["End_Timestamp"])
dur=collections.defaultdict(list)
for r in tail:
    m=re.search(r"(k_cs_\w+)", r["Kernel_Name"]); dur[m.group(1)].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("$o window us", (t1-t0)/1e3, {k:(len(v), round(sum(v)/len(v)/1e3,2)) for k,v in dur.items()})
# union busy time and first 24 kernels as a timeline
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in tail)
busy=0; ce=ev[0][0]
for s,e in ev:
    if e>ce: busy+= e-max(s,ce); ce=e
print("busy frac", round(busy/(t1-t0),3))
for r in tail[300:318]:
    m=re.search(r"(k_cs_\w+)", r["Kernel_Name"]); print(m.group(1), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Queue_Id"))
PY
  find $OUT/tr_$o -name "*.csv" -size +1M -delete
done
