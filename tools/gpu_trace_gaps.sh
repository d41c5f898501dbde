#!/bin/bash
# tools/gpu_trace_gaps.sh [c2|c4] [pipeline] — rocprofv3 kernel trace of a short bench run: per step, how much of the time between the first
# kernel's start and the last kernel's end is NOT covered by any kernel (launch gaps between dependent kernels), and the mean gap after each kernel.
WL=${1:-c2}; PL=${2:-1}
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd /tmp
rm -rf /tmp/trace_gaps
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_gaps -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 40 --warmup 10 --cpu-seconds 0 --prewarm 0 --pipeline $PL --no-sub > /tmp/trace_gaps.log 2>&1
python - <<PY
import csv,glob,re,collections
f=glob.glob("/tmp/trace_gaps/**/*kernel_trace.csv", recursive=True)[0]
rows=[]
for r in csv.DictReader(open(f)):
    m=re.search(r"(k_\w+)", r["Kernel_Name"])
    if not m: continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Stream_Id") or r.get("Queue_Id")))
rows.sort()
rows=rows[len(rows)//3:]   # steady state
t0,t1=rows[0][0],rows[-1][1]
# union coverage
cov=0; cur_s,cur_e=rows[0][0],rows[0][1]
for s,e,_,_ in rows[1:]:
    if s>cur_e: cov+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
cov+=cur_e-cur_s
print(f"$WL pipeline $PL: {len(rows)} kernels over {(t1-t0)/1e3:.1f} us; covered by at least one kernel {100*cov/(t1-t0):.1f} %; sum of kernel durations {sum(e-s for s,e,_,_ in rows)/1e3:.1f} us")
gap=collections.defaultdict(list); dur=collections.defaultdict(list)
by_stream=collections.defaultdict(list)
for r in rows: by_stream[r[3]].append(r)
for q,rs in by_stream.items():
    for a,b in zip(rs,rs[1:]):
        gap[a[2]+" -> "+b[2]].append(b[0]-a[1])
    for r in rs: dur[r[2]].append(r[1]-r[0])
for k,v in sorted(gap.items(), key=lambda kv:-len(kv[1]))[:12]:
    v=sorted(v); print(f"  gap {k:40s} n={len(v):4d} median {v[len(v)//2]/1e3:7.2f} us  mean {sum(v)/len(v)/1e3:7.2f} us")
for k,v in dur.items(): print(f"  duration {k:24s} n={len(v):4d} mean {sum(v)/len(v)/1e3:7.2f} us")
PY
