#!/bin/bash
# per-launch kernel durations of one bench step (rocprofv3 kernel trace), to see the launch structure of the pyramid
export TMPDIR=/tmp; WL=${1:-c2}
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 3 --warmup 1 --cpu-seconds 0 --prewarm 0 --pipeline 1 > /dev/null 2>&1
python - <<PY
import csv,glob,re
f=glob.glob('/tmp/tr/**/t_kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ks=[(re.search(r'(k_\w+|__amd\w+)',r['Kernel_Name']).group(1) if re.search(r'(k_\w+|__amd\w+)',r['Kernel_Name']) else r['Kernel_Name'][:20], int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Grid_Size_X') or r.get('Grid_Size')) for r in rows]
# last step: find last k_gray
gi=[i for i,k in enumerate(ks) if k[0].startswith('k_gray')]
i0=gi[-2]; t0=ks[i0][1]
for k in ks[i0:gi[-1]]:
    print(f"{k[0]:28s} start {(k[1]-t0)/1000:9.1f} us  dur {(k[2]-k[1])/1000:8.1f} us  grid {k[3]}")
PY
