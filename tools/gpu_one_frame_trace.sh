#!/bin/bash
# tools/gpu_one_frame_trace.sh [W H N] — rocprofv3 kernel trace of single-frame (N-frame) detect calls issued strictly in turn (graph replay as shipped):
# per kernel the median duration and the median gap to the previous kernel of the same call, from the trace's own timestamps (no events).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
W=${1:-320}; H=${2:-240}; N=${3:-1}
cat > /tmp/one_frame_loop.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from headtrackr_amd import synth
from headtrackr_amd.api import Context
W, H, n = $W, $H, $N
fr = np.stack([synth.face_frame(W, H, [(W // 3 + 5 * i, H // 4, min(W, H) // 3)]) for i in range(n)])
dev = torch.from_numpy(fr).cuda()
c = Context(options="${OPTS:-}" or None)
c.set_geometry(W, H, n)
c.bind_device(dev.data_ptr(), n)
for _ in range(60):
    c.detect_enqueue(0)
    c.detect_collect_best(1)
c.close()
PY
cd /tmp
rm -rf $OUT/trace1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace1 -o t -- python /tmp/one_frame_loop.py > $OUT/trace1.log 2>&1
python - <<PY
import csv, glob, re, collections, statistics as st
fs = glob.glob("$OUT/trace1/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_" in r["Kernel_Name"]]
name = lambda r: re.search(r"(k_\w+)", r["Kernel_Name"]).group(1)
# calls: a call starts with k_gray_linear
calls, cur = [], []
for r in rows:
    if name(r).startswith("k_gray") and cur:
        calls.append(cur); cur = []
    cur.append(r)
calls.append(cur)
calls = calls[20:]  # warm
seqlen = collections.Counter(len(c) for c in calls).most_common(1)[0][0]
calls = [c for c in calls if len(c) == seqlen]
print(f"${W}x${H} n=${N}: {len(calls)} calls of {seqlen} kernels")
tot = []
for i in range(seqlen):
    d = [ (int(c[i]["End_Timestamp"]) - int(c[i]["Start_Timestamp"])) / 1e3 for c in calls]
    g = [ (int(c[i]["Start_Timestamp"]) - int(c[i-1]["End_Timestamp"])) / 1e3 for c in calls] if i else [0.0]
    print(f"  {name(calls[0][i]):28s} duration {st.median(d):6.2f} us   gap before {st.median(g):6.2f} us   grid {calls[0][i].get('Grid_Size_X', calls[0][i].get('Grid_Size','?'))}")
span = [ (int(c[-1]["End_Timestamp"]) - int(c[0]["Start_Timestamp"])) / 1e3 for c in calls]
print(f"  first kernel start -> last kernel end: {st.median(span):.1f} us")
PY
find $OUT/trace1 -name "*.csv" -size +1M -delete
