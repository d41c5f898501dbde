#!/bin/bash
# tools/gpu_cs_one_stream_trace.sh [W H N] — rocprofv3 kernel trace of camshift track() calls of N streams issued strictly in turn (the drop-in
# tracker's common call): per kernel of a call the median duration, and the wall clock per call.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
W=${1:-320}; H=${2:-240}; N=${3:-1}
cat > /tmp/cs_loop.py <<PY
import sys, time, numpy as np, torch
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from headtrackr_amd import synth
from headtrackr_amd.api import Context
W, H, n = $W, $H, $N
fr = np.stack([synth.face_frame(W, H, [(W // 3 + 5 * i, H // 4, min(W, H) // 3)]) for i in range(n)])
dev = torch.from_numpy(fr).cuda()
c = Context(options="${OPTS:-}" or None)
c.set_geometry(W, H, n)
c.bind_device(dev.data_ptr(), n)
c.camshift_reserve(n)
rects = np.zeros(n, dtype=[("x", "<i4"), ("y", "<i4"), ("width", "<i4"), ("height", "<i4")])
rects["x"], rects["y"], rects["width"], rects["height"] = W // 3, H // 4, min(W, H) // 3, min(W, H) // 3
c.camshift_init(rects)
lat = []
for i in range(120):
    t0 = time.perf_counter()
    c.camshift_track(n, calc_angles=True)
    lat.append((time.perf_counter() - t0) * 1e6)
print(f"wall per track() call of {n} stream(s) at {W}x{H}: p50 {np.percentile(lat[20:], 50):.1f} us")
c.close()
PY
cd /tmp
rm -rf $OUT/trace_cs
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_cs -o t -- python /tmp/cs_loop.py 2>/dev/null | grep "wall per"
python - <<PY
import csv, glob, re, collections, statistics as st
fs = glob.glob("$OUT/trace_cs/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_cs" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
name = lambda r: (re.search(r"(k_\w+)", r["Kernel_Name"]) or re.search(r"(__amd_rocclr_\w+)", r["Kernel_Name"])).group(1)
rows = rows[len(rows) // 3:]
by = collections.defaultdict(list)
for r in rows:
    by[name(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in by.items():
    print(f"  {k:32s} {len(v):4d} launches, median {st.median(v):6.2f} us")
PY
find $OUT/trace_cs -name "*.csv" -size +1M -delete
