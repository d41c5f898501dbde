"""tools/gpu_vj_step_wall.py [options] — wall clock of the drop-in tracker's two per-frame calls on one 320x240 frame from pageable host memory:
VJ step = upload + detect (best face) + initTracker, CS step = upload + synchronous track()."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from headtrackr_amd import synth  # noqa: E402
from headtrackr_amd.api import Context  # noqa: E402

W, H = 320, 240
fr = np.stack([synth.face_frame(W, H, [(W // 3, H // 4, min(W, H) // 3)])])
c = Context(options=(sys.argv[1] if len(sys.argv) > 1 else None) or None)
c.set_geometry(W, H, 1)
c.camshift_reserve(4)
rects = np.zeros(1, dtype=[("x", "<i4"), ("y", "<i4"), ("width", "<i4"), ("height", "<i4")])
rects["x"], rects["y"], rects["width"], rects["height"] = W // 3, H // 4, min(W, H) // 3, min(W, H) // 3
parts = {"upload": [], "detect": [], "init": [], "track": []}
for i in range(1500):
    vj = i % 30 == 0
    t0 = time.perf_counter()
    c.upload(fr)
    t1 = time.perf_counter()
    if vj:
        c.detect_enqueue(0)
        c.detect_collect_best(1)
        if os.environ.get("VJ_SYNC"):
            c.synchronize()
        t2 = time.perf_counter()
        c.camshift_init(rects)
        t3 = time.perf_counter()
        if os.environ.get("VJ_SYNC2"):
            c.synchronize()
        if i >= 60:
            parts["detect"].append((t2 - t1) * 1e6), parts["init"].append((t3 - t2) * 1e6)
    else:
        c.camshift_track(1, calc_angles=True)
        t2 = time.perf_counter()
        if i >= 60:
            parts["track"].append((t2 - t1) * 1e6)
    if i >= 60:
        parts["upload"].append((t1 - t0) * 1e6)
print(f"options={sys.argv[1] if len(sys.argv) > 1 else None}: " + "  ".join(f"{k} p50 {np.percentile(v, 50):.1f} us" for k, v in parts.items()))
c.close()
