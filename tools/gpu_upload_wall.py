"""tools/gpu_upload_wall.py — wall clock of ht_upload_frames (pageable host memory, as the drop-in tracker's canvas data) alone and followed by a
synchronous camshift track() call, one 320x240 / 640x480 / 1920x1080 frame."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from headtrackr_amd import synth  # noqa: E402
from headtrackr_amd.api import Context  # noqa: E402

for (W, H) in ((320, 240), (640, 480), (1920, 1080)):
    fr = np.stack([synth.face_frame(W, H, [(W // 3, H // 4, min(W, H) // 3)])])
    c = Context(options=(sys.argv[1] if len(sys.argv) > 1 else None) or None)
    c.set_geometry(W, H, 1)
    c.upload(fr)
    c.camshift_reserve(1)
    rects = np.zeros(1, dtype=[("x", "<i4"), ("y", "<i4"), ("width", "<i4"), ("height", "<i4")])
    rects["x"], rects["y"], rects["width"], rects["height"] = W // 3, H // 4, min(W, H) // 3, min(W, H) // 3
    c.camshift_init(rects)
    up, both = [], []
    for i in range(300):
        t0 = time.perf_counter()
        c.upload(fr)
        t1 = time.perf_counter()
        c.camshift_track(1, calc_angles=True)
        t2 = time.perf_counter()
        up.append((t1 - t0) * 1e6)
        both.append((t2 - t0) * 1e6)
    print(f"{W}x{H}: ht_upload_frames p50 {np.percentile(up[50:], 50):.1f} us ({fr.nbytes / 1e3:.0f} KB), upload + track() p50 {np.percentile(both[50:], 50):.1f} us")
    c.close()
