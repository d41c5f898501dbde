"""how long is the deep kernel's critical path?  detect on n copies of one face frame and time the kernels"""
import sys
import numpy as np
sys.path.insert(0, ".")
from headtrackr_amd import synth
from headtrackr_amd.api import Context
f = synth.face_frame(320, 240, [(100, 60, 96)])
for n in (1, 8, 64, 256):
    frames = np.stack([f] * n)
    ctx = Context(); ctx.set_geometry(320, 240, n); ctx.upload(frames)
    for _ in range(3):
        ctx.detect_enqueue(16); hits, _ = ctx.detect_collect()
    sc = ctx.stage_counts()
    ctx.profile(True); ctx.kernel_times(True)
    K = 20
    for _ in range(K):
        ctx.detect_enqueue(0); ctx.detect_collect()
    kt = ctx.kernel_times(True)
    print(n, "frames: hits", len(hits), "stage8 entries", int(sc[8]), "stage15", int(sc[15]), {k: round(v["ms"] / K * 1e3, 1) for k, v in kt.items()}, "us")
    ctx.close()
