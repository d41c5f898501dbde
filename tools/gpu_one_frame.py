#!/usr/bin/env python3
"""tools/gpu_one_frame.py [options] — detect of ONE resident frame (and of 8) at 320x240 / 1280x720 / 1920x1080: device us per kernel and
launches (HIP events, graph replay off), and the wall clock per enqueue + collect_best call with graph replay."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from headtrackr_amd import synth  # noqa: E402
from headtrackr_amd.api import Context  # noqa: E402

opts = (sys.argv[1] if len(sys.argv) > 1 else None) or None
for (W, H) in ((320, 240), (1280, 720), (1920, 1080)):
    for n in (1, 8):
        fr = np.stack([synth.face_frame(W, H, [(W // 3 + 5 * i, H // 4, min(W, H) // 3)]) for i in range(n)])
        dev = torch.from_numpy(fr).cuda()
        c = Context(options=opts)
        c.set_geometry(W, H, n)
        c.bind_device(dev.data_ptr(), n)
        for _ in range(10):
            c.detect_enqueue(0)
            c.detect_collect_best(1)
        lat = []
        for _ in range(200):
            t0 = time.perf_counter()
            c.detect_enqueue(0)
            c.detect_collect_best(1)
            lat.append((time.perf_counter() - t0) * 1e6)
        c.profile(True)
        c.kernel_times(reset=True)
        K = 20
        for _ in range(K):
            c.detect_enqueue(0)
            c.detect_collect_best(1)
        kt = c.kernel_times(reset=True)
        c.profile(False)
        per = {k: (round(v["ms"] / K * 1e3, 1), v["launches"] // K) for k, v in kt.items()}
        print(f"{W}x{H} n={n} options={opts}: wall p50 {np.percentile(lat, 50):.1f} us  device us (launches) {per} sum {sum(v[0] for v in per.values()):.1f}")
        c.close()
