#!/usr/bin/env python3
"""tools/gpu_cs_step.py FEEDS [options] — the C5 track step on FEEDS x 1080p feeds (one context): device ms per kernel (HIP events, strictly
in turn) and the wall clock per step of the pipelined loop (two enqueue-only track steps outstanding), for a context created with `options`.
A/B of camshift schedules / builds (HEADTRACKR_HIP_LIB) on one box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from headtrackr_amd import synth  # noqa: E402
from headtrackr_amd.api import Context  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
opts = (sys.argv[2] if len(sys.argv) > 2 else None) or None
W, H, NU = 1920, 1080, 30
uniq = synth.stream_feed_frames(NU, W, H, 0)
host = np.empty((NU, K, H, W, 4), dtype=np.uint8)
for k in range(NU):
    for f in range(K):
        host[k, f] = uniq[synth.stream_frame_index(k, f, NU)]
dev = torch.from_numpy(host).cuda()
sb = K * W * H * 4
c = Context(options=opts)
c.set_geometry(W, H, K)
c.camshift_reserve(K)
c.bind_device(dev.data_ptr(), K)
c.camshift_init([(700 + 3 * 7 * f % 200, 300, 360, 360) for f in range(K)])
for i in range(40):
    c.bind_device(dev.data_ptr() + (i % NU) * sb, K)
    c.camshift_track(K)
c.profile(True)
c.kernel_times(reset=True)
N = 60
for i in range(N):
    c.bind_device(dev.data_ptr() + (i % NU) * sb, K)
    c.camshift_track(K)
kt = c.kernel_times(reset=True)
c.profile(False)
per = {k: round(v["ms"] / N * 1e3, 2) for k, v in kt.items()}


def block(n):
    pend = 0
    for i in range(n):
        c.bind_device(dev.data_ptr() + (i % NU) * sb, K)
        c.camshift_track(K, fetch=False)
        pend += 1
        if pend > 1:
            c.camshift_track_collect(K)
            pend -= 1
    while pend:
        c.camshift_track_collect(K)
        pend -= 1


block(200)
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    block(600)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 600 * 1e6)
print(f"{K} feeds options={opts} lib={os.path.basename(os.environ.get('HEADTRACKR_HIP_LIB', 'product'))}: device us/step {per} sum {sum(per.values()):.1f}; "
      f"pipelined wall us/step median {np.median(ts):.1f} min {min(ts):.1f} -> {K / np.median(ts) * 1e3:.1f} k track frames/s")
