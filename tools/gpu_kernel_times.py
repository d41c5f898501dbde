#!/usr/bin/env python3
"""tools/gpu_kernel_times.py [c2|c4] [options] — device ms per step and kernel (HIP events on the context's stream, one batch in flight) and the
wall clock per step with two batches in flight, for a context created with `options` (ht_config.options).  A/B of schedules on one box:
    python tools/gpu_kernel_times.py c2 pyr_frame=0
    python tools/gpu_kernel_times.py c2 "" 3        (third argument: batches in flight, default 2)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from headtrackr_amd import synth  # noqa: E402
from headtrackr_amd.api import Context  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
opts = (sys.argv[2] if len(sys.argv) > 2 else None) or None
DEPTH = int(sys.argv[3]) if len(sys.argv) > 3 else 2
W, H, n = {"c2": (320, 240, 256), "c4": (1280, 720, 128)}[wl]
n = int(os.environ.get("HT_KT_FRAMES", n))  # another batch size at the same frame size
base = synth.mixed_batch(n, W, H, seed0=1234)
dev = torch.from_numpy(base).cuda()
ctxs = []
for _ in range(DEPTH):
    c = Context(options=opts)
    c.set_geometry(W, H, n)
    c.bind_device(dev.data_ptr(), n)
    ctxs.append(c)
c = ctxs[0]
for _ in range(20):
    c.detect_enqueue(0)
    c.detect_collect_best(1)
c.profile(True)
c.kernel_times(reset=True)
K = 20
for _ in range(K):
    c.detect_enqueue(0)
    c.detect_collect_best(1)
kt = c.kernel_times(reset=True)
c.profile(False)
per = {k: round(v["ms"] / K, 5) for k, v in kt.items()}
best = np.zeros(n, dtype=c.detect_collect_best.__globals__["RECT_DTYPE"])


def block(k):
    for cx in ctxs:
        cx.detect_enqueue(0)
    for i in range(k):
        ctxs[i % DEPTH].detect_collect_best_requeue(1, best) if i + DEPTH < k else ctxs[i % DEPTH].detect_collect_best(1, best)


block(100)
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    block(400)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 400 * 1e3)
print(f"{wl} options={opts}: device ms/step {per} sum {sum(per.values()):.4f}; wall ms/step ({DEPTH} in flight) median {np.median(ts):.4f} min {min(ts):.4f} -> {n / np.median(ts):.0f} k frames/s")
