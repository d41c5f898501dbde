#!/usr/bin/env python3
"""tools/gpu_subbatch.py N_SUB DEPTH [options] — the C2 workload (256 x 320x240) cut into sub-batches of N_SUB frames, DEPTH of them in flight
on their own contexts (hipGraph replay where options allow): frames/s of the detect step (collect_best + requeue), host-paced from Python."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from headtrackr_amd import synth  # noqa: E402
from headtrackr_amd.api import Context  # noqa: E402
from headtrackr_amd.native import RECT_DTYPE  # noqa: E402

n_sub, depth = int(sys.argv[1]), int(sys.argv[2])
opts = (sys.argv[3] if len(sys.argv) > 3 else None) or None
W, H, N = 320, 240, 256
base = synth.mixed_batch(N, W, H, seed0=1234)
dev = torch.from_numpy(base).cuda()
fb = W * H * 4
ctxs = []
for i in range(depth):
    c = Context(options=opts)
    c.set_geometry(W, H, n_sub)
    c.bind_device(dev.data_ptr() + ((i * n_sub) % N) * fb, n_sub)
    ctxs.append(c)
best = np.zeros(n_sub, dtype=RECT_DTYPE)
for c in ctxs:  # plain run, capture, first replay
    for _ in range(3):
        c.detect_enqueue(0)
        c.detect_collect_best(1, best)


def block(k):
    for cx in ctxs:
        cx.detect_enqueue(0)
    for i in range(k):
        ctxs[i % depth].detect_collect_best_requeue(1, best) if i + depth < k else ctxs[i % depth].detect_collect_best(1, best)


K = max(400, 4 * depth) * (256 // n_sub)
K = min(K, 6400)
block(K // 4)
ts = []
for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    block(K)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / K)
g = sum(c.graph_launches for c in ctxs)
print(f"sub-batch {n_sub} x {depth} in flight, options={opts}: {np.median(ts) * 1e6:.1f} us per sub-batch -> {n_sub / np.median(ts) / 1e3:.0f} k frames/s (graph launches {g})")
