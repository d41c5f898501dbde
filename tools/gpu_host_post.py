"""Host time of one batch's post-processing (C2 shape): the GPU has finished before the collect call, so the call's duration
is D2H + sort + rects + grouping + best face.  Usage: python tools/gpu_host_post.py"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from headtrackr_amd import native, synth
from headtrackr_amd.api import Context
W, H, nf = 320, 240, 256
dev = torch.from_numpy(synth.mixed_batch(nf, W, H, seed0=1234)).cuda()
cx = Context(device=0); cx.set_geometry(W, H, nf); cx.bind_device(dev.data_ptr(), nf, W * H * 4)
buf = np.zeros(nf, dtype=native.RECT_DTYPE)
for mode in ("collect_best", "collect_raw"):
    ts = []
    for r in range(200):
        t0 = time.perf_counter(); cx.detect_enqueue(0); t1 = time.perf_counter()
        while time.perf_counter() - t1 < 0.0012: pass  # the GPU finishes (~0.35 ms); spinning keeps the core hot
        t2 = time.perf_counter()
        if mode == "collect_best": best, nh = cx.detect_collect_best(1, buf)
        else: hits, counts = cx.detect_collect(cap=1 << 17); nh = len(hits)
        t3 = time.perf_counter()
        ts.append((t1 - t0, t3 - t2))
    ts = np.array(ts[5:]) * 1e3
    print(f"{mode}: enqueue {np.median(ts[:,0]):.4f} ms, collect {np.median(ts[:,1]):.4f} ms (min {ts[:,1].min():.4f}), raw hits {nh}")
hits, counts = (cx.detect_enqueue(0), cx.detect_collect(cap=1 << 17))[1]
print("hits per frame: max", int(counts.max()), "mean", float(counts.mean()), "frames with hits", int((counts > 0).sum()), "sum n^2", int((counts.astype(np.int64) ** 2).sum()))
