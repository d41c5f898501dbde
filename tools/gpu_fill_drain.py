"""Where the fixed cost of a short timed block goes (the driver times 20 steps = 5 ms): host timestamps of every collect in a
k-step block of bench.py's C2 loop, for several pipeline depths.  Usage: python tools/gpu_fill_drain.py [k] [depths...]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from headtrackr_amd import native, synth
from headtrackr_amd.api import Context

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
depths = [int(x) for x in sys.argv[2:]] or [2, 3, 4]
import os
STAGGER = float(os.environ.get("STAGGER_MS", "0")) * 1e-3  # host-side delay before every up-front enqueue but the first
W, H, nf = 320, 240, 256
dev = torch.from_numpy(synth.mixed_batch(nf, W, H, seed0=1234)).cuda()
for depth in depths:
    ctxs = []
    for _ in range(depth):
        cx = Context(device=0); cx.set_geometry(W, H, nf); cx.bind_device(dev.data_ptr(), nf, W * H * 4); ctxs.append(cx)
    bufs = {id(c): np.zeros(nf, dtype=native.RECT_DTYPE) for c in ctxs}
    def block(k, stamps=None):
        started = min(depth, k)
        for i in range(started):
            if i and STAGGER:
                t_s = time.perf_counter()
                while time.perf_counter() - t_s < STAGGER: pass
            ctxs[i].detect_enqueue(0)
        if stamps is not None: stamps.append(time.perf_counter())
        for i in range(k):
            more = started < k
            c = ctxs[i % depth]
            if more: c.detect_collect_best_requeue(1, bufs[id(c)], 0)
            else: c.detect_collect_best(1, bufs[id(c)])
            started += more
            if stamps is not None: stamps.append(time.perf_counter())
    t = time.perf_counter()
    while time.perf_counter() - t < 0.3: block(24)
    res = []
    for r in range(15):
        torch.cuda.synchronize(); st = [time.perf_counter()]
        block(k, st); torch.cuda.synchronize(); st.append(time.perf_counter())
        res.append(np.diff(np.array(st)) * 1e3)
    res = np.median(np.array(res), axis=0)
    tot = res.sum()
    big = []
    for kk in (200,):
        torch.cuda.synchronize(); t0 = time.perf_counter(); block(kk); torch.cuda.synchronize(); big.append((time.perf_counter() - t0) * 1e3 / kk)
    print(f"depth {depth}: {k} steps {tot:.3f} ms = {tot / k:.4f} ms/step; steady (200 steps) {big[0]:.4f} ms/step; fixed cost {tot - k * big[0]:.3f} ms")
    print("  enqueue-up-front %.3f | collects: %s | final sync %.3f" % (res[0], " ".join("%.3f" % x for x in res[1:-1]), res[-1]))
    del ctxs
