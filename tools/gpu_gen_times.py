#!/usr/bin/env python3
"""device time per pyramid generation (HT_DEBUG_RS_GENNAMES=1) and per scan kernel, one batch in flight: python tools/gpu_gen_times.py c2|c4"""
import os, sys
os.environ["HT_DEBUG_RS_GENNAMES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from headtrackr_amd import synth
from headtrackr_amd.api import Context
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
W, H, n, uniq = (320, 240, 256, 256) if wl == "c2" else (1280, 720, 128, 12)
base = synth.mixed_batch(uniq, W, H, seed0=1234)
dev = torch.from_numpy(base[np.arange(n) % uniq]).cuda()
c = Context()
c.set_geometry(W, H, n)
c.bind_device(dev.data_ptr(), n)
for _ in range(20):
    c.detect_enqueue(0); c.detect_collect(cap=1 << 17)
c.profile(True); c.kernel_times(reset=True)
K = 20
for _ in range(K):
    c.detect_enqueue(0); c.detect_collect(cap=1 << 17)
kt = c.kernel_times()
tot = sum(v["ms"] for v in kt.values()) / K
print(wl, "device ms per step:", round(tot, 4))
for k, v in kt.items():
    print(f"  {k:16s} {v['ms'] / K:8.4f} ms  ({v['launches'] // K} launch)")
