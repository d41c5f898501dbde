#!/usr/bin/env python3
"""tools/gpu_cs_timeline.py — phase timeline of k_cs_track_fused from a -DHT_CS_TIMELINE build (HEADTRACKR_HIP_LIB=alt/cstl.so): shader-clock stamps of workgroups 0..7 for one track() call of the C3 workload.  Stamps: 0 start, 1 histogram done,
2 LUT done, 3 region cached, 4.. after each moment pass, last = loop done."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from headtrackr_amd import synth  # noqa: E402
from headtrackr_amd.api import Context  # noqa: E402
from hipmem import DeviceArray  # noqa: E402

W, H, n, NV = 320, 240, 256, 4
walk = synth.lcg_stream(4242, 2 * NV * n).astype(np.int64) >> 20
vers = np.empty((NV, n, H, W, 4), dtype=np.uint8)
for f in range(n):
    s0 = 48 + (f * 7) % 80
    x, y = 20 + (f * 13) % (W - s0 - 40), 16 + (f * 29) % (H - s0 - 32)
    for v in range(NV):
        vers[v, f] = synth.face_frame(W, H, [(x, y, s0)])
        x += int(walk[2 * (f * NV + v)] % 7) - 3
        y += int(walk[2 * (f * NV + v) + 1] % 7) - 3
c = Context(options="cs_keep_hist=1")
dev = [DeviceArray(vers[v]) for v in range(NV)]
c.set_geometry(W, H, n)
c.bind_device(dev[0].ptr, n)
c.camshift_reserve(n)
c.detect_enqueue(0)
hits, counts = c.detect_collect(cap=1 << 17)
best = c.best_faces(hits, counts, 1)
rects = [(int(best["x"][f]), int(best["y"][f]), int(best["width"][f]), int(best["height"][f])) if best["neighbors"][f] > 0 else (80, 60, 160, 120) for f in range(n)]
c.camshift_init(rects)
for k in range(6):
    c.bind_device(dev[(k + 1) % NV].ptr, n)
    out = c.camshift_track(n)
for s in (0, 1, 2, 3, 100, 200, 255):
    model, cur = c.camshift_debug_hist(s)
    raw = cur[4032:4092].view(np.uint64).astype(np.int64)
    fine = raw[16:24]
    print("   first pass fine: pixel loop", fine[1] - fine[0], "barrier", fine[2] - fine[1], "wave sums", fine[3] - fine[2], "barrier", fine[4] - fine[3], "final sums", fine[5] - fine[4], "scalar logic", fine[7] - fine[6])
    st = raw[:16]
    st = st[st > 0]
    d = np.diff(st)
    print(f"stream {s}: window {int(out[s]['sw_width'])}x{int(out[s]['sw_height'])} total {int(st[-1] - st[0])} cycles; hist {d[0]} lut {d[1]} region {d[2]} passes {list(d[3:])}")
