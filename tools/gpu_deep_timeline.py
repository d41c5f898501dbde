#!/usr/bin/env python3
"""tools/gpu_deep_timeline.py [c2|c4] — where k_scan_deep_lds spends its time, from a -DHT_DEEP_TIMELINE build (python tools/build_alt.py
dltl HT_DEEP_TIMELINE; run with HEADTRACKR_HIP_LIB=alt/dltl.so).  Per wavefront, first window only: entry -> table copied -> patch loaded -> window
done, by the last stage the window ran."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from headtrackr_amd import synth  # noqa: E402
from headtrackr_amd.api import Context  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
W, H, n, uniq = (320, 240, 256, 256) if wl == "c2" else (1280, 720, 128, 12)
base = synth.mixed_batch(uniq, W, H, seed0=1234)
frames = base[np.arange(n) % uniq]
c = Context()
lib = c._lib
lib.ht_debug_deep_timeline.argtypes = [C.c_void_p, C.c_int]
raw = np.zeros((8192, 8), dtype=np.uint64)
for rep in range(3):
    c.detect_raw(frames, cap=1 << 18)
    lib.ht_debug_deep_timeline(raw.ctypes.data, 8192)
t = raw.astype(np.int64)
ok = (t[:, 0] > 0) & (t[:, 3] >= t[:, 2]) & (t[:, 2] >= t[:, 1]) & (t[:, 1] >= t[:, 0])
t = t[ok]
t0 = t[:, 0].min()
print(f"{wl}: {len(t)} wavefronts with a window; kernel span (first entry -> last window done) {t[:, 3].max() - t0} cycles")
print(f"  entry spread {t[:, 0].max() - t0}; table copy mean {np.mean(t[:, 1] - t[:, 0]):.0f} (max {np.max(t[:, 1] - t[:, 0])}); patch load mean {np.mean(t[:, 2] - t[:, 1]):.0f}")
for j in sorted(set(t[:, 4].tolist())):
    m = t[:, 4] == j
    d = t[m, 3] - t[m, 2]
    ex = t[m, 5] > 0
    extra = f"; exact sum {np.mean(t[m, 3][ex] - t[m, 5][ex]):.0f}" if ex.any() else ""
    print(f"  last stage {int(j):2d}: {int(m.sum()):5d} windows, stages take mean {d.mean():8.0f} cycles (max {d.max()}), done at mean {np.mean(t[m, 3] - t0):8.0f} after the first entry{extra}")
