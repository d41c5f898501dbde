#!/usr/bin/env python3
"""tools/gpu_rs_phases.py [c2|c4] — where a k_resample workgroup spends its life, from a -DHT_RS_PHASES build (python tools/build_alt.py
rsph HT_RS_PHASES=1; run with HEADTRACKR_HIP_LIB=alt/rsph.so).  Phases: 1 record + taps + first loads issued + barrier (once per workgroup);
per frame of the group: 2 loop top, 3 next frame's loads issued, 4 pixels, 5 stores issued + barrier (every wave done with the tile),
6 wait for the prefetched loads + registers -> LDS + barrier."""
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
lib.ht_debug_rs_phases.argtypes = [C.c_void_p, C.c_int]
raw = np.zeros(16, dtype=np.uint64)
c.detect_raw(frames, cap=1 << 18)
lib.ht_debug_rs_phases(raw.ctypes.data, 1)
for rep in range(1 if os.environ.get("HT_RS_LAUNCHES") else 3):  # one batch when the per-launch spans are asked for: they are first entry -> last exit per key
    c.detect_raw(frames, cap=1 << 18)
lib.ht_debug_rs_phases(raw.ctypes.data, 1)
cyc, cnt = raw[:8].astype(np.float64), raw[8:].astype(np.float64)
tot = cyc.sum()
names = {1: "setup + first loads + barrier", 2: "stamp 1 -> first loop top", 3: "prefetch issue", 4: "pixels", 5: "stores + barrier", 6: "load wait + regs -> LDS + barrier"}
print(f"{wl}: {int(cnt[3])} frame iterations sampled, mean iteration {sum(cyc[3:7]) / max(cnt[3], 1):.0f} cycles")
for p in range(1, 7):
    if cnt[p] > 0:
        print(f"  {names[p]:30s} samples {int(cnt[p]):8d}  mean {cyc[p] / cnt[p]:8.0f} cycles")
