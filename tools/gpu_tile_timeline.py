#!/usr/bin/env python3
"""tools/gpu_tile_timeline.py [c2|c4] — where a k_scan_tiles workgroup spends its life, from a -DHT_TILE_TIMELINE build
(python tools/build_alt.py tltl HT_TILE_TIMELINE; run with HEADTRACKR_HIP_LIB=alt/tltl.so, as tools/gpu_final.sh does).  Thread 0 of every workgroup adds the
shader-clock time of each phase to the statistics rows; this prints mean cycles per phase over the workgroups that went
through it, and each phase's share of all workgroup-cycles."""
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
names = ["staging", "stage 0", "stage 1", "stage 2", "stage 3", "stage 4", "stage 5", "stage 6", "stage 7", "stages >= 8", "hand-off"]
for rep in range(3):
    c.detect_raw(frames, flags=16, cap=1 << 18)
    raw = np.zeros(64, dtype=np.uint64)
    c._check(c._lib.ht_stage_counts(c._h, raw.ctypes.data, 64))
cyc, cnt = raw[32:48].astype(np.float64), raw[48:64].astype(np.float64)
tot = cyc.sum()
print(f"{wl}: {int(cnt[0])} workgroups, mean life {tot / max(cnt[0], 1):.0f} cycles")
for p, name in enumerate(names):
    if cnt[p] > 0:
        print(f"  {name:12s} {int(cnt[p]):7d} workgroups  mean {cyc[p] / cnt[p]:8.0f} cycles  {100 * cyc[p] / tot:5.1f} % of all workgroup-cycles")
print("  windows entering each stage:", [int(v) for v in raw[:17]])
