#!/usr/bin/env python3
"""tools/gpu_rsb_phases.py [c2|c4] — where a k_resample_bands WAVEFRONT spends a frame iteration, from a -DHT_RS_PHASES build (python
tools/build_alt.py rsph HT_RS_PHASES=1; run with HEADTRACKR_HIP_LIB=alt/rsph.so): pixels (LDS taps + arithmetic), issue of the next frame's four
LDS-DMA loads, the wait for them (s_waitcnt vmcnt(0): the only wait of the loop — there is no workgroup barrier), the stores' issue.  The
counterpart of tools/gpu_rs_phases.py, which stamps k_resample's workgroup phases (option rs_bands=0)."""
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
lib.ht_debug_rsb_phases.argtypes = [C.c_void_p, C.c_int]
raw = np.zeros(8, dtype=np.uint64)
c.detect_raw(frames, cap=1 << 18)
lib.ht_debug_rsb_phases(raw.ctypes.data, 1)
for rep in range(3):
    c.detect_raw(frames, cap=1 << 18)
lib.ht_debug_rsb_phases(raw.ctypes.data, 1)
r = raw.astype(np.float64)
it = max(r[0], 1.0)
print(f"{wl}: k_resample_bands, {int(r[0])} wavefront frame iterations sampled (every iteration but a group's last), mean {sum(r[1:5]) / it:.0f} cycles")
for name, v in (("pixels (LDS taps + arithmetic)", r[1]), ("issue of the next band's LDS-DMA loads", r[2]), ("wait for the band (vmcnt)", r[3]), ("stores issued", r[4])):
    print(f"  {name:40s} mean {v / it:8.0f} cycles  {v / max(sum(r[1:5]), 1) * 100:5.1f} %")
print(f"  record -> first loop top (once per workgroup, {int(r[5])} workgroups): mean {r[6] / max(r[5], 1):.0f} cycles")
