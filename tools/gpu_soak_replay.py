#!/usr/bin/env python3
"""tools/gpu_soak_replay.py CASE.npz [options ...] — replays a camshift case that tools/gpu_soak.py dumped (gpurun_out/soak_fail_cs_N.npz) under each
options string and prints, per stream and call, the library's track object next to the oracle's."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from headtrackr_amd.api import Context  # noqa: E402
from oracle import ht_oracle as ho  # noqa: E402

z = np.load(sys.argv[1])
frames, rects = z["frames"], [tuple(int(v) for v in r) for r in z["rects"]]
n, steps, h, w = frames.shape[0], frames.shape[1], frames.shape[2], frames.shape[3]
print(f"{w}x{h}, {n} streams, {steps} frames, rects {rects}, failing (stream, call) = ({int(z['stream'])}, {int(z['call'])})")
want = []
for s in range(n):
    o = ho.Camshift(True)
    o.init_tracker(frames[s][0], rects[s])
    want.append([o.track(frames[s][k]) for k in range(1, steps)])
for opts in (sys.argv[2:] or [""]):
    c = Context(options=opts or None)
    c.set_geometry(w, h, n)
    c.camshift_reserve(n)
    c.upload(np.ascontiguousarray(frames[:, 0]))
    c.camshift_init(rects)
    bad = 0
    for k in range(1, steps):
        c.upload(np.ascontiguousarray(frames[:, k]))
        got = c.camshift_track(n, calc_angles=True)
        for s in range(n):
            sw, to = want[s][k - 1]
            g = [float(got[s][f]) for f in ("x", "y", "width", "height", "angle")] + [int(got[s][f]) for f in ("sw_x", "sw_y", "sw_width", "sw_height")]
            t = [float(to[f]) for f in ("x", "y", "width", "height", "angle")] + [int(v) for v in sw]
            same = all(a == b or (a != a and b != b) for a, b in zip(g[:4] + g[5:], t[:4] + t[5:]))
            bad += not same
            if not same or (s == int(z["stream"])):
                print(f"  [{opts}] stream {s} call {k}: {'==' if same else '!='} got {g} want {t}")
    print(f"[{opts}] {bad} calls differ from the oracle in x / y / width / height / search window")
    c.close()
