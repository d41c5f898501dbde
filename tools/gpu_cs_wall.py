"""tools/gpu_cs_wall.py [options] — wall clock of a single-stream camshift track() call at 320x240 / 640x480 / 1920x1080, synchronous and as
enqueue-only + collect (no profiler attached)."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from headtrackr_amd import synth
from headtrackr_amd.api import Context
for (W, H) in ((320, 240), (640, 480), (1920, 1080)):
    for fetch in (True, False):
        n = 1
        fr = np.stack([synth.face_frame(W, H, [(W // 3, H // 4, min(W, H) // 3)])])
        dev = torch.from_numpy(fr).cuda()
        c = Context(options=(sys.argv[1] if len(sys.argv) > 1 else None) or None)
        c.set_geometry(W, H, n)
        c.bind_device(dev.data_ptr(), n)
        c.camshift_reserve(n)
        rects = np.zeros(n, dtype=[("x", "<i4"), ("y", "<i4"), ("width", "<i4"), ("height", "<i4")])
        rects["x"], rects["y"], rects["width"], rects["height"] = W // 3, H // 4, min(W, H) // 3, min(W, H) // 3
        c.camshift_init(rects)
        lat = []
        for i in range(300):
            t0 = time.perf_counter()
            if fetch:
                c.camshift_track(n, calc_angles=True)
            else:
                c.camshift_track(n, calc_angles=True, fetch=False)
                c.camshift_track_collect(n)
            lat.append((time.perf_counter() - t0) * 1e6)
        print(f"{W}x{H} 1 stream, {'synchronous' if fetch else 'enqueue + collect'}: p50 {np.percentile(lat[50:], 50):.1f} us  min {min(lat[50:]):.1f}")
        c.close()
