"""camshift kernel timing probe (GPU box): per-kernel device time of track() for 256 x 320x240 and 128 x 1280x720 streams"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from headtrackr_amd import synth
from headtrackr_amd.api import Context

for (W, H, n) in ((320, 240, 256), (1280, 720, 128)):
    base = np.stack([synth.blob_frame(W, H, W // 2 + 3 * i, H // 2 - 2 * i, W // 8, H // 10, (4, 3, 5), seed=5 + i) for i in range(4)])
    frames = base[np.arange(n) % 4]
    ctx = Context()
    ctx.set_geometry(W, H, n)
    ctx.upload(frames)
    ctx.camshift_reserve(n)
    ctx.camshift_init([(W // 2 - W // 8, H // 2 - H // 10, W // 4, H // 5)] * n)
    for _ in range(3):
        ctx.camshift_track(n, True)
    ctx.profile(True); ctx.kernel_times(True)
    t0 = time.perf_counter()
    K = 20
    for i in range(K):
        out = ctx.camshift_track(n, True, fetch=(i == K - 1))
    dt = time.perf_counter() - t0
    kt = ctx.kernel_times(True)
    print(f"{W}x{H} x{n}: {dt / K * 1e3:.3f} ms/track-call wall;", {k: round(v['ms'] / K, 4) for k, v in kt.items()}, "sample", out[0])
    ctx.close()
