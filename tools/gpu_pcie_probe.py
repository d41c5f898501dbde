"""PCIe-inclusive rate: frames start in HOST memory (pinned and pageable) and go through ht_upload_frames + detect + collect"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from headtrackr_amd import synth
from headtrackr_amd.api import Context
for (W, H, n, uniq) in ((320, 240, 256, 256), (1280, 720, 128, 12)):
    base = synth.mixed_batch(uniq, W, H)
    frames = base[np.arange(n) % uniq].copy()
    pinned = torch.from_numpy(frames).pin_memory()
    ctx = Context(); ctx.set_geometry(W, H, n)
    for name, ptr in (("pageable", frames.ctypes.data), ("pinned", pinned.data_ptr())):
        for _ in range(3):
            ctx.upload_ptr(ptr, n); ctx.detect_enqueue(0); ctx.detect_collect()
        t0 = time.perf_counter(); K = 10
        for _ in range(K):
            ctx.upload_ptr(ptr, n); ctx.detect_enqueue(0); ctx.detect_collect()
        dt = (time.perf_counter() - t0) / K
        print(f"{W}x{H} x{n} {name}: {dt*1e3:.3f} ms/batch = {n/dt:.0f} frames/s ({n*W*H*4/dt/1e9:.1f} GB/s over PCIe)")
    # double-buffered ingest: the next batch crosses PCIe on the copy stream while the current one is scanned
    ptr = pinned.data_ptr()
    ctx.upload_async_ptr(ptr, n); ctx.swap_frames()
    for _ in range(3):
        ctx.upload_async_ptr(ptr, n); ctx.detect_enqueue(0); ctx.detect_collect(); ctx.swap_frames()
    t0 = time.perf_counter(); K = 20
    for _ in range(K):
        ctx.upload_async_ptr(ptr, n); ctx.detect_enqueue(0); ctx.detect_collect(); ctx.swap_frames()
    dt = (time.perf_counter() - t0) / K
    print(f"{W}x{H} x{n} pinned, double-buffered: {dt*1e3:.3f} ms/batch = {n/dt:.0f} frames/s ({n*W*H*4/dt/1e9:.1f} GB/s over PCIe)")
    ctx.close()
