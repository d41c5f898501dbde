"""The JavaScript host (north_star: "Host code stays JavaScript (Node)"): the same C ABI driven from Node through the
N-API addon."""
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np

from .common import ROOT


def js_host_bench(seconds=2.0):
    """tests/js/bench_host.js on this GPU: detect frames/s at the C2 shape from Node — ccv.detect_objects_batch on host
    frames (PCIe every call) and ccv.DeviceBatch (frames resident in HBM, enqueue / collect-best / re-enqueue over 2
    contexts: bench.py's headline loop, driven from JavaScript) — and the per-call latency of the drop-in
    facetrackr.Tracker.track() at 320x240, next to the unmodified reference JS on the same frames; then the C5 loop
    (tests/js/c5_stream.js).  None when node or the addon is missing."""
    from headtrackr_amd import synth

    node = shutil.which("node")
    script = os.path.join(ROOT, "tests", "js", "bench_host.js")
    addon = os.path.join(ROOT, "headtrackr_amd", "js", "headtrackr_hip.node")
    if not node or not os.path.exists(script) or not os.path.exists(addon):
        return None
    W, H, n, nt = 320, 240, 256, 30
    try:
        with tempfile.TemporaryDirectory() as td:
            c2 = os.path.join(td, "c2.raw")
            synth.mixed_batch(n, W, H, seed0=1234).tofile(c2)
            tr = os.path.join(td, "track.raw")
            np.stack([synth.face_frame(W, H, [(90 + 2 * k, 50 + k, 96)]) for k in range(nt)]).tofile(tr)
            r = subprocess.run([node, script, str(seconds), c2, str(n), tr, str(nt)], capture_output=True, text=True,
                               timeout=seconds * 20 + 240)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            # C5 from the JavaScript host: 8 frame-synchronous 1080p feeds, DeviceBatch.detectStep / trackStep
            try:
                uq = os.path.join(td, "uniq.raw")
                synth.stream_feed_frames(30, 1920, 1080, 0).tofile(uq)
                r5 = subprocess.run([node, os.path.join(ROOT, "tests", "js", "c5_stream.js"), "bench", uq, "30", "8",
                                     str(seconds)], capture_output=True, text=True, timeout=seconds * 20 + 240)
                j["c5"] = json.loads(r5.stdout.strip().splitlines()[-1])
            except Exception as e:
                j["c5"] = {"error": f"{type(e).__name__}: {e}"}
        j["config"] = {"workload": f"JS host (Node + N-API addon): {n} x {W}x{H} detect per batch (the C2 frames), "
                                   f"facetrackr.Tracker.track() on a {W}x{H} canvas with one drifting face, and the "
                                   "C5 loop (8 x 1080p feeds per step) through ccv.DeviceBatch.detectStep / trackStep"}
        return j
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}
