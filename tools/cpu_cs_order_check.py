#!/usr/bin/env python3
"""tools/cpu_cs_order_check.py CASE.npz — does a camshift case's result depend on the ORDER of the binary64 moment sums?  (CPU only.)

Replays a case dumped by tools/gpu_soak.py through the oracle (the reference's column-major pixel loop, camshift.js:79-120) and through the same
oracle compiled with -DHO_MOMENTS_ROW_MAJOR (rows outer, columns inner), -DHO_MOMENTS_REVERSED (the reference's loops walked backwards) and
-DHO_MOMENTS_TWO_ACCUMULATORS (even and odd rows summed apart): three of the many orders a parallel reduction may resemble.  A call whose track
object differs between them is order-sensitive: its sums sit on a truncation boundary, and no implementation that does not add the pixels in
exactly the reference's sequence can promise the reference's result there."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

VARIANTS = (("row-major", "-DHO_MOMENTS_ROW_MAJOR"), ("reversed", "-DHO_MOMENTS_REVERSED"), ("two accumulators", "-DHO_MOMENTS_TWO_ACCUMULATORS"))


def run(case, variant_flags):
    """[(stream, call, [x, y, width, height], search window, angle)] of every track() call of the case, oracle built with `variant_flags`"""
    from oracle import ht_oracle as ho

    real = (ho._SO, ho._lib)
    so = os.path.join(tempfile.mkdtemp(), "libht_oracle_variant.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-std=c11", *variant_flags, "-shared", "-o", so,
                           os.path.join(ROOT, "oracle", "ht_oracle.c"), "-lm"])
    ho._SO, ho._lib = so, None  # the binding loads whatever _SO names (it is newer than the source: no rebuild)
    try:
        z = np.load(case)
        frames, rects = z["frames"], [tuple(int(v) for v in r) for r in z["rects"]]
        out = []
        for s in range(frames.shape[0]):
            o = ho.Camshift(True)
            o.init_tracker(frames[s][0], rects[s])
            for k in range(1, frames.shape[1]):
                sw, to = o.track(frames[s][k])
                out.append((s, k, [float(to[f]) for f in ("x", "y", "width", "height")], [int(v) for v in sw], float(to["angle"])))
        return out
    finally:
        ho._SO, ho._lib = real  # back to the real oracle


if __name__ == "__main__":
    a = run(sys.argv[1], [])
    sensitive = set()
    for name, flag in VARIANTS:
        b = run(sys.argv[1], [flag])
        for (s, k, ta, swa, aa), (_, _, tb, swb, ab) in zip(a, b):
            da = 0.0 if (aa != aa and ab != ab) else abs(aa - ab)
            da = min(da, abs(da - 3.141592653589793))
            if ta != tb or swa != swb or not da <= 0.008726646259971648:  # half a degree
                sensitive.add((s, k))
                print(f"stream {s} call {k}: column-major (reference) {ta} {swa} angle {aa!r}   {name} {tb} {swb} angle {ab!r}")
    print(f"{len(sensitive)} of {len(a)} calls depend on the summation order: {sorted(sensitive)}")
