#!/usr/bin/env python3
"""tools/gpu_soak.py [seconds] [seed] — randomised parity soak on the GPU, beyond the fixed seeds of tests/.

Draws geometries, frame families and tracker set-ups from a seeded generator until the time budget is spent and compares,
through the C ABI, with the oracle (oracle/ht_oracle.c, pinned to the reference JS by tests/test_oracle_golden.py):
  detect    raw hits of every frame: (scale, q, x, y) and the binary64 confidence BITS; every third case also every pyramid plane;
  camshift  every track() call within BASELINE.json's tolerance (+-1 px, sizes equal, +-0.5 degrees); counted separately: calls that
            are exact in search window, x, y, width, height, and the largest angle difference (the moment sums are added in a
            different order than the reference's pixel loop, so the angle agrees to ~1e-9 rad, not to the bit).
Prints one summary line per family and a final verdict; exit 1 on any mismatch (the failing case's parameters are printed, so
it can be replayed with the same seed).  The oracle is the checker here, never the product path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from headtrackr_amd import synth  # noqa: E402
from headtrackr_amd.api import Context  # noqa: E402
from headtrackr_amd.cascade import load_cascade  # noqa: E402
from oracle import ht_oracle as ho  # noqa: E402

BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) & 0x7FFFFFFF
rng = np.random.default_rng(SEED)
cascade = load_cascade()
fails = []
STATS = {"cs_exact": 0, "angle_max": 0.0, "cs_not_exact": []}


def rand_frame(w, h):
    kind = rng.integers(0, 4)
    seed = int(rng.integers(1, 1 << 30))
    if kind == 0:
        return "N", synth.noise_frame(w, h, seed)
    if kind == 1:
        return "S", synth.smooth_frame(w, h, seed)
    faces = []
    for _ in range(int(rng.integers(1, 4))):
        s = int(rng.integers(24, max(25, min(w, h))))
        if s > min(w, h):
            continue
        faces.append((int(rng.integers(0, w - s + 1)), int(rng.integers(0, h - s + 1)), s))
    fr = synth.face_frame(w, h, faces, gray=int(rng.integers(60, 180)))
    if kind == 3 and w * h > 0:  # faces over texture: deep-stage survivors next to real detections
        tex = synth.smooth_frame(w, h, seed)
        m = (fr[..., 0] == fr[0, 0, 0])[..., None]
        fr = np.where(m, tex, fr).astype(np.uint8)
        return "FS", fr
    return "F", fr


def detect_case(ctx, planes):
    big = rng.random() < 0.15
    w = int(rng.integers(24, 1400 if big else 520))
    h = int(rng.integers(24, 900 if big else 400))
    n = int(rng.integers(1, 4 if big else 9))
    kinds, frames = zip(*[rand_frame(w, h) for _ in range(n)])
    frames = np.stack(frames)
    hits, counts = ctx.detect_raw(frames)
    k = nh = 0
    for i in range(n):
        want = ho.detect_raw(frames[i], cascade.blob)
        got = hits[k:k + int(counts[i])]
        ok = len(got) == len(want) and all(np.array_equal(got[f].astype(np.int64), want[f].astype(np.int64)) for f in ("scale", "q", "x", "y")) \
            and np.array_equal(got["sum"].view(np.uint64), want["sum"].view(np.uint64))
        if not ok:
            fails.append(("detect", w, h, n, i, kinds[i], len(got), len(want)))
        k += int(counts[i])
        nh += len(want)
    npl = 0
    if planes:
        levels, arena = ho.pyramid(frames[0])
        for lv, (_lw, _lh, off) in enumerate(levels):
            for slot in range(4):
                if off[slot] < 0:
                    continue
                want = ho.plane(levels, arena, lv, slot)
                got = ctx.pyramid_readback(0, lv, slot)
                npl += 1
                if got.shape != want.shape or not np.array_equal(got, want):
                    fails.append(("plane", w, h, lv, slot))
    return n, nh, npl, (w, h)


def camshift_case():
    w, h = int(rng.integers(48, 700)), int(rng.integers(40, 500))
    n, steps = int(rng.integers(1, 7)), int(rng.integers(3, 9))
    fused = bool(rng.integers(0, 2))
    specs = []
    for s in range(n):
        a, b = int(rng.integers(4, max(5, w // 4))), int(rng.integers(3, max(4, h // 4)))
        cx, cy = int(rng.integers(a, w - a)), int(rng.integers(b, h - b))
        rot = [(1, 0, 1), (4, 3, 5), (3, 4, 5), (12, 5, 13), (0, 1, 1)][int(rng.integers(0, 5))]
        color = [(200, 60, 40), (40, 200, 80), (40, 80, 230), (220, 200, 30)][int(rng.integers(0, 4))]
        specs.append((cx, cy, a, b, rot, color, rng.integers(-4, 5, size=2 * steps)))
    seqs = []
    for s, (cx, cy, a, b, rot, color, walk) in enumerate(specs):
        x, y, fr = cx, cy, []
        for k in range(steps):
            fr.append(synth.blob_frame(w, h, x, y, a, b, rot, color, seed=int(rng.integers(1, 1 << 20))))
            x = int(np.clip(x + walk[2 * k], 0, w - 1))
            y = int(np.clip(y + walk[2 * k + 1], 0, h - 1))
        seqs.append(fr)
    rects = [(max(0, cx - a), max(0, cy - b), 2 * a, 2 * b) for (cx, cy, a, b, *_r) in specs]
    c = Context(options=",".join(x for x in ("cs_fused_min=1" if fused else "cs_fused_min=1000000", os.environ.get("HT_SOAK_OPTIONS", "")) if x))
    calls = 0
    try:
        c.set_geometry(w, h, n)
        c.camshift_reserve(n)
        c.upload(np.stack([seqs[s][0] for s in range(n)]))
        c.camshift_init(rects)
        oracles = []
        for s in range(n):
            o = ho.Camshift(True)
            o.init_tracker(seqs[s][0], rects[s])
            oracles.append(o)
        for k in range(1, steps):
            c.upload(np.stack([seqs[s][k] for s in range(n)]))
            got = c.camshift_track(n, calc_angles=True)
            for s in range(n):
                sw, to = oracles[s].track(seqs[s][k])
                g = got[s]
                exact = all(float(g[f]) == float(to[f]) or (np.isnan(float(g[f])) and np.isnan(float(to[f]))) for f in ("x", "y", "width", "height")) \
                    and [int(g["sw_x"]), int(g["sw_y"]), int(g["sw_width"]), int(g["sw_height"])] == [int(v) for v in sw]
                wa, ga = float(to["angle"]), float(g["angle"])
                d = 0.0 if (np.isnan(wa) and np.isnan(ga)) else abs(ga - wa)
                d = min(d, abs(d - np.pi))
                # a LOST object (0 x 0 on both sides) has no orientation: a, b, c of camshift.js:230-245 are rounding noise of the sums there and
                # atan2 of noise is anything — the reference's own loop returns 0 or pi / 2 for such calls depending on the summation order
                # (tools/cpu_cs_order_check.py).  Counted, not compared.
                if float(to["width"]) == 0.0 and float(to["height"]) == 0.0 and float(g["width"]) == 0.0 and float(g["height"]) == 0.0 and not d <= np.deg2rad(0.5):
                    STATS["lost_angle"] = STATS.get("lost_angle", 0) + 1
                    d = 0.0
                calls += 1
                STATS["cs_exact"] += int(exact)
                if d <= np.deg2rad(0.5):  # (a call out of tolerance is reported on its own below)
                    STATS["angle_max"] = max(STATS["angle_max"], d)
                same = lambda a_, b_: a_ == b_ or (np.isnan(a_) and np.isnan(b_))  # noqa: E731
                tol = all(same(float(g[f]), float(to[f])) or abs(float(g[f]) - float(to[f])) <= 1 for f in ("x", "y")) \
                    and all(same(float(g[f]), float(to[f])) for f in ("width", "height")) \
                    and all(abs(int(g[f]) - int(v)) <= 1 for f, v in zip(("sw_x", "sw_y"), sw[:2])) and [int(g["sw_width"]), int(g["sw_height"])] == [int(v) for v in sw[2:]]
                if not tol or not d <= np.deg2rad(0.5):  # BASELINE.json: +-1 px, +-0.5 degrees
                    dump = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"soak_fail_cs_{len(fails)}.npz")
                    os.makedirs(os.path.dirname(dump), exist_ok=True)  # the whole case, for tools/gpu_soak_replay.py
                    np.savez_compressed(dump, frames=np.stack([np.stack(fr) for fr in seqs]), rects=np.array(rects, dtype=np.int32), fused=int(fused), stream=s, call=k)
                    rec = ("camshift", w, h, n, steps, fused, s, k, d, [float(g[f]) for f in ("x", "y", "width", "height")], [float(to[f]) for f in ("x", "y", "width", "height")])
                    # is it a tie?  The reference's own loop in another summation order (tools/cpu_cs_order_check.py): if this call, or an earlier call of the
                    # stream, changes under it, the difference is rounding noise at a truncation boundary, not a defect of the kernel
                    import importlib.util
                    spec = importlib.util.spec_from_file_location("cpu_cs_order_check", os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpu_cs_order_check.py"))
                    oc = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(oc)
                    base = oc.run(dump, [])
                    flips = {(a[0], a[1]) for _nm, fl in oc.VARIANTS for a, b in zip(base, oc.run(dump, [fl])) if a[2:] != b[2:]}
                    if any(fs == s and fk <= k for fs, fk in flips):
                        STATS.setdefault("cs_ties", []).append(rec)
                    else:
                        fails.append(rec)
                elif not exact:
                    STATS["cs_not_exact"].append((w, h, s, k))
    finally:
        c.close()
    return calls


def main():
    t0 = time.time()
    ctx = Context()
    nd = nf = nh = npl = ncs = ncalls = 0
    sizes = set()
    while time.time() - t0 < BUDGET and len(fails) < int(os.environ.get("HT_SOAK_MAXFAILS", "5")):
        if nd % 4 == 3 or os.environ.get("HT_SOAK_ONLY") == "camshift":
            ncalls += camshift_case()
            ncs += 1
            if os.environ.get("HT_SOAK_ONLY") == "camshift":
                continue
        n, h_, p_, wh = detect_case(ctx, planes=(nd % 3 == 0))
        nd, nf, nh, npl = nd + 1, nf + n, nh + h_, npl + p_
        sizes.add(wh)
    ctx.close()
    print(f"soak seed {SEED}, {time.time() - t0:.0f} s: detect {nd} cases / {len(sizes)} geometries / {nf} frames / {nh} raw hits (positions + confidence bits) "
          f"and {npl} pyramid planes vs oracle; camshift {ncs} cases / {ncalls} track() calls, {STATS['cs_exact']} exact in window, x, y, width, height "
          f"(the rest within +-1 px: {STATS['cs_not_exact'][:6]}), max |angle difference| {STATS['angle_max']:.3e} rad")
    if STATS.get("lost_angle"):
        print(f"{STATS['lost_angle']} call(s) returned a lost (0 x 0) object whose angle differs from the oracle's: the angle of a 0 x 0 object is atan2 of rounding noise, not compared")
    for rec in STATS.get("cs_ties", []):
        print(f"order-sensitive tie (the reference's own loop returns another object in another summation order; dumped to gpurun_out/): {rec}")
    if fails:
        print(f"MISMATCHES ({len(fails)}):")
        for f in fails:
            print("  ", f)
        raise SystemExit(1)
    print("all within tolerance; detect and pyramid bit-exact")


if __name__ == "__main__":
    main()
