"""Cascades other than the built-in one (the `cascade` argument of ccv.detect_objects is data, ccv.js:109): random small
BBF cascades through the generic table-driven kernels, checked against the oracle.  Includes cascades whose alphas make
EXACT ties with the stage threshold likely (integer-decision fallback to the sequential binary64 sum), non-symmetric
alphas, alphas that are not 8-digit decimals (integer decisions must switch themselves off), features with 1..8 points."""
import struct

import numpy as np
import pytest

from headtrackr_amd import synth
from headtrackr_amd.api import HT_SCAN_NO_SPLIT, HT_SCAN_SIMPLE, Context
from headtrackr_amd.cascade import parse_cascade
from oracle import ht_oracle as ho

pytestmark = pytest.mark.gpu

THR_PASS = {"ties": 0.18, "asym": 0.2, "binary": 0.2}


def make_cascade(seed, nstages, feats_per_stage, alpha_mode, thr_pass=0.5):
    rng = np.random.RandomState(seed)
    stages, feats = [], []
    first = 0
    for j in range(nstages):
        n = feats_per_stage[j]
        alphas = []
        for k in range(n):
            size = int(rng.choice([1, 1, 2, 2, 3, 5, 8]))
            px = np.zeros(8, np.int8); py = np.zeros(8, np.int8); pz = -np.ones(8, np.int8)
            nx = np.zeros(8, np.int8); ny = np.zeros(8, np.int8); nz = -np.ones(8, np.int8)
            for q in range(size):
                for (xs, ys, zs) in ((px, py, pz), (nx, ny, nz)):
                    if q == 0 or rng.rand() < 0.7:  # slot 0 is always valid (ccv.js:191-192)
                        z = int(rng.randint(0, 3)); lim = 24 >> z
                        xs[q], ys[q], zs[q] = rng.randint(0, lim), rng.randint(0, lim), z
            if alpha_mode == "ties":          # few distinct values -> subset sums hit the threshold exactly
                a1 = float(rng.choice([0.25, 0.5, 0.75, 1.0])); a0 = -a1
            elif alpha_mode == "asym":        # alpha[2k] != -alpha[2k+1], 8-digit decimals
                a0 = round(float(rng.uniform(-2, 0.5)), 6); a1 = round(float(rng.uniform(-0.5, 2)), 6)
            else:                              # "binary": not decimal-representable -> no integer decisions
                a1 = float(rng.uniform(0.1, 2.0)); a0 = -a1 * float(rng.uniform(0.5, 1.0))
            alphas.append((a0, a1))
            feats.append(struct.pack("<B7x8b8b8b8b8b8bdd", size, *px, *py, *pz, *nx, *ny, *nz, a0, a1))
        lo = sum(min(a) for a in alphas); hi = sum(max(a) for a in alphas)
        if alpha_mode == "ties":
            thr = float(np.round((lo + thr_pass * (hi - lo)) * 4) / 4)  # a reachable multiple of 0.25
        else:
            thr = round(lo + thr_pass * (hi - lo), 6)
        stages.append(struct.pack("<IId", n, first, thr))
        first += n
    blob = struct.pack("<4sIIIIIII", b"HTCB", 1, nstages, 24, 24, first, 8, 0) + b"".join(stages) + b"".join(feats)
    return parse_cascade(blob)


@pytest.mark.parametrize("mode,seed", [("ties", 1), ("ties", 2), ("asym", 3), ("binary", 4), ("ties", 5)])
def test_random_cascade_vs_oracle(mode, seed):
    nst = 6
    casc = make_cascade(seed, nst, [3, 4, 6, 9, 70, 130], mode, thr_pass=THR_PASS[mode])
    frames = np.stack([synth.noise_frame(160, 120, 10 + seed), synth.smooth_frame(160, 120, 20 + seed),
                       synth.face_frame(160, 120, [(30, 20, 70)])])
    want = []
    for i, f in enumerate(frames):
        h = ho.detect_raw(f, casc.blob, cap=1 << 20)
        want.append((i, h))
    nref = sum(len(h) for _, h in want)
    assert nref > 0, "test cascade passes nothing: pick other parameters"
    for flags in (0, HT_SCAN_NO_SPLIT, HT_SCAN_SIMPLE):
        ctx = Context(cascade=casc, hit_capacity=1 << 20)
        hits, counts = ctx.detect_raw(frames, flags=flags, cap=1 << 20)
        assert len(hits) == nref, (flags, len(hits), nref)
        k = 0
        for i, h in want:
            g = hits[k : k + len(h)]
            assert np.all(g["frame"] == i)
            for name in ("scale", "q", "x", "y"):
                assert np.array_equal(g[name].astype(np.int64), h[name].astype(np.int64)), (flags, name)
            assert np.array_equal(g["sum"].view(np.uint64), h["sum"].view(np.uint64)), (flags, "confidence bits")
            k += len(h)
        ctx.close()


def test_rejects_bad_cascades():
    from headtrackr_amd.api import HtError

    good = make_cascade(1, 2, [2, 2], "asym")
    with pytest.raises(HtError):
        Context(cascade=_broken(good))


def _broken(c):
    b = bytearray(c.blob)
    off = 32 + 16 * c.count  # first feature: make its first positive slot invalid (pz[0] = -1)
    b[off + 8 + 16] = 0xFF
    return parse_cascade(bytes(b))
