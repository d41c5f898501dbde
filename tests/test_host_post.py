"""Host post-processing of the C-ABI that needs no device: ht_group_rects (ccv.js:36-105 union-find, 250-330 grouping) against the
oracle on seeded rectangle sets shaped like a cascade's raw hits — clusters over a few adjacent pyramid scales, strays, exact
duplicates, empty and single inputs.  Bit-exact in every field."""
import ctypes as C

import numpy as np
import pytest

from headtrackr_amd import native
from oracle import ht_oracle as ho


def _lib_group(rects, min_neighbors):
    L = native.lib()
    rects = np.ascontiguousarray(rects, dtype=native.RECT_DTYPE)
    out = np.zeros(max(1, len(rects)), dtype=native.RECT_DTYPE)
    n = C.c_uint32(0)
    st = L.ht_group_rects(rects.ctypes.data, len(rects), min_neighbors, out.ctypes.data, C.byref(n))
    assert st == 0
    return out[: n.value]


def _raw_like(rng, n_clusters, per_cluster, strays, w=320, h=240):
    """rectangles as ht_hits_to_rects emits them: (4 x + 2 qx) * s, (4 y + 2 qy) * s, 24 s, 24 s with s = 2^(k/6)"""
    scale = 2.0 ** (1.0 / 6.0)
    rows = []
    for c in range(n_clusters):
        k0 = int(rng.integers(0, 12)); cx = rng.uniform(0, w - 60); cy = rng.uniform(0, h - 60)
        for _ in range(int(rng.integers(1, per_cluster + 1))):
            s = scale ** (k0 + int(rng.integers(0, 3)))
            x = (4 * int((cx + rng.normal(0, 3)) / (4 * s)) + 2 * int(rng.integers(0, 2))) * s
            y = (4 * int((cy + rng.normal(0, 3)) / (4 * s)) + 2 * int(rng.integers(0, 2))) * s
            rows.append((x, y, 24 * s, 24 * s, float(rng.normal(3, 2)), 1, 0))
    for _ in range(strays):
        s = scale ** int(rng.integers(0, 18))
        rows.append((float(rng.uniform(0, w)), float(rng.uniform(0, h)), 24 * s, 24 * s, float(rng.normal(0, 1)), 1, 0))
    a = np.array(rows, dtype=native.RECT_DTYPE) if rows else np.zeros(0, dtype=native.RECT_DTYPE)
    return a[rng.permutation(len(a))] if len(a) else a


@pytest.mark.parametrize("min_neighbors", [1, 2, 3, 5])
def test_group_rects_matches_oracle(min_neighbors):
    rng = np.random.default_rng(20260921 + min_neighbors)
    cases = [np.zeros(0, dtype=native.RECT_DTYPE)]
    for t in range(60):
        cases.append(_raw_like(rng, int(rng.integers(0, 6)), int(rng.integers(1, 30)), int(rng.integers(0, 8))))
    one = _raw_like(rng, 1, 1, 0)
    cases += [one[:1], np.concatenate([one[:1]] * 7), _raw_like(rng, 12, 40, 30, 1920, 1080)]
    for i, rects in enumerate(cases):
        want = ho.group(rects.astype(ho.RECT_DTYPE), min_neighbors)
        got = _lib_group(rects, min_neighbors)
        assert len(got) == len(want), (i, len(rects))
        for f in ("x", "y", "width", "height", "confidence", "neighbors"):
            assert np.array_equal(np.asarray(got[f]), np.asarray(want[f])), (i, f)


def test_group_rects_nan_and_degenerate_widths():
    """NaN coordinates never compare as similar; zero and huge widths go through the same floor() terms as ccv.js:252-261"""
    r = np.zeros(6, dtype=native.RECT_DTYPE)
    r["x"] = [10, 10, np.nan, 10, 1e9, 11]
    r["y"] = [10, 11, 10, 10, 1e9, 10]
    r["width"] = [24, 24, 24, 0, 1e12, 30]
    r["height"] = r["width"]
    r["confidence"] = [1, 2, 3, 4, 5, 6]
    r["neighbors"] = 1
    for mn in (1, 2):
        want = ho.group(r.astype(ho.RECT_DTYPE), mn)
        got = _lib_group(r, mn)
        assert len(got) == len(want)
        for f in ("x", "y", "width", "height", "confidence", "neighbors"):
            assert np.array_equal(np.asarray(got[f]), np.asarray(want[f]), equal_nan=True), (mn, f)
