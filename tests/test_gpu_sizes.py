"""Geometry edge cases of the detect path against the oracle: widths that are not multiples of 4, tiny frames (no window
fits), frames whose deep levels collapse to zero pixels, 1920x1080, and a batch whose tiles are all partially filled."""
import numpy as np
import pytest

from headtrackr_amd import synth
from headtrackr_amd.api import Context
from oracle import ht_oracle as ho

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context()
    yield c
    c.close()


def check(ctx, frames, cascade):
    hits, counts = ctx.detect_raw(frames)
    k = 0
    for i in range(len(frames)):
        h = ho.detect_raw(frames[i], cascade.blob)
        g = hits[k : k + int(counts[i])]
        assert len(g) == len(h), (i, len(g), len(h))
        for name in ("scale", "q", "x", "y"):
            assert np.array_equal(g[name].astype(np.int64), h[name].astype(np.int64)), (i, name)
        assert np.array_equal(g["sum"].view(np.uint64), h["sum"].view(np.uint64))
        k += len(h)
    return hits


@pytest.mark.parametrize("w,h", [(97, 81), (131, 99), (333, 217), (258, 130), (150, 400), (511, 97), (96, 96), (40, 30), (23, 23), (1, 1)])
def test_odd_sizes(ctx, cascade, w, h):
    s = max(8, min(w, h) * 3 // 4)
    frames = np.stack([synth.face_frame(w, h, [(max(0, (w - s) // 2), max(0, (h - s) // 2), s)]), synth.smooth_frame(w, h, w + h),
                       synth.noise_frame(w, h, 3 * w + h)])
    check(ctx, frames, cascade)


def test_1080p(ctx, cascade):
    w, h = 1920, 1080
    # the C5 geometry: one F, one N and one S frame (raw hits incl. confidence bits; the planes are compared in
    # test_gpu_detect.py::test_pyramid_planes_vs_oracle[1920-1080])
    frames = np.stack([synth.face_frame(w, h, [(300, 200, 400), (1200, 500, 96), (1700, 100, 64)]), synth.noise_frame(w, h, 5), synth.smooth_frame(w, h, 6)])
    hits = check(ctx, frames, cascade)
    assert len(hits) >= 20 and ctx.windows_per_frame == 2344044
    p = ctx.plane(38, 0)
    assert (p.width, p.height) == (23, 13)  # level 2 (1523 x 857) halved six times


def test_many_small_frames_batch(ctx, cascade):
    """512 frames of 96x80: every tile is a partial tile and the XCD-aware frame order wraps around"""
    w, h, n = 96, 80, 512
    base = np.stack([synth.face_frame(w, h, [(10 + (i % 7), 6 + (i % 5), 64)]) if i % 2 else synth.smooth_frame(w, h, i) for i in range(16)])
    frames = base[np.arange(n) % 16]
    hits, counts = ctx.detect_raw(frames)
    ref = [ho.detect_raw(base[i], cascade.blob) for i in range(16)]
    k = 0
    for f in range(n):
        h_ = ref[f % 16]
        g = hits[k : k + int(counts[f])]
        assert len(g) == len(h_) and np.all(g["frame"] == f)
        assert np.array_equal(g["sum"].view(np.uint64), h_["sum"].view(np.uint64))
        assert np.array_equal(g["x"].astype(np.int64), h_["x"].astype(np.int64))
        k += len(h_)
    assert k == len(hits) and k > 0
