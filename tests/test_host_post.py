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


# ---------------------------------------------------------------------------------------------------------------------
# The per-batch host pass (ht_hostpost.h: counting sort by frame, per-frame ordering, seq rects, grouping, best face, worker pool)
# compiled with AddressSanitizer + UBSan into a host-only harness and run on seeded raw-hit sets: sanitizer-clean, equal to the oracle,
# and byte-identical with 0, 3 and 7 workers.


def _build_harness(tmp_path_factory, name, sanitize):
    import os
    import subprocess

    from conftest import ROOT

    exe = str(tmp_path_factory.mktemp(name) / "hostpost_harness")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitize}", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-pthread",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "headtrackr_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "hostpost_harness.cc"), "-o", exe])
    return exe


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    return _build_harness(tmp_path_factory, "hostpost", "address,undefined")


@pytest.fixture(scope="module")
def harness_tsan(tmp_path_factory):
    return _build_harness(tmp_path_factory, "hostpost_tsan", "thread")


def _raw_hits(rng, nfr, faces_per_frame, strays):
    """raw hits as the kernels append them: index form, frames interleaved in arrival order"""
    rows = []
    for f in range(nfr):
        for _ in range(int(rng.integers(0, faces_per_frame + 1))):
            s0, x0, y0 = int(rng.integers(0, 14)), int(rng.integers(0, 40)), int(rng.integers(0, 30))
            for _ in range(int(rng.integers(1, 14))):
                rows.append((f, x0 + int(rng.integers(0, 2)), y0 + int(rng.integers(0, 2)), min(26, s0 + int(rng.integers(0, 3))), int(rng.integers(0, 4)), 0, 0, float(rng.normal(3, 2))))
        for _ in range(int(rng.integers(0, strays + 1))):
            rows.append((f, int(rng.integers(0, 60)), int(rng.integers(0, 45)), int(rng.integers(0, 27)), int(rng.integers(0, 4)), 0, 0, float(rng.normal(0, 1))))
    seen, uniq = set(), []
    for r in rows:  # a window is scanned once: (frame, scale, q, y, x) is unique in a batch's raw hits
        key = (r[0], r[3], r[4], r[2], r[1])
        if key not in seen:
            seen.add(key)
            uniq.append(r)
    a = np.array(uniq, dtype=native.HIT_DTYPE) if uniq else np.zeros(0, dtype=native.HIT_DTYPE)
    return a[rng.permutation(len(a))] if len(a) else a


def _run_harness(exe, tmp_path, raw, nfr, min_neighbors, workers, repeat=1):
    import os
    import subprocess

    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as fh:
        fh.write(np.array([nfr, len(raw), min_neighbors, 5], dtype=np.int32).tobytes())
        fh.write(np.ascontiguousarray(raw).tobytes())
    r = subprocess.run([exe, fin, fout, str(workers), str(repeat)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
                                TSAN_OPTIONS="halt_on_error=1:exitcode=66"))
    assert r.returncode == 0, r.stderr[-3000:]
    blob = open(fout, "rb").read()
    ok = int(np.frombuffer(blob[:4], dtype=np.uint32)[0])
    if not ok:
        return 0, None, None, None
    o = 4
    hits = np.frombuffer(blob[o:o + len(raw) * 24], dtype=native.HIT_DTYPE); o += len(raw) * 24
    counts = np.frombuffer(blob[o:o + 4 * nfr], dtype=np.uint32); o += 4 * nfr
    best = np.frombuffer(blob[o:o + 48 * nfr], dtype=native.RECT_DTYPE)
    return ok, hits, counts, best


@pytest.mark.parametrize("seed,nfr,faces,strays", [(1, 256, 2, 3), (2, 64, 1, 0), (3, 300, 3, 6), (4, 1, 2, 2), (5, 40, 0, 0)])
def test_host_pass_under_asan_equals_oracle_for_any_worker_count(harness, tmp_path, seed, nfr, faces, strays):
    rng = np.random.default_rng(seed)
    raw = _raw_hits(rng, nfr, faces, strays)
    # the oracle's answer: per frame, hits in emission order (scale, q, y, x), seq rects, grouping, strict-'>' best face
    order = np.lexsort((raw["x"], raw["y"], raw["q"], raw["scale"], raw["frame"])) if len(raw) else np.zeros(0, dtype=np.int64)
    want_hits = raw[order]
    want_counts = np.bincount(raw["frame"], minlength=nfr).astype(np.uint32) if len(raw) else np.zeros(nfr, dtype=np.uint32)
    want_best = np.zeros(nfr, dtype=ho.RECT_DTYPE)
    k = 0
    for f in range(nfr):
        h = want_hits[k:k + int(want_counts[f])]
        k += int(want_counts[f])
        oh = np.zeros(len(h), dtype=ho.HIT_DTYPE)
        for fld in ("scale", "q", "x", "y", "sum"):
            oh[fld] = h[fld]
        g = ho.group(ho.hits_to_rects(oh), 1)
        want_best[f]["confidence"] = -10000.0
        for i in range(len(g)):
            if i == 0 or g[i]["confidence"] > want_best[f]["confidence"]:
                want_best[f] = g[i]
    outs = []
    for workers in (0, 3, 7):
        ok, hits, counts, best = _run_harness(harness, tmp_path, raw, nfr, 1, workers, repeat=3)
        assert ok == 1
        assert hits.tobytes() == want_hits.tobytes() and np.array_equal(counts, want_counts)
        for fld in ("x", "y", "width", "height", "confidence", "neighbors"):
            assert np.array_equal(best[fld], want_best[fld]), (workers, fld)
        outs.append(best.tobytes())
    assert outs[0] == outs[1] == outs[2]


def test_host_pass_rejects_a_frame_index_outside_the_batch(harness, tmp_path):
    """the counting sort indexes its bucket table with the hit's frame: a corrupt index must be reported, never written through (ASan watches)"""
    rng = np.random.default_rng(9)
    raw = _raw_hits(rng, 8, 2, 2).copy()
    raw["frame"][len(raw) // 2] = 8
    ok, *_ = _run_harness(harness, tmp_path, raw, 8, 1, 3)
    assert ok == 0


def test_pool_with_a_worker_count_that_changes_every_batch_is_race_free(harness_tsan, harness, tmp_path):
    """HtPool decided a worker's participation from `active_` read AFTER the generation it belonged to — a worker
    outside batch g (id >= active) that was preempted there could join batch g + 1 uncounted, decrement pending_ twice, and run()
    could return while it was still inside fn.  Now {generation, participants} travel in one atomic word.  The product varies the
    worker count per batch (ht_host_workers), so this runs 60 batches with the count cycling 7, 3, 7, 1, 5, 0, 2 inside ONE process
    — under ThreadSanitizer (no data race, no use of a dead batch's lambda) and under ASan — and compares with the oracle."""
    from oracle import ht_oracle as ho

    rng = np.random.default_rng(77)
    nfr = 96
    raw = _raw_hits(rng, nfr, 2, 3)
    want = None
    for exe in (harness_tsan, harness):
        ok, hits, counts, best = _run_harness(exe, tmp_path, raw, nfr, 1, "7,3,7,1,5,0,2", repeat=60)
        assert ok == 1
        if want is None:
            want = (hits.copy(), counts.copy(), best.copy())
            ok0, h0, c0, b0 = _run_harness(harness, tmp_path, raw, nfr, 1, 0)
            assert ok0 == 1
            assert h0.tobytes() == hits.tobytes() and c0.tobytes() == counts.tobytes() and b0.tobytes() == best.tobytes()
        else:
            assert hits.tobytes() == want[0].tobytes() and counts.tobytes() == want[1].tobytes() and best.tobytes() == want[2].tobytes()
