"""Parity of the HIP detect path (gray -> pyramid -> cascade scan) with the CPU oracle and the golden vectors.
Everything goes through the C ABI (headtrackr_amd.api.Context -> libheadtrackr_hip.so).  Bit-exact: bytes, indices
and the binary64 confidence."""
import zlib

import numpy as np
import pytest

from conftest import load_golden
from headtrackr_amd import synth
from headtrackr_amd.api import HT_INPUT_GRAY_IN_R, HT_INPUT_RGBA, HT_SCAN_NO_SPLIT, HT_SCAN_SIMPLE, Context
from headtrackr_amd.native import HT_SCAN_GENERIC, HT_SCAN_STATS
from oracle import ht_oracle as ho

pytestmark = pytest.mark.gpu

DETECT = load_golden("detect.json")
CASES = [c for c in DETECT["cases"] if "interval3" not in c["name"]]


@pytest.fixture(scope="module")
def ctx():
    c = Context()
    yield c
    c.close()


def oracle_hits(frame, cascade, frame_index=0, interval=5):
    h = ho.detect_raw(frame, cascade.blob, interval=interval)
    out = np.zeros(len(h), dtype=[("frame", "<u4"), ("scale", "<i4"), ("q", "<i4"), ("y", "<i4"), ("x", "<i4"), ("sum", "<f8")])
    out["frame"] = frame_index
    for k in ("scale", "q", "x", "y", "sum"):
        out[k] = h[k]
    return out


def assert_hits_equal(gpu_hits, ref):
    assert len(gpu_hits) == len(ref), f"{len(gpu_hits)} hits on the GPU, {len(ref)} from the oracle"
    for k in ("frame", "scale", "q", "y", "x"):
        assert np.array_equal(gpu_hits[k].astype(np.int64), ref[k].astype(np.int64)), k
    assert np.array_equal(gpu_hits["sum"].view(np.uint64), ref["sum"].view(np.uint64)), "confidence bits differ"


def test_grayscale_exhaustive_2p24(ctx):
    """every (R,G,B) triple: ccv.js:29 in binary64 + round-half-even, vs the oracle"""
    v = np.arange(1 << 24, dtype=np.uint32)
    rgba = np.empty((1, 4096, 4096, 4), dtype=np.uint8)
    flat = rgba.reshape(-1, 4)
    flat[:, 0] = v & 0xFF
    flat[:, 1] = (v >> 8) & 0xFF
    flat[:, 2] = (v >> 16) & 0xFF
    flat[:, 3] = (v * 7) & 0xFF  # alpha must be preserved
    got = ctx.grayscale(rgba)[0]
    want = ho.grayscale_rgba(rgba[0])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_golden_case(ctx, case, cascade):
    """the committed reference-JS vectors: pyramid planes by CRC, raw hits, grouped rects"""
    w, h = case["w"], case["h"]
    frame = synth.make(case["gen"], w, h)
    assert zlib.crc32(frame.tobytes()) == case["input_crc"]
    hits, counts = ctx.detect_raw(frame)
    order = [(i, 0) for i in range(1, ctx.num_levels)] + [(i, s) for i in range(12, ctx.num_levels) for s in (1, 2, 3)]
    assert zlib.crc32(ctx.pyramid_readback(0, 0, 0).tobytes()) == case["gray_crc"]
    for (i, s), g in zip(order, case["pyramid"]):
        p = ctx.pyramid_readback(0, i, s)
        assert (p.shape[1], p.shape[0]) == (g["w"], g["h"])
        assert zlib.crc32(p.tobytes()) == g["crc"], f"pyramid level {i} slot {s}"
    rects = ctx.hits_to_rects(hits)
    assert len(rects) == len(case["raw"])
    for r, g in zip(rects, case["raw"]):
        for k in ("x", "y", "width", "height", "confidence"):
            assert r[k] == g[k], (k, r, g)
    grouped = ctx.group_rects(rects, case["min_neighbors"])
    assert len(grouped) == len(case["grouped"])
    for r, g in zip(grouped, case["grouped"]):
        for k in ("x", "y", "width", "height", "confidence", "neighbors"):
            assert r[k] == g[k], (k, r, g)
    ctx.upload(frame[None])
    assert ctx.whitebalance()[0] == case["whitebalance"]


def _check_pyramid(ctx, w, h):
    frames = np.stack([synth.noise_frame(w, h, 3), synth.smooth_frame(w, h, 4), synth.face_frame(w, h, [(w // 8, h // 8, min(w, h) // 2)])])
    ctx.set_geometry(w, h, len(frames))
    ctx.upload(frames)
    ctx.detect_enqueue(HT_INPUT_RGBA)
    ctx.detect_collect()
    for f in range(len(frames)):
        levels, arena = ho.pyramid(frames[f])
        assert ctx.num_levels == len(levels)
        for i, (lw, lh, off) in enumerate(levels):
            for s in range(4):
                if off[s] < 0:
                    continue
                got = ctx.pyramid_readback(f, i, s)
                want = ho.plane(levels, arena, i, s)
                assert got.shape == want.shape and np.array_equal(got, want), f"frame {f} level {i} slot {s}"


@pytest.mark.parametrize("w,h", [(320, 240), (201, 157), (64, 48), (38, 30), (641, 363), (1280, 720), (1920, 1080)])
def test_pyramid_planes_vs_oracle(ctx, w, h):
    """all 119 canvases of an N, an S and an F frame, plane by plane; 1920x1080 is the C5 geometry (the binary32-estimate fast path of
    k_resample and the tile split are size-dependent)"""
    _check_pyramid(ctx, w, h)


def _random_sizes(n, seed):
    rng = np.random.RandomState(seed)
    return [(int(rng.randint(24, 1100)), int(rng.randint(24, 800))) for _ in range(n)]


@pytest.mark.parametrize("w,h", _random_sizes(16, 20260921) + [(1023, 65), (65, 799), (129, 513), (512, 512), (2048, 64)])
def test_pyramid_planes_random_geometries(ctx, w, h):
    """k_resample's flat frame loop is specialised on the pass count of a tile (1-4) and on the exact-2:1 box mode, masks the pixels
    outside the drawn rect byte by byte and reads clamped taps for undrawn pixels: seeded random and extreme aspect ratios hit every
    combination of partial tiles, partial passes, odd / even parents and narrow last columns.  All planes of an N, an S and an F frame."""
    _check_pyramid(ctx, w, h)


@pytest.mark.parametrize("w,h", [(320, 240), (201, 157), (38, 30)])
def test_pyramid_without_tail_kernel(w, h):
    """The last (tiny) generations are built by k_resample_tail by default; with option rs_notail every generation goes
    through k_resample.  Both must produce the oracle's planes bit for bit."""
    c = Context(options="rs_notail=1")
    try:
        _check_pyramid(c, w, h)
    finally:
        c.close()


@pytest.mark.parametrize("opts", ["rs_bands=1", "rs_bands=0", "rs_bands=0,rs_notail=1", "rs_bands=0,rs_nofast=1", "rs_bands=0,rs_rpt=2", "rs_bands=0,rs_k=3",
                                  "rs_bands=1,rs_notail=1", "rs_bands=1,rs_rpt=2", "rs_bands=1,rs_k=3"])
@pytest.mark.parametrize("w,h", [(320, 240), (201, 157), (38, 30), (641, 363), (1280, 720), (1920, 1080)] + _random_sizes(6, 20260930) + [(1023, 65), (65, 799), (2048, 64)])
def test_pyramid_both_generation_kernels(w, h, opts):
    """The pyramid generations are built by k_resample_bands (the default: LDS-DMA into wave-private source bands, no barrier in the frame
    loop) or by k_resample (option rs_bands=0: register-staged tile, two barriers per frame): the same tile records, the same pixel
    arithmetic, another thread -> row map.  Forced here on the same inputs, with every generation going through the kernel (rs_notail),
    the binary64-everywhere mode, smaller tiles and an odd frame group: every plane equals the oracle's."""
    if "=1," in opts and (w, h) not in [(320, 240), (201, 157), (641, 363), (1280, 720)]:
        pytest.skip("variants of the bands kernel: four geometries")
    c = Context(options=opts)
    try:
        _check_pyramid(c, w, h)
    finally:
        c.close()


@pytest.mark.parametrize("mode", [0, HT_SCAN_NO_SPLIT, HT_SCAN_SIMPLE, HT_SCAN_GENERIC, HT_SCAN_GENERIC | HT_SCAN_NO_SPLIT],
                         ids=["gen+deep", "gen-nosplit", "simple", "generic+deep", "generic-nosplit"])
def test_mixed_batch_hits_vs_oracle(ctx, cascade, mode):
    """N / S / F frames in one batch, every scan schedule: raw hits == oracle, in the reference's order"""
    w, h, n = 320, 240, 12
    frames = synth.mixed_batch(n, w, h, seed0=1234)
    hits, counts = ctx.detect_raw(frames, flags=mode | HT_SCAN_STATS)
    ref = np.concatenate([oracle_hits(frames[i], cascade, i) for i in range(n)])
    assert_hits_equal(hits, ref)
    assert int(counts.sum()) == len(ref) and len(ref) > 0
    # stage survival statistics are part of the measurement contract: compare with the oracle's counters
    sp = np.zeros(cascade.count + 1, dtype=np.int64)
    for i in range(n):
        ho.detect_raw(frames[i], cascade.blob, stage_pass=sp)
    assert np.array_equal(ctx.stage_counts().astype(np.int64), sp)
    assert ctx.windows_per_frame * n == sp[0]
    hits2, _ = ctx.detect_raw(frames, flags=mode)  # same results without the statistics counters
    assert hits2.tobytes() == hits.tobytes()


@pytest.mark.parametrize("fp_sparse", [0, 1])
def test_sparse_stage_schedules_return_the_same_hits(cascade, fp_sparse):
    """option fp_sparse: the tile kernel's sparse stages as four feature slices (0) or one lane per (window, feature) pair when <= 256
    pairs are left (1, the default) — the header promises identical results for every option key: raw hits incl. the binary64
    confidence and the stage counters equal the oracle's with either (only the default had a parity test)."""
    w, h, n = 320, 240, 24
    frames = synth.mixed_batch(n, w, h, seed0=777)
    c = Context(options=f"fp_sparse={fp_sparse}")
    try:
        hits, counts = c.detect_raw(frames, flags=HT_SCAN_STATS)
        ref = np.concatenate([oracle_hits(frames[i], cascade, i) for i in range(n)])
        assert len(ref) > 50
        assert_hits_equal(hits, ref)
        sp = np.zeros(cascade.count + 1, dtype=np.int64)
        for i in range(n):
            ho.detect_raw(frames[i], cascade.blob, stage_pass=sp)
        assert np.array_equal(c.stage_counts().astype(np.int64), sp)
    finally:
        c.close()


@pytest.mark.parametrize("grid", [1, 2, 3, 17])
def test_deep_kernel_grid_option_never_loses_queue_entries(cascade, grid):
    """k_scan_deep_lds hands queue entries out through 16 work counters, counter c serving the entries
    nwaves + 16 k + c — with option deep_grid=1 (12 wavefronts) counters 12..15 had no wavefront and their entries were silently
    skipped.  The launch now keeps the grid at >= 16 wavefronts; every grid size returns the oracle's hits."""
    w, h, n = 320, 240, 24
    frames = synth.mixed_batch(n, w, h, seed0=4321)
    c = Context(options=f"deep_grid={grid}")
    try:
        hits, counts = c.detect_raw(frames)
        ref = np.concatenate([oracle_hits(frames[i], cascade, i) for i in range(n)])
        assert len(ref) > 50
        assert_hits_equal(hits, ref)
    finally:
        c.close()


def test_destroying_the_owner_of_a_shared_frame_buffer_keeps_it_alive_for_its_binders(cascade):
    """ht_destroy of the context that ht_device_alloc'ed a frame buffer used to free it under the other contexts
    bound inside it (dangling d_frames + graphs keyed on it).  The buffer now outlives its owner until no live context is bound to
    it: the binder still detects the same hits after the owner is gone — from fresh launches and from its replayed graph."""
    import ctypes as C

    from headtrackr_amd import native

    L = native.lib()
    w, h, n = 320, 240, 8
    frames = np.ascontiguousarray(synth.mixed_batch(n, w, h, seed0=99))
    ref = np.concatenate([oracle_hits(frames[i], cascade, i) for i in range(n)])
    owner, binder = Context(), Context()
    try:
        p = C.c_void_p()
        assert L.ht_device_alloc(owner._h, frames.nbytes, C.byref(p)) == 0
        assert L.ht_device_upload(owner._h, p, frames.ctypes.data, frames.nbytes) == 0
        for cx in (owner, binder):
            cx.set_geometry(w, h, n)
            cx.bind_device(p.value, n)
        for _ in range(4):  # the third enqueue onwards replays the captured graph
            binder.detect_enqueue(0)
            hits, _ = binder.detect_collect()
        assert_hits_equal(hits, ref)
        assert L.ht_device_free(owner._h, p) != 0  # refused while the binder is bound
        owner.close()  # the owner goes first: the buffer must stay valid
        junk = [Context() for _ in range(2)]  # allocations that would reuse freed HBM
        for j in junk:
            j.set_geometry(w, h, n)
            j.upload(frames[::-1].copy())
            j.detect_enqueue(0)
            j.detect_collect()
        for _ in range(3):
            binder.detect_enqueue(0)
            hits, _ = binder.detect_collect()
            assert_hits_equal(hits, ref)
        for j in junk:
            j.close()
    finally:
        owner.close()
        binder.close()  # releases the orphaned buffer


def test_orphaned_frame_buffer_is_released_when_its_last_binder_moves_away(cascade):
    """a shared frame buffer that outlived its owner (an orphan) used to be reclaimed only inside a LATER ht_destroy — a
    host whose last binder re-binds elsewhere and never destroys a context kept the HBM for the life of the process.  Orphans are now
    swept whenever a context's frames move: 512 MB come back as soon as the binder uploads frames of its own."""
    import ctypes as C

    import torch

    from headtrackr_amd import native

    L = native.lib()
    w, h, n = 320, 240, 4
    frames = np.ascontiguousarray(synth.mixed_batch(n, w, h, seed0=5))
    ref = np.concatenate([oracle_hits(frames[i], cascade, i) for i in range(n)])
    big = 512 << 20
    owner, binder = Context(), Context()
    try:
        torch.cuda.synchronize()
        p = C.c_void_p()
        assert L.ht_device_alloc(owner._h, big, C.byref(p)) == 0
        assert L.ht_device_upload(owner._h, p, frames.ctypes.data, frames.nbytes) == 0
        binder.set_geometry(w, h, n)
        binder.bind_device(p.value, n)
        binder.detect_enqueue(0)
        hits, _ = binder.detect_collect()
        assert_hits_equal(hits, ref)
        owner.close()  # the buffer becomes an orphan: the binder is still bound inside it
        binder.detect_enqueue(0)  # still valid memory
        hits, _ = binder.detect_collect()
        assert_hits_equal(hits, ref)
        torch.cuda.synchronize()
        free_orphaned = torch.cuda.mem_get_info()[0]
        binder.upload(frames)  # the last binder moves to a buffer of its own: the orphan goes, here and now
        assert torch.cuda.mem_get_info()[0] > free_orphaned + big // 2
        binder.detect_enqueue(0)
        hits, _ = binder.detect_collect()
        assert_hits_equal(hits, ref)
    finally:
        owner.close()
        binder.close()


def test_gray_in_r_entry(ctx, cascade):
    """HT_INPUT_GRAY_IN_R == calling ccv.detect_objects on an already gray canvas"""
    frame = synth.face_frame(320, 240, [(100, 60, 96)])
    gray = ho.grayscale_rgba(frame)
    a, _ = ctx.detect_raw(frame, flags=HT_INPUT_RGBA)
    b, _ = ctx.detect_raw(gray, flags=HT_INPUT_GRAY_IN_R)
    assert len(a) == 9 and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("cap", ["", ",rs_tailcap=32768", ",rs_tailcap=1000"], ids=["small-batch-cap", "cap-32768", "cap-1000"])
@pytest.mark.parametrize("table", ["0", "1", "2"], ids=["binary64-tail", "table-tail", "table-tail-small"])
@pytest.mark.parametrize("w,h", [(320, 240), (201, 157), (38, 30)])
def test_pyramid_both_tail_kernels(w, h, table, cap):
    """The last generations are built by one of the tail kernels, chosen by batch size: the round-1 one (taps re-derived in
    registers, binary64 lerps) and the table-driven one (host tap tables, binary32 estimate + binary64 fallback, integer box
    means; compact taps in LDS, or — the small-footprint form — read from L2).  Forced here on the same inputs: every plane
    equals the oracle's with each.  Which generations the tail takes depends on its pixel cap — 32 768 for batches that fill the chip,
    4 000 for batches of <= 16 frames like this one (round 6), forced here to both and to 1 000."""
    c = Context(options=f"rs_tailtable={table}{cap}")
    try:
        _check_pyramid(c, w, h)
    finally:
        c.close()


@pytest.mark.parametrize("w,h", [(320, 240), (201, 157), (1280, 720), (1920, 1080)])
def test_pyramid_fast_paths_equal_the_declared_binary64_sequence(w, h):
    """k_resample evaluates a pixel in binary32 and falls back to the declared binary64 sequence next to a rounding boundary;
    exact 2:1 canvases are integer box means.  Option rs_nofast keeps every pixel on the binary64 sequence: all planes of
    both builds must be identical (and both equal the oracle, test_pyramid_planes_vs_oracle)."""
    frames = np.stack([synth.noise_frame(w, h, 11), synth.smooth_frame(w, h, 12), synth.face_frame(w, h, [(w // 8, h // 8, min(w, h) // 2)])])

    def planes(options=None):
        c = Context(options=options)
        try:
            c.set_geometry(w, h, len(frames))
            c.upload(frames)
            c.detect_enqueue(HT_INPUT_RGBA)
            c.detect_collect()
            out = []
            for f in range(len(frames)):
                for i in range(c.num_levels):
                    for s in range(4):
                        if c.plane(i, s).present:
                            out.append(c.pyramid_readback(f, i, s).tobytes())
            return out
        finally:
            c.close()

    fast = planes()
    slow = planes("rs_nofast=1")
    assert len(fast) == len(slow) == 3 * 120 and fast == slow


def test_720p_vs_oracle(ctx, cascade):
    w, h = 1280, 720
    frames = np.stack([synth.smooth_frame(w, h, 77), synth.face_frame(w, h, [(400, 200, 240), (900, 100, 64), (100, 500, 150)])])
    hits, counts = ctx.detect_raw(frames)
    ref = np.concatenate([oracle_hits(frames[i], cascade, i) for i in range(2)])
    assert_hits_equal(hits, ref)
    assert ctx.windows_per_frame == 1007428


def test_full_batch_is_frame_independent(ctx, cascade):
    """C2 size (256 x 320x240): every frame's hits equal those of the same frame detected alone (size-independent
    property; the oracle checks a sample)"""
    w, h, n = 320, 240, 256
    frames = synth.mixed_batch(n, w, h, seed0=1234)
    hits, counts = ctx.detect_raw(frames)
    assert len(counts) == n
    starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    for i in (0, 1, 2, 17, 128, 255):
        alone, _ = ctx.detect_raw(frames[i])
        got = hits[starts[i] : starts[i + 1]].copy()
        got["frame"] = 0
        assert got.tobytes() == alone.tobytes()
        assert_hits_equal(alone, oracle_hits(frames[i], cascade, 0))
    # frames of family F must have produced detections, N must not
    assert all(counts[i] == 0 for i in range(0, n, 3))
    assert sum(int(counts[i] > 0) for i in range(2, n, 3)) > n // 3 * 0.9


def test_early_scan_on_second_stream(cascade):
    """Option early_scan=1: scale 0 is scanned on a second HIP stream while the late pyramid generations are built; same hits
    (two tile launches + event dependencies instead of one launch)."""
    frames = synth.mixed_batch(12, 320, 240, seed0=1234)
    c = Context(options="early_scan=1")
    try:
        for _ in range(3):  # back-to-back batches on one context: the next batch must not overtake the second stream
            hits, _ = c.detect_raw(frames)
        ref = np.concatenate([oracle_hits(frames[i], cascade, i) for i in range(12)])
        assert_hits_equal(hits, ref)
    finally:
        c.close()


def test_collect_best_requeue_matches_collect_best(cascade):
    """ht_detect_collect_best_requeue: same best faces as ht_detect_collect_best, and the next batch is already enqueued"""
    frames = synth.mixed_batch(24, 320, 240, seed0=4321)
    c = Context()
    try:
        c.set_geometry(320, 240, 24)
        c.upload(frames)
        c.detect_enqueue()
        ref, nref = c.detect_collect_best(1)
        ref = ref.copy()
        c.detect_enqueue()
        for _ in range(3):  # every call collects one batch and starts the next
            got, n = c.detect_collect_best_requeue(1)
            assert n == nref and got.tobytes() == ref.tobytes()
        got, n = c.detect_collect_best(1)  # the batch the last requeue started
        assert n == nref and got.tobytes() == ref.tobytes()
        with pytest.raises(Exception):
            c.detect_collect_best(1)  # nothing enqueued any more
    finally:
        c.close()


def test_tiny_hit_capacity_reports_overflow(cascade):
    from headtrackr_amd.api import HtError

    c = Context(hit_capacity=4)
    with pytest.raises(HtError) as e:
        c.detect_raw(synth.face_frame(320, 240, [(100, 60, 96)]))
    assert e.value.status == -4
    c.close()


def test_queue_overflow_falls_back_inline(cascade):
    """a survivor queue that is far too small: the tile kernel must finish the overflow itself, results unchanged"""
    c = Context(queue_capacity=8)
    frames = synth.mixed_batch(6, 320, 240, seed0=1234)
    hits, _ = c.detect_raw(frames)
    ref = np.concatenate([oracle_hits(frames[i], cascade, i) for i in range(6)])
    assert_hits_equal(hits, ref)
    c.close()


def test_best_faces_matches_per_frame_grouping(ctx):
    frames = synth.mixed_batch(9, 320, 240, seed0=1234)
    hits, counts = ctx.detect_raw(frames)
    best = ctx.best_faces(hits, counts, 1)
    per = ctx.detect_objects(frames, min_neighbors=1)
    for f in range(9):
        if len(per[f]) == 0:
            assert best[f]["neighbors"] == 0 and best[f]["confidence"] == -10000.0
        else:
            b = per[f][int(np.argmax(per[f]["confidence"]))]
            assert best[f].tobytes() == b.tobytes()


@pytest.mark.parametrize("deep_v", ["4", "2"])
def test_exact_tie_fallback_path(ctx, cascade, deep_v):
    """The integer stage decisions fall back to the sequential binary64 sum on an exact tie with the threshold — a case the
    built-in cascade practically never produces.  Option force_exact makes every decision take that fallback (tile kernel
    incl. its sparse phase, and both deep kernels): results must be unchanged."""
    frames = synth.mixed_batch(6, 320, 240, seed0=1234)
    want, _ = ctx.detect_raw(frames)
    forced = Context(options=f"force_exact=1,deep_v={deep_v}")
    try:
        got, _ = forced.detect_raw(frames)
    finally:
        forced.close()
    assert len(want) > 20 and got.tobytes() == want.tobytes()
    ref = np.concatenate([oracle_hits(frames[i], cascade, i) for i in range(6)])
    assert_hits_equal(got, ref)


def test_async_upload_and_swap(cascade):
    """ht_upload_frames_async + ht_swap_frames (double-buffered ingest): the copy of the next batch runs on the copy stream
    while the current batch is scanned; results must equal plain uploads, also after several ping-pong swaps."""
    A = np.ascontiguousarray(synth.mixed_batch(4, 320, 240, seed0=1234))
    B = np.ascontiguousarray(synth.mixed_batch(4, 320, 240, seed0=4321))
    c = Context()
    try:
        c.set_geometry(320, 240, 4)
        want_a, _ = c.detect_raw(A)
        want_b, _ = c.detect_raw(B)
        assert len(want_a) > 0 and want_a.tobytes() != want_b.tobytes()
        c.upload(A)
        for k in range(4):
            nxt, cur_want = (B, want_a) if k % 2 == 0 else (A, want_b)
            c.upload_async_ptr(nxt.ctypes.data, 4)   # next batch on the copy stream ...
            c.detect_enqueue(HT_INPUT_RGBA)         # ... while the current one is scanned
            got, _ = c.detect_collect()
            assert got.tobytes() == cur_want.tobytes(), k
            c.swap_frames()
        c.detect_enqueue(HT_INPUT_RGBA)
        got, _ = c.detect_collect()
        assert got.tobytes() == want_a.tobytes()
    finally:
        c.close()


def test_options_are_per_context_and_result_changing_keys_are_rejected(cascade):
    """ht_config.options: an unknown key, a malformed value and — in the product build — the keys that make results incomplete by design
    (they exist only in -DHT_DEBUG_KNOBS builds) fail ht_create with HT_ERR_INVALID; a dict is accepted; an ABI-1 caller (struct without the
    options member) still gets a context."""
    import ctypes as C

    from headtrackr_amd import native
    from headtrackr_amd.api import HtError

    for bad in ("nonsense=1", "cs_fused_min=abc", "stop_stage=3", "cs_iters=2", "rs_maxgen=1"):
        with pytest.raises(HtError) as e:
            Context(options=bad)
        assert e.value.status == -1, bad
    frames = synth.mixed_batch(3, 320, 240, seed0=1234)
    a = Context(options={"graph_max_frames": 0, "host_threads": 3})
    b = Context()
    try:
        assert a.detect_raw(frames)[0].tobytes() == b.detect_raw(frames)[0].tobytes()
    finally:
        a.close()
        b.close()
    L = native.lib()
    cfg = native.Config(32, 0, 5, 0, None, 0, 0, None)  # ABI 1: struct_size stops before `options`
    h = C.c_void_p()
    assert L.ht_create(C.byref(cfg), cascade.blob, len(cascade.blob), C.byref(h)) == 0
    L.ht_destroy(h)
