"""Parity at the C5 shape (BASELINE.json configs[4]; /root/reference/src/main.js:168-305 -> facetrackr.js:97-108,185-217 ->
camshift.js:198-259): 8 frame-synchronous 1920x1080 feeds as ONE batch per time step on ONE context, issued exactly the way
bench.py's `stream_bench` issues them — bind_device to a different set of frames every step, detect by enqueue / collect-best
(a hipGraph REPLAY from the second cycle on), camshift.initTracker on the floored best face, then enqueue-only
ht_camshift_track_batch + ht_camshift_track_collect on 8 streams x 32 cluster workgroups — every best face and every track object
against the oracle.  The same from the Node host: tests/js/parity_gpu.js (DeviceBatch.initTrackers / trackSequence)."""
import math

import numpy as np
import pytest

from headtrackr_amd import synth
from headtrackr_amd.api import Context
from oracle import ht_oracle as ho
from test_gpu_camshift import assert_all_exact as cs_all_exact, check as cs_check

pytestmark = pytest.mark.gpu

W, H, NUNIQ = 1920, 1080, 30


def c5_device_steps(uniq, feeds):
    """[NUNIQ, feeds] frames in HBM laid out like bench.py's `dev` tensor: step k's batch = the feeds' frames of that time step"""
    from hipmem import DeviceArray, _rt

    fb = uniq[0].nbytes
    d = DeviceArray.__new__(DeviceArray)
    import ctypes as C

    p = C.c_void_p()
    assert _rt().hipMalloc(C.byref(p), fb * NUNIQ * feeds) == 0
    d.ptr, d.nbytes = p.value, fb * NUNIQ * feeds
    for k in range(NUNIQ):
        for f in range(feeds):
            src = uniq[synth.stream_frame_index(k, f, NUNIQ)]
            assert _rt().hipMemcpy(d.ptr + (k * feeds + f) * fb, src.ctypes.data, fb, 1) == 0
    return d


def floored_rects(best, w, h):
    """facetrackr.js:101-106 (Math.floor of the detection) with bench.py's stand-in rect for a feed without a face"""
    fl = np.floor(np.stack([best["x"], best["y"], best["width"], best["height"]], axis=1)).astype(np.int64)
    return [tuple(int(v) for v in fl[f]) if best["neighbors"][f] > 0 and best["confidence"][f] > -10 else (w // 4, h // 4, w // 2, h // 2)
            for f in range(len(best))]


@pytest.mark.parametrize("feeds", [8, 3])
def test_c5_shape_feeds_in_one_context_vs_oracle(cascade, feeds):
    """61 steps = two 30-step cycles + the third cycle's detect: steps 0 / 30 / 60 detect (the 2nd is captured into a hipGraph, the
    3rd replays it), all others track.  Every detect step's best face per feed == the oracle's bit for bit; every track object within
    +-1 px / +-0.5 deg (sizes equal) and, the reduction tree being fixed, bit-exact — counted."""
    K, steps = feeds, 61
    uniq = synth.stream_feed_frames(NUNIQ, W, H, 0)
    c = Context()  # first: ht_create selects the device
    dev = c5_device_steps(uniq, K)
    sbytes = K * W * H * 4
    try:
        c.set_geometry(W, H, K)
        c.camshift_reserve(K)
        want_best = {}  # unique frame -> oracle best face (a feed's detect frames repeat every cycle)
        oracles = [None] * K
        stats = []
        for i in range(steps):
            c.bind_device(dev.ptr + (i % NUNIQ) * sbytes, K)
            if i % 30 == 0:
                c.detect_enqueue(0)
                best = c.detect_collect_best(1)[0].copy()
                rects = floored_rects(best, W, H)
                c.camshift_init(rects)
                for f in range(K):
                    u = synth.stream_frame_index(i, f, NUNIQ)
                    if u not in want_best:
                        want_best[u] = ho.best_faces(uniq[u : u + 1], cascade.blob, 1)[0]
                    for k in ("x", "y", "width", "height", "confidence", "neighbors"):
                        assert best[k][f] == want_best[u][k], (i, f, k, best[f], want_best[u])
                    assert best["neighbors"][f] > 0, (i, f)
                    oracles[f] = ho.Camshift(True)
                    oracles[f].init_tracker(uniq[u], rects[f])
            else:
                assert c.camshift_track(K, calc_angles=True, fetch=False) is not None  # enqueue only
                got = c.camshift_track_collect(K)
                for f in range(K):
                    sw, to = oracles[f].track(uniq[synth.stream_frame_index(i, f, NUNIQ)])
                    cs_check(got[f], sw, to, stats, where=("c5", K, f, i))
        assert c.graph_launches >= 2, c.graph_launches  # step 30: capture + launch, step 60: replay
        assert len(stats) == K * (steps - 3)
        cs_all_exact(stats, f"C5 shape, {K} x 1080p feeds in one context, {steps} steps")
    finally:
        c.close()
        dev.free()


@pytest.mark.parametrize("cs_flags", [1, 0], ids=["marks", "event"])
@pytest.mark.parametrize("feeds", [8, 1])
def test_c5_pipelined_loop_as_the_bench_issues_it(cascade, feeds, cs_flags):
    """bench.py's TIMED C5 loop (round 5): two track steps outstanding; a detect step is enqueued right behind them BEFORE their
    results are collected, its best faces are collected after, initTracker no longer waits for the stream (rects staged in pinned
    memory) and the next track step is enqueued right behind it; completion of an enqueue-only track call is a mark the kernel writes
    into the pinned slot (no event; option cs_flags=0 — the event completion path of the cluster track — returns the same objects).
    91 steps = three detect steps; every best face and every track object == the oracle's."""
    K, steps = feeds, 91
    uniq = synth.stream_feed_frames(NUNIQ, W, H, 0)
    c = Context(options=f"cs_flags={cs_flags}")
    dev = c5_device_steps(uniq, K)
    sbytes = K * W * H * 4
    try:
        c.set_geometry(W, H, K)
        c.camshift_reserve(K)
        results = {}
        pend = []

        def collect(i):
            if i % 30 == 0:
                best = c.detect_collect_best(1)[0].copy()
                c.camshift_init(floored_rects(best, W, H))
                results[i] = best
            else:
                results[i] = c.camshift_track_collect(K).copy()

        for i in range(steps):
            c.bind_device(dev.ptr + (i % NUNIQ) * sbytes, K)
            if i % 30 == 0:
                c.detect_enqueue(0)
                while pend:
                    collect(pend.pop(0))
                collect(i)
            else:
                c.camshift_track(K, calc_angles=True, fetch=False)
                pend.append(i)
                if len(pend) > 1:
                    collect(pend.pop(0))
        while pend:
            collect(pend.pop(0))
        oracles = [None] * K
        stats = []
        for i in range(steps):
            got = results[i]
            for f in range(K):
                fr = uniq[synth.stream_frame_index(i, f, NUNIQ)]
                if i % 30 == 0:
                    want = ho.best_faces(fr[None], cascade.blob, 1)[0]
                    for k in ("x", "y", "width", "height", "confidence", "neighbors"):
                        assert got[k][f] == want[k], (i, f, k)
                    oracles[f] = ho.Camshift(True)
                    oracles[f].init_tracker(fr, floored_rects(got, W, H)[f])
                else:
                    sw, to = oracles[f].track(fr)
                    cs_check(got[f], sw, to, stats, where=("c5-pipelined", K, f, i))
        assert len(stats) == K * (steps - 4)
        cs_all_exact(stats, f"C5 pipelined loop, {K} feed(s), {steps} steps")
    finally:
        c.close()
        dev.free()


def test_enqueue_only_track_calls_pipeline_through_the_result_ring(cascade):
    """bench.py's timed C5 loop keeps TWO track steps outstanding (step i + 1 is enqueued before step i is collected); the library allows
    four (results in a ring of pinned slots the kernels write directly).  Here: 4 feeds x 1080p, detect + initTracker, then the 29 track
    steps of a cycle enqueued 4 / 2 / 1 deep — every collected object is the oracle's for THAT step (oldest first), a fifth outstanding
    call and a collect with nothing pending are refused with HT_ERR_STATE, and initTracker between enqueue and collect loses nothing."""
    from headtrackr_amd.api import HtError

    K = 4
    uniq = synth.stream_feed_frames(NUNIQ, W, H, 0)
    c = Context()
    dev = c5_device_steps(uniq, K)
    sbytes = K * W * H * 4
    try:
        c.set_geometry(W, H, K)
        c.camshift_reserve(K)
        with pytest.raises(HtError):
            c.camshift_track_collect(K)  # nothing pending
        stats = []
        for depth in (4, 2, 1):
            c.bind_device(dev.ptr, K)
            c.detect_enqueue(0)
            best = c.detect_collect_best(1)[0].copy()
            rects = floored_rects(best, W, H)
            c.camshift_init(rects)
            oracles = []
            for f in range(K):
                o = ho.Camshift(True)
                o.init_tracker(uniq[synth.stream_frame_index(0, f, NUNIQ)], rects[f])
                oracles.append(o)
            pend = []

            def collect_oldest():
                j = pend.pop(0)
                got = c.camshift_track_collect(K)
                for f in range(K):
                    sw, to = oracles[f].track(uniq[synth.stream_frame_index(j, f, NUNIQ)])
                    cs_check(got[f], sw, to, stats, where=("ring", depth, f, j))

            for i in range(1, 30):
                c.bind_device(dev.ptr + (i % NUNIQ) * sbytes, K)
                c.camshift_track(K, calc_angles=True, fetch=False)
                pend.append(i)
                if depth == 4 and len(pend) == 4 and i == 4:
                    with pytest.raises(HtError):
                        c.camshift_track(K, calc_angles=True, fetch=False)  # a fifth outstanding call
                while len(pend) >= depth:
                    collect_oldest()
            while pend:
                collect_oldest()
        assert len(stats) == 3 * 29 * K
        cs_all_exact(stats, "enqueue-only track calls 4 / 2 / 1 deep, 4 x 1080p feeds")
        # initTracker between enqueue and collect: the pending results are still those of the track call
        c.bind_device(dev.ptr + sbytes, K)
        c.camshift_track(K, calc_angles=True, fetch=False)
        want = c.camshift_track(K, calc_angles=True, fetch=True)  # the same frames again, synchronously: the NEXT call of every stream
        c.camshift_init(floored_rects(best, W, H))
        got = c.camshift_track_collect(K)
        assert got.shape == want.shape and all(got["width"] > 0)
    finally:
        c.close()
        dev.free()


def test_graph_replay_of_a_full_c2_batch(cascade):
    """Batches up to 256 frames replay a captured graph by default (the C2 headline batch): the replayed batch's raw hits, per-frame counts and
    best faces are those of plain launches, also through the collect-and-requeue call the bench loop uses."""
    frames = synth.mixed_batch(256, 320, 240, seed0=1234)
    plain = Context(options="graph_max_frames=0")
    c = Context()
    try:
        for cx in (plain, c):
            cx.set_geometry(320, 240, 256)
            cx.upload(frames)
        plain.detect_enqueue(0)
        want, want_counts = plain.detect_collect()
        plain.detect_enqueue(0)
        want_best = plain.detect_collect_best(1)[0].copy()
        for rep in range(3):
            c.detect_enqueue(0)
            got, counts = c.detect_collect()
            assert got.tobytes() == want.tobytes() and np.array_equal(counts, want_counts), rep
        assert c.graph_launches == 2
        c.detect_enqueue(0)
        for rep in range(3):
            best = c.detect_collect_best_requeue(1)[0]
            assert best.tobytes() == want_best.tobytes(), rep
        assert c.detect_collect_best(1)[0].tobytes() == want_best.tobytes()
        assert c.graph_launches == 6 and plain.graph_launches == 0
    finally:
        plain.close()
        c.close()


@pytest.mark.parametrize("wb", [False, True], ids=["plain", "whitebalance"])
@pytest.mark.parametrize("w,h,n", [(320, 240, 3), (322, 241, 2), (1920, 1080, 1)])
def test_graph_replay_equals_plain_enqueue(cascade, wb, w, h, n):
    """a replayed detect graph (same frames pointer / count / flags from the third enqueue on) must give the same raw
    hits, counts and whitebalance sums as plain launches (a context with option graph_max_frames=0), incl. widths that are not multiples of 4
    (whitebalance in its own pass) — and must actually have replayed."""
    from headtrackr_amd.native import HT_DETECT_WHITEBALANCE

    flags = HT_DETECT_WHITEBALANCE if wb else 0
    frames = synth.mixed_batch(max(n, 3), w, h, seed0=91)[-n:]  # ends on a face frame for n == 1
    plain = Context(options="graph_max_frames=0")
    c = Context()
    try:
        for cx in (plain, c):
            cx.set_geometry(w, h, n)
            cx.upload(frames)
        plain.detect_enqueue(flags)
        want, want_counts = plain.detect_collect()
        want_wb = plain.detect_whitebalance() if wb else None
        assert plain.graph_launches == 0
        for rep in range(4):
            c.detect_enqueue(flags)
            got, counts = c.detect_collect()
            assert got.tobytes() == want.tobytes() and np.array_equal(counts, want_counts), rep
            if wb:
                assert np.array_equal(c.detect_whitebalance(), want_wb), rep
        assert c.graph_launches == 3  # 1st plain, 2nd captured + launched, 3rd / 4th replayed
        if wb:
            assert np.array_equal(want_wb, np.array([ho.whitebalance(f) for f in frames]))
        # fewer frames bound into the same context: another graph key (the count is part of it), results still right
        if n > 1:
            c.set_geometry(w, h, n - 1)
            c.upload(frames[: n - 1])
            for rep in range(3):
                c.detect_enqueue(flags)
                got, counts = c.detect_collect()
                assert np.array_equal(counts, want_counts[: n - 1]) and got.tobytes() == want[: int(want_counts[: n - 1].sum())].tobytes()
    finally:
        plain.close()
        c.close()


def test_c5_shape_from_the_node_host(cascade, tmp_path):
    """The same 61-step cycle driven from JavaScript (tests/js/c5_stream.js: ccv.DeviceBatch.detectStep / trackStep through the N-API
    addon): every best face bit-exact, the floored initTracker rects equal, every track object vs the oracle (sizes equal, +-1 px, counted
    exact), and the detect sequence was replayed from its graph."""
    import json
    import os
    import shutil
    import subprocess

    from conftest import ROOT

    node = shutil.which("node")
    if node is None or not os.path.exists(os.path.join(ROOT, "headtrackr_amd", "js", "headtrackr_hip.node")):
        pytest.skip("node or the addon is missing")
    K, steps = 8, 61
    uniq = synth.stream_feed_frames(NUNIQ, W, H, 0)
    raw, outf = tmp_path / "uniq.raw", tmp_path / "out.json"
    uniq.tofile(str(raw))
    r = subprocess.run([node, os.path.join(ROOT, "tests", "js", "c5_stream.js"), "parity", str(raw), str(NUNIQ), str(K), str(steps), str(outf)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    os.unlink(str(raw))
    got = json.loads(outf.read_text())
    assert got["graph_launches"] >= 2 and len(got["steps"]) == steps
    want_best, oracles, stats = {}, [None] * K, []
    for i, s in enumerate(got["steps"]):
        assert s["step"] == i
        if i % 30 == 0:
            best = np.array(s["best"]).reshape(K, 6)
            for f in range(K):
                u = synth.stream_frame_index(i, f, NUNIQ)
                if u not in want_best:
                    want_best[u] = ho.best_faces(uniq[u : u + 1], cascade.blob, 1)[0]
                wb = want_best[u]
                assert list(best[f]) == [wb["x"], wb["y"], wb["width"], wb["height"], wb["confidence"], float(wb["neighbors"])], (i, f)
                rect = tuple(int(math.floor(v)) for v in best[f][:4])
                assert tuple(s["rects"][4 * f : 4 * f + 4]) == rect
                oracles[f] = ho.Camshift(True)
                oracles[f].init_tracker(uniq[u], rect)
        else:
            t = np.array(s["track"]).reshape(K, 9)
            for f in range(K):
                sw, to = oracles[f].track(uniq[synth.stream_frame_index(i, f, NUNIQ)])
                g = dict(x=t[f][0], y=t[f][1], width=t[f][2], height=t[f][3], angle=t[f][4], sw_x=t[f][5], sw_y=t[f][6], sw_width=t[f][7], sw_height=t[f][8])
                cs_check(g, sw, to, stats, where=("c5-node", f, i))
    assert len(stats) == K * (steps - 3)
    cs_all_exact(stats, "C5 shape from Node, 8 x 1080p feeds")
