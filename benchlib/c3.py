"""C3 (BASELINE.json configs[2]): 256 streams of 320x240 — detect once, initTracker, 60 camshift track() calls."""
import numpy as np

from .baseline import cpu_camshift_baseline
from .common import GEOM, HBM_PEAK_GBS, WORKLOAD_TEXT, dominant_roofline, round_stats

NV, CALLS = 4, 60  # frame versions per stream (the face moved by a seeded <= 3 px walk), track() calls per step


def c3_frames(synth, rank, nf, W, H):
    """family F only: every stream has one face; NV versions of each stream's frame; track() call i
    sees version (i + 1) % NV"""
    walk = synth.lcg_stream(4242 + rank, 2 * NV * nf).astype(np.int64) >> 20
    vers = np.empty((NV, nf, H, W, 4), dtype=np.uint8)
    for f in range(nf):
        s0 = 48 + (f * 7) % 80
        x, y = 20 + (f * 13) % (W - s0 - 40), 16 + (f * 29) % (H - s0 - 32)
        for v in range(NV):
            vers[v, f] = synth.face_frame(W, H, [(x, y, s0)])
            x += int(walk[2 * (f * NV + v)] % 7) - 3
            y += int(walk[2 * (f * NV + v) + 1] % 7) - 3
    return vers


def track_exact(g, sw, to):
    """one track() result against the oracle's: search window, x, y, width, height bit-exact, angle to 1e-6 rad"""
    return int([int(g["sw_x"]), int(g["sw_y"]), int(g["sw_width"]), int(g["sw_height"])] == list(sw) and
               all(float(g[q]) == to[q] for q in ("x", "y", "width", "height")) and
               abs(float(g["angle"]) - to["angle"]) < 1e-6)


def c3_bench(env, a, steps, warmup, cpu_seconds=0.0):
    torch, rank, world, local = env.torch, env.rank, env.world, env.local
    from headtrackr_amd import synth
    from headtrackr_amd.api import Context

    W, H, nf = GEOM["c3"]
    nf = a.frames or nf
    vers = c3_frames(synth, rank, nf, W, H)
    dev_vers = [torch.from_numpy(vers[v]).cuda() for v in range(NV)]
    # Three contexts take the steps in turn (own HIP streams, own tracker states): while one batch of streams is in
    # its 60 track() calls (one launch, one workgroup per stream: half of every CU idle) the next steps' detects run
    # on the other contexts.  --pipeline 1 keeps the steps strictly in turn.  Measured (round 4): 6.23 / 6.82 / 6.41 M
    # frames/s at 2 / 3 / 4 steps in flight.
    depth = a.pipeline if a.pipeline > 0 else 3
    ctxs = []
    for _ in range(depth):
        cx = Context(device=local, options=a.options or None)
        cx.set_geometry(W, H, nf)
        cx.bind_device(dev_vers[0].data_ptr(), nf, W * H * 4)
        cx.camshift_reserve(nf)
        ctxs.append(cx)
    ctx = ctxs[0]
    seq_ptrs = [dev_vers[(it + 1) % NV].data_ptr() for it in range(CALLS)]
    state = {}
    pending = []  # contexts whose track sequence is enqueued but not collected

    def finish(cx):
        state["tracked"] = cx.camshift_sequence_collect(nf, CALLS)  # the track objects of the 60th call

    def step(i=0):
        cx = ctxs[i % depth]
        cx.detect_enqueue(a.flags)
        if len(pending) == depth - 1 and pending:  # the other context's tracking result, while this detect runs
            finish(pending.pop(0))
        hits, counts = cx.detect_collect(cap=1 << 17)
        best = cx.best_faces(hits, counts, 1)  # facetrackr.js:147-175 for the whole batch
        fl = np.floor(np.stack([best["x"], best["y"], best["width"], best["height"]], axis=1)).astype(np.int64)
        rects = [tuple(fl[f]) if best["neighbors"][f] > 0 else (W // 4, H // 4, W // 2, H // 2)  # facetrackr.js:101
                 for f in range(nf)]
        cx.camshift_init(rects)
        cx.camshift_track_sequence(seq_ptrs, nf, calc_angles=True, fetch="none")  # 60 calls, one host call
        pending.append(cx)
        if depth == 1:
            finish(pending.pop(0))
        state.update(best=best, rects=rects)

    def drain():
        while pending:
            finish(pending.pop(0))

    for i in range(max(warmup, 1)):
        step(i)
    drain()

    def block(k):
        for i in range(k):
            step(i)
        drain()

    for cx in ctxs:
        cx.kernel_times(reset=True)  # the per-form launch counters restart with the timed region
    dts = env.timed_rounds(block, steps, a.rounds)
    dt, spread = round_stats(dts, steps)
    total_frames = world * nf * steps * (CALLS + 1)  # every processed frame: 1 detected + 60 tracked per stream
    if rank != 0:
        for cx in ctxs:
            cx.close()
        return None
    # Which form of k_cs_track_fused did the timed region launch (the library takes the 512-thread form, two workgroups per CU, when
    # another context of the device has work in flight at launch time)?  The library counts the launches per form (pseudo-timers of
    # ht_kernel_times, reset before the timed region).
    forms = {}
    for cx in ctxs:
        for k, v in cx.kernel_times(reset=True).items():
            if k.startswith("cs_fused_launches_"):
                forms[k[len("cs_fused_launches_"):]] = forms.get(k[len("cs_fused_launches_"):], 0) + int(v["launches"])
    form = 512 if forms.get("512", 0) > forms.get("1024", 0) else 1024
    for cx in ctxs[1:]:
        cx.close()
    # camshift roofline: HIP-event timing of the track kernel — THAT form, alone on the chip (a context of its own with the form
    # forced: the launch's own duration; side by side two launches of the small form get more calls done than this says) — and
    # the window pixels actually visited
    ctx.close()
    ctx = Context(device=local, options=",".join(x for x in (a.options, f"cs_fused_nt={form}") if x))
    ctx.set_geometry(W, H, nf)
    ctx.bind_device(dev_vers[0].data_ptr(), nf, W * H * 4)
    ctx.camshift_reserve(nf)
    ctx.camshift_init(state["rects"])
    ctx.camshift_track_sequence(seq_ptrs, nf, calc_angles=True)  # warm
    ctx.camshift_stats(nf, reset=True)
    ctx.profile(True)
    ctx.kernel_times(reset=True)
    ctx.camshift_init(state["rects"])
    ctx.camshift_track_sequence(seq_ptrs, nf, calc_angles=True)
    kt = {}
    for k, v in ctx.kernel_times(reset=True).items():  # both forms of the fused kernel count as cs_track: summed, not renamed over each other
        k = "cs_track" if k == "cs_track_512" else k
        if k in kt:
            kt[k] = dict(ms=kt[k]["ms"] + v["ms"], launches=kt[k]["launches"] + v["launches"])
        else:
            kt[k] = dict(v)
    ctx.profile(False)
    px, calls = ctx.camshift_stats(nf, reset=True)
    win_px_per_call = float(px.sum()) / max(float(calls.sum()), 1.0)
    # one full-frame histogram pass + the window passes, per stream and call
    b_track = 4 * W * H + 4 * win_px_per_call
    # >= 192 streams: ONE kernel per call (k_cs_track_fused: histogram + LUT + mean-shift), and a launch carries up to
    # 64 calls of every stream: times below are per CALL
    launches = {k: v["launches"] for k, v in kt.items() if k in ("cs_hist", "cs_lut", "cs_meanshift", "cs_track")}
    per_launch = {k: v["ms"] / CALLS for k, v in kt.items() if k in launches}
    call_ms = sum(per_launch.values())
    own = {"cs_hist": 4 * W * H * nf, "cs_meanshift": 4 * win_px_per_call * nf, "cs_track": b_track * nf}
    croof = dominant_roofline(
        per_launch, {k: launches[k] / CALLS for k in per_launch}, b_track * nf,
        dict(algorithmic_bytes_per_stream_call=round(b_track, 1), window_pixels_per_call=round(win_px_per_call, 1),
             streams_per_launch=nf, per="track() call of all streams (a launch carries up to 64 calls of every stream)"))
    # parity in the same run: the first PAR streams' 60 calls against the oracle (the checker), after the timed region
    PAR = 8
    from oracle import ht_oracle as ho

    ctx.camshift_init(state["rects"])
    got = ctx.camshift_track_sequence(seq_ptrs, nf, calc_angles=True, fetch="all")
    exact = tot = 0
    for f in range(PAR):
        o = ho.Camshift(True)
        o.init_tracker(vers[0, f], state["rects"][f])
        for k in range(CALLS):
            sw, to = o.track(vers[(k + 1) % NV, f])
            tot += 1
            exact += track_exact(got[k, f], sw, to)
    rec = {
        "value": round(total_frames / dt, 2), "unit": "frames/s", "steps": steps, "warmup": warmup, **spread,
        "scaling": "weak", "parity_exact": f"{exact}/{tot}",
        "parity_note": f"track() calls of the first {PAR} streams of this run vs oracle/ht_oracle.c: search window, x, "
                       "y, width, height bit-exact, angle to 1e-6 rad (all 256 x 60 calls: tests/test_gpu_shapes.py)",
        "config": {"workload": WORKLOAD_TEXT["c3"], "streams_per_gpu": nf, "track_calls_per_step": CALLS, "width": W,
                   "height": H, "steps_in_flight": depth,
                   "frame_mix": "family F only: one vote-image face per stream, moved by a seeded <= 3 px walk over 4 "
                                "frame versions",
                   "host_calls_per_step": "ht_detect_enqueue/collect + ht_best_faces + ht_camshift_init_batch + ONE "
                                          "ht_camshift_track_sequence (60 calls) + ht_camshift_sequence_collect"},
        "roofline": croof,
        "kernel_ms_per_track_call": {k: round(v, 5) for k, v in per_launch.items()},
        "kernel_rooflines": {k: dict(own_bytes_per_call=round(own[k]),
                                     gbs=round(own[k] / (per_launch[k] * 1e-3) / 1e9, 1),
                                     frac=round(own[k] / (per_launch[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
                             for k in per_launch},
        "track_path_hbm_frac": round(b_track * nf / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        # the same bytes against the wall clock of the timed region (detect steps included): what several launches side by side achieve
        "track_wall_hbm_frac": round(b_track * nf * CALLS * steps / dt / 1e9 / HBM_PEAK_GBS, 5),
        "fused_kernel_form": {"roofline_measured_on_threads": form, "launches_in_timed_region": forms,
                              "note": "k_cs_track_fused<SEQ, 512> = two workgroups per CU, taken per launch when another context of "
                                      "the device has work in flight; roofline = the majority form's launch alone on the chip"},
        "track_calls_per_s_device": round(nf / (call_ms * 1e-3), 1),
        "detected": int((state["best"]["neighbors"] > 0).sum()), "alive": int((state["tracked"]["width"] > 0).sum()),
    }
    if world == 1 and cpu_seconds > 0:
        f = int(np.argmax(state["best"]["neighbors"] > 0))
        cpu = cpu_camshift_baseline(vers[:, f], state["rects"][f], W, H, cpu_seconds)
        rec["cpu_baseline"] = cpu
        rec["vs_cpu_track_calls"] = round(rec["track_calls_per_s_device"] / cpu["value"], 1)
    else:
        rec["cpu_baseline"] = None
    ctx.close()
    return rec
