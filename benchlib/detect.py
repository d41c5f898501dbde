"""Detect workloads C2 / C4 (BASELINE.json configs[1], configs[3]): gray -> pyramid -> cascade scan -> grouped best face
per frame, `depth` batches in flight; plus the depth-1, PCIe-inclusive and single-frame variants of the same step."""
import time

import numpy as np

from .baseline import cpu_detect_baseline
from .common import (GEOM, HBM_PEAK_GBS, TRAFFIC_SOURCE, VALU_PEAK_WAVE_INSTS_PER_S, WORKLOAD_TEXT, dominant_roofline,
                     load_pmc, round_stats)


def arena_estimate(W, H):
    """bytes of RGBA + pyramid per frame, for the batches-in-flight choice (P ~ 5.7 B per source pixel)"""
    return 4 * W * H + 440000 * (W * H) // 76800


def detect_bench(env, a, name, steps, warmup, scaling="weak", frames_per_gpu=0, cpu_seconds=0.0, prewarm=0.0,
                 full=True, gather=None, unique=0, extras=True):
    """One detect workload: K timed steps (barrier + synchronize on both sides, max over ranks), then — on rank 0 —
    the live HIP-event roofline of the dominant kernel and the CPU baseline.  extras: the depth-1 and PCIe-inclusive
    variants of the same step (bounded, after the timed region).  Returns the record (rank 0) or None."""
    torch, dist, rank, world, local = env.torch, env.dist, env.rank, env.world, env.local
    from headtrackr_amd import distributed as hd
    from headtrackr_amd import native, synth
    from headtrackr_amd.api import Context

    W, H, nf_default = GEOM[name]
    if scaling == "strong":  # fixed total = 8 x the per-GPU default (C4: the 1024 frames of configs[3]), block-sharded
        total = 8 * nf_default
        f0, f1 = hd.shard_range(total, rank, world)
        nf = f1 - f0
    else:
        nf = frames_per_gpu or nf_default
        total = nf * world
        f0 = rank * nf
    nf_max = -(-total // world)
    # C4: every frame of the per-GPU batch is distinct; the 1024-frame strong-scaling batch repeats the 128
    uniq = min(unique or a.unique or (128 if name == "c4" else 256), nf)
    # frame g of the job is synthetic frame g mod uniq of the N/S/F mix, seeded per rank
    base = synth.mixed_batch(uniq, W, H, seed0=1234 + 1000 * rank)
    dev_uniq = torch.from_numpy(base).cuda()
    idx = torch.arange(nf, device="cuda") % uniq
    dev = dev_uniq[idx].contiguous() if nf != uniq else dev_uniq  # resident in HBM before the timed region
    del dev_uniq
    # batches in flight: 3 at 320x240, 2 at 1280x720 and for 1024-frame batches (LABLOG.md round 4 has the sweeps: the
    # deep kernel on 192 workgroups leaves a third batch something to overlap with; a third 700 MB arena buys nothing)
    depth = a.pipeline if a.pipeline > 0 else (3 if nf * arena_estimate(W, H) <= 256 * 1024 * 1024 else 2)
    ctxs = []
    for _ in range(depth):
        cx = Context(device=local, options=a.options or None)
        cx.set_geometry(W, H, nf)
        cx.bind_device(dev.data_ptr(), nf, W * H * 4)
        ctxs.append(cx)
    ctx = ctxs[0]
    # the exchange step's buffers, one set per batch in flight: pinned host records -> device records -> gathered
    # table.  Nothing in it blocks the host: the copy is asynchronous, the collective is enqueued on RCCL's stream, and
    # a set is only reused `depth` steps later (its event is checked first — by then it has long completed).
    gather_on = world > 1 or bool(gather)  # gather=True runs the exchange step on one GPU too (sub.gather_n1)
    xch = {}
    if gather_on:
        for cx in ctxs:
            xch[id(cx)] = dict(pin=torch.zeros((nf_max, hd.RECORD_F64), dtype=torch.float64).pin_memory(),
                               dev=torch.zeros((nf_max, hd.RECORD_F64), dtype=torch.float64, device="cuda"),
                               out=torch.zeros((world, nf_max, hd.RECORD_F64), dtype=torch.float64, device="cuda"),
                               ev=torch.cuda.Event())
    state = {}
    best_bufs = {id(cx): np.zeros(nf, dtype=native.RECT_DTYPE) for cx in ctxs}

    def finish(cx, requeue=False):
        # raw hits -> sorted -> seq rects -> ccv's grouping -> facetrackr's best face per frame: all inside the timed
        # step (one C-ABI call).  requeue: the context's next batch is enqueued inside that call, right after the raw
        # hits reached the host and before they are sorted and grouped — `depth` batches stay in flight meanwhile
        if requeue:
            best, nhits = cx.detect_collect_best_requeue(1, best_bufs[id(cx)], a.flags)
        else:
            best, nhits = cx.detect_collect_best(1, best_bufs[id(cx)])
        state["nhits"], state["best"] = nhits, best
        if gather_on:  # the path's one exchange step: every rank ends up with every frame's best-face rectangle
            x = xch[id(cx)]
            x["ev"].synchronize()  # the previous use of this set (depth steps ago) has been copied to the device
            rec = hd.pack_best_records(best, f0, nf_max)
            x["pin"].numpy()[:] = rec
            x["dev"].copy_(x["pin"], non_blocking=True)
            x["ev"].record()
            state["gathered"] = hd.allgather_records(x["dev"], world, nf_max, out=x["out"],
                                                     force_collective=bool(gather))
            state["rec"] = rec
        return best

    def run_steps(k):
        # k batches in all: the first min(depth, k) are enqueued up front, every collected batch re-enqueues its
        # context while batches remain to be started, the last ones are only collected
        started = min(depth, k)
        for i in range(started):
            ctxs[i].detect_enqueue(a.flags)
        for i in range(k):
            more = started < k
            finish(ctxs[i % depth], requeue=more and not a.no_requeue)
            if more:
                if a.no_requeue:
                    ctxs[i % depth].detect_enqueue(a.flags)
                started += 1

    # ~0.2 s of the same work before the W warm-up steps so that clocks, allocator and page tables are in their steady
    # state whatever W the caller chose (a 3-step warm-up is 1 ms of GPU time; a cold first run was up to 10 % slower)
    if prewarm > 0:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < prewarm:
            run_steps(8 * depth)
    run_steps(max(warmup, depth))
    dts = env.timed_rounds(run_steps, steps, a.rounds)
    dt, spread = round_stats(dts, steps)
    fps = total * steps / dt
    rank_ms = env.gather_scalar(float(np.median(env.last_own)) / steps * 1e3)  # every rank's own median block

    gather_ok = None
    if gather_on:  # outside the timed region: the gathered tensor must be the concatenation of every rank's records
        mine = state["rec"]
        everyone = [None] * world
        if world > 1:
            dist.all_gather_object(everyone, mine)
        else:
            everyone = [mine]
        if rank == 0:
            got = state["gathered"].cpu().numpy()
            gather_ok = all(np.array_equal(got[r], everyone[r]) for r in range(world))
            if not gather_ok:
                raise SystemExit("all-gather mismatch: gathered best-face records differ from the per-rank results")
    if rank != 0:
        for cx in ctxs:
            cx.close()
        return None

    # ---- roofline of the dominant kernel: live HIP-event timing on the ctx stream ---------------------------------
    ctx.profile(True)
    ctx.kernel_times(reset=True)
    psteps = max(3, min(10, steps))
    for _ in range(psteps):
        ctx.detect_enqueue(a.flags)
        ctx.detect_collect(cap=1 << 17)
    kt = ctx.kernel_times(reset=True)
    ctx.profile(False)
    per_step = {k: v["ms"] / psteps for k, v in kt.items()}
    P = ctx.pyramid_bytes_per_frame
    b_detect = 4 * W * H + 2 * P  # read RGBA once, write each gray plane once, read it once in the scan
    roofline = dominant_roofline(per_step, {k: v["launches"] / psteps for k, v in kt.items()}, b_detect * nf,
                                 dict(algorithmic_bytes_per_frame=b_detect, frames_per_step=nf))
    # PMC figures per step come from the committed rocprofv3 counter passes — only for the shape they were taken at
    traffic, valu, stale = load_pmc(name) if (nf, scaling) == (nf_default, "weak") else ({}, None, [])
    # the counters are constants of the build they were measured on: if the library being timed is another build they stay in the
    # record as what they are (`traffic_stale`) and the figure derived from them for THIS run (valu_issue) is dropped
    traffic_stale = bool(traffic) and (stale is None or len(stale) > 0)
    roofline["traffic"] = traffic.get(roofline["kernel"])
    roofline["traffic_source"] = TRAFFIC_SOURCE if roofline["traffic"] else None
    if traffic_stale:
        roofline["traffic_stale"] = True
        roofline["traffic_source"] = TRAFFIC_SOURCE + " — STALE: measured on another build of " + (", ".join(stale) if stale else "the library (no fingerprint recorded)")
        valu = None
    for k in roofline["co_dominant"]:
        roofline["co_dominant"][k]["traffic"] = traffic.get(k)
    dev_ms = sum(per_step.values())
    wall_ms = dt / steps * 1e3
    par = (f"frames block-sharded over {world} GPU(s), all-gather of {nf_max}x64B best-face rect records (verified "
           "against the per-rank results)") if world > 1 else "1 GPU"
    rec = {
        "value": round(fps, 2), "unit": "frames/s", "steps": steps, "warmup": warmup, **spread, "scaling": scaling,
        "config": {"workload": WORKLOAD_TEXT[name], "frames_per_gpu": nf, "frames_total": total,
                   "batches_in_flight": depth, "width": W, "height": H, "unique_frames": uniq,
                   "frame_mix": "1/3 LCG noise, 1/3 smooth, 1/3 faces", "parallelism": par},
        "roofline": roofline,
        "rank_ms_per_step_min": round(min(rank_ms), 4), "rank_ms_per_step_max": round(max(rank_ms), 4),
    }
    if gather_ok is not None:
        rec["allgather_verified"] = bool(gather_ok)
    # each kernel against its OWN algorithmic bytes (per step): gray 5*W*H, pyramid build 2*(P - W*H) (every derived
    # plane written once, its source read once), tile scan P (every plane read once)
    own = {"gray": 5 * W * H * nf, "resample": 2 * (P - W * H) * nf, "scan_tiles": P * nf}
    rec["kernel_ms_per_step"] = {k: round(v, 5) for k, v in per_step.items()}
    rec["kernel_rooflines"] = {
        k: dict(own_bytes_per_step=own[k], gbs=round(own[k] / (per_step[k] * 1e-3) / 1e9, 1),
                frac=round(own[k] / (per_step[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), traffic=traffic.get(k))
        for k in own if k in per_step}
    if traffic:
        rec["path_traffic_over_algorithmic"] = round(sum(v for v in traffic.values() if v) / (b_detect * nf), 3)
        rec["traffic_stale"] = traffic_stale
    rec["device_ms_per_step"] = round(dev_ms, 5)
    # whole-path figures (every kernel of a step): device time, and the wall clock of the timed region
    rec["path_hbm_gbs"] = round(b_detect * nf / (dev_ms * 1e-3) / 1e9, 2)
    rec["path_hbm_frac"] = round(b_detect * nf / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    rec["wall_hbm_frac"] = round(b_detect * total / world / (dt / steps) / 1e9 / HBM_PEAK_GBS, 5)
    if valu:  # the path's real bound: VALU wave instructions of a step against the chip's issue rate
        rec["valu_issue"] = dict(
            wave_insts_per_step=valu, peak_wave_insts_per_s=VALU_PEAK_WAVE_INSTS_PER_S,
            frac_wall=round(valu / (wall_ms * 1e-3) / VALU_PEAK_WAVE_INSTS_PER_S, 4),
            frac_device=round(valu / (dev_ms * 1e-3) / VALU_PEAK_WAVE_INSTS_PER_S, 4),
            source="SQ_INSTS_VALU per step from " + TRAFFIC_SOURCE + "; peak = 1024 SIMDs x 2.4 GHz / 4")
    rec["hits_per_step"] = int(state["nhits"])
    rec["faces_per_step"] = int((state["best"]["neighbors"] > 0).sum())
    if full:
        ctx.detect_enqueue(a.flags | 16)  # one extra untimed pass with HT_SCAN_STATS for the survival curve
        ctx.detect_collect(cap=1 << 17)
        sc = ctx.stage_counts()
        per_stage = [int(v) for v in ctx.cascade.stages["count"]]
        feat_evals = sum(int(sc[j]) * per_stage[j] for j in range(len(per_stage)))
        rec.update(feature_evals_per_s_device=round(feat_evals / (dev_ms * 1e-3), 1),
                   windows_per_frame=int(ctx.windows_per_frame),
                   windows_per_s_device=round(float(sc[0]) / (dev_ms * 1e-3), 1),  # over the summed device time of the profiling pass
                   windows_per_s=round(float(sc[0]) / (wall_ms * 1e-3), 1),  # over the timed region's wall clock
                   stage_in=[int(v) for v in sc])
    if extras:
        rec["depth1"] = depth1_block(ctx, finish, a.flags, nf, wall_ms)
        rec["pcie_inclusive"] = pcie_block(env, ctx, base, idx.cpu().numpy(), nf, W, H, finish, a.flags)
        ctx.bind_device(dev.data_ptr(), nf, W * H * 4)
    if world == 1 and cpu_seconds > 0:
        frames = base[np.arange(min(nf, 64)) % uniq]
        cpu, port = cpu_detect_baseline(frames, W, H, ctx.cascade.blob, cpu_seconds)
        rec["cpu_baseline"], rec["cpu_baseline_port"] = cpu, port
        rec["vs_cpu"] = round(fps / cpu["value"], 1)
    else:
        rec["cpu_baseline"] = None
        if world > 1:
            rec["cpu_baseline_note"] = "measured at N = 1 only (rank 0's host cores are shared by N ranks here)"
    for cx in ctxs:
        cx.close()
    return rec


def depth1_block(ctx, finish, flags, nf, pipelined_ms, target_s=0.15):
    """The same step strictly in turn on ONE context (enqueue, wait, post-process, then the next batch): what a
    depth-1 caller — the drop-in facetrackr.Tracker, a live feed — pays, with nothing to hide the under-filled
    launches behind.  Median of 5 blocks."""
    def block(k):
        for _ in range(k):
            ctx.detect_enqueue(flags)
            finish(ctx)

    block(3)
    t0 = time.perf_counter()
    block(3)
    k = int(max(3, min(200, target_s / 5 / max((time.perf_counter() - t0) / 3, 1e-6))))
    dts = []
    for _ in range(5):
        ctx.synchronize()
        t0 = time.perf_counter()
        block(k)
        dts.append((time.perf_counter() - t0) / k * 1e3)
    med = float(np.median(dts))
    return dict(ms_per_step=round(med, 4), ms_per_step_min=round(min(dts), 4), ms_per_step_max=round(max(dts), 4),
                steps_per_block=k, value=round(nf / med * 1e3, 1), unit="frames/s",
                vs_pipelined=round(med / pipelined_ms, 3),
                what="one context, ht_detect_enqueue + ht_detect_collect_best strictly in turn, frames resident in HBM")


def pcie_block(env, ctx, base, idx, nf, W, H, finish, flags, target_s=0.3):
    """Every batch crosses PCIe inside the step: pinned host frames -> ht_upload_frames_async (copy stream, back
    buffer) while the previous batch is processed -> ht_swap_frames.  SURVEY.md §8(d): "end-to-end incl. pinned-host
    H2D separately" — link-bound by construction (320x240: 0.31 MB, 1280x720: 3.69 MB per frame)."""
    torch = env.torch
    fbytes = W * H * 4
    sets = 2
    host = torch.empty((sets, nf, H, W, 4), dtype=torch.uint8).pin_memory()
    hv = host.numpy()
    for s in range(sets):
        hv[s] = base[idx] if s == 0 else base[idx][::-1]  # two different batches take turns
    sbytes = nf * fbytes

    def block(k):
        ctx.upload_async_ptr(host.data_ptr(), nf)
        ctx.swap_frames()
        for i in range(k):
            ctx.upload_async_ptr(host.data_ptr() + ((i + 1) % sets) * sbytes, nf)
            ctx.detect_enqueue(flags)
            finish(ctx)
            ctx.swap_frames()
        ctx.synchronize()

    block(3)
    t0 = time.perf_counter()
    block(3)
    k = int(max(3, min(100, target_s / 3 / max((time.perf_counter() - t0) / 3, 1e-6))))
    dts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        block(k)
        torch.cuda.synchronize()
        dts.append((time.perf_counter() - t0) / k * 1e3)
    med = float(np.median(dts))
    fps = nf / med * 1e3
    del host
    return dict(value=round(fps, 1), unit="frames/s", ms_per_step=round(med, 4), ms_per_step_min=round(min(dts), 4),
                ms_per_step_max=round(max(dts), 4), steps_per_block=k, h2d_gbs=round(fps * fbytes / 1e9, 2),
                mb_per_frame=round(fbytes / 1e6, 3),
                what="pinned host frames, double-buffered ht_upload_frames_async + ht_swap_frames, one context: the "
                     "H2D copy of batch i+1 overlaps the kernels of batch i")


def single_frame_latency(env, W, H, flags=0, iters=200):
    """One frame, one context, resident in HBM: ht_detect_enqueue + ht_detect_collect_best in turn (graph replay from
    the third call on) — the detect latency of the drop-in single-frame path without PCIe."""
    from headtrackr_amd import synth
    from headtrackr_amd.api import Context

    fr = synth.face_frame(W, H, [(W // 3, H // 4, min(W, H) // 3)])
    dev = env.torch.from_numpy(fr[None].copy()).cuda()
    ctx = Context(device=env.local)
    ctx.set_geometry(W, H, 1)
    ctx.bind_device(dev.data_ptr(), 1, W * H * 4)
    out = None
    for _ in range(10):
        ctx.detect_enqueue(flags)
        out = ctx.detect_collect_best(1)
    lat = []
    for _ in range(iters):
        t0 = time.perf_counter()
        ctx.detect_enqueue(flags)
        out = ctx.detect_collect_best(1)
        lat.append((time.perf_counter() - t0) * 1e3)
    ctx.profile(True)
    ctx.kernel_times(reset=True)
    for _ in range(5):
        ctx.detect_enqueue(flags)
        ctx.detect_collect_best(1)
    kt = ctx.kernel_times(reset=True)
    ctx.profile(False)
    ctx.close()
    return dict(p50_ms=round(float(np.percentile(lat, 50)), 4), p99_ms=round(float(np.percentile(lat, 99)), 4),
                min_ms=round(min(lat), 4), device_ms=round(sum(v["ms"] for v in kt.values()) / 5, 4),
                faces=int(out[0]["neighbors"][0] > 0), iters=iters,
                what=f"one {W}x{H} frame resident in HBM: enqueue + collect_best in turn, wall clock per call")
