#!/usr/bin/env python3
"""bench.py — frames/s of the detect(+camshift) hot path on N MI355X, one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic frames that are already resident in HBM:
gray -> 39-level pyramid -> full BBF cascade scan -> raw hits copied back and sorted (ht_detect_enqueue +
ht_detect_collect), plus, for N > 1, one RCCL all-gather of the fixed-size per-frame best-face records.
Workloads (BASELINE.json configs): c2 = 256 x 320x240 detect on every GPU (default; the configuration the metric is
quoted on), c4 = 128 x 1280x720 per GPU detect (1024 frames on 8 GPUs), c3 = c2's frames, detect once + 60 camshift
track() calls per step.  Weak scaling: per-GPU work is fixed, value = all ranks' frames / max-over-ranks time.

Extra objects on the line: "roofline" (dominant kernel, HBM bound, live HIP-event timing through the C ABI's
profiling scopes on the same stream) and "cpu_baseline" (the CPU oracle port, 1 core, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 1000 for c2, 300 for c4 / c5, 20 for c3: ~0.3-0.5 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed warm-up steps (default: a tenth of --steps, at least 3)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (default: 256 for c2/c3, 128 for c4)")
    ap.add_argument("--unique", type=int, default=0, help="distinct synthetic frames generated (tiled to --frames)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--flags", type=int, default=0, help="ht_detect flags (A/B of scan schedules)")
    ap.add_argument("--pipeline", type=int, default=2, help="batches in flight (contexts on their own HIP streams); 1 = enqueue+collect strictly in turn")
    ap.add_argument("--prewarm", type=float, default=0.2, help="seconds of untimed steady-state work before the warm-up steps (0 for profiler runs)")
    a = ap.parse_args()
    if a.steps <= 0:
        a.steps = {"c2": 1000, "c4": 300, "c3": 20, "c5": 300}[a.workload]
    if a.warmup < 0:
        a.warmup = max(3, a.steps // 10) if a.workload != "c5" else 1
    return a


def stream_bench(a, torch, dist, rank, world, local):
    """C5 (BASELINE.json configs[4]): one live 1920x1080 feed per GPU; every frame travels host -> GPU (pinned buffer,
    PCIe) -> result on the host.  Frame 0, 30, 60, ... : full-cascade detect + camshift.initTracker on the best face
    (facetrackr.js:97-108); every other frame: camshift.track.  A step is one frame of every feed; reports aggregate
    frames/s and the per-frame end-to-end latency distribution."""
    from headtrackr_amd import synth
    from headtrackr_amd.api import Context

    W, H = 1920, 1080
    nuniq = 30
    host = torch.empty((nuniq, H, W, 4), dtype=torch.uint8).pin_memory()
    hv = host.numpy()
    for k in range(nuniq):  # a face drifting 3 px / frame over a flat background
        hv[k] = synth.face_frame(W, H, [(700 + 3 * k + 40 * rank, 300 + k, 360)])
    ctx = Context(device=local)
    ctx.set_geometry(W, H, 1)
    ctx.camshift_reserve(1)
    fbytes = W * H * 4
    lat = {"detect": [], "track": []}

    def process(i):
        if i % 30 == 0:
            ctx.detect_enqueue(0)
            hits, counts = ctx.detect_collect(cap=1 << 14)
            best = ctx.best_faces(hits, counts, 1)[0]
            if best["neighbors"] > 0 and best["confidence"] > -10:
                ctx.camshift_init([[int(np.floor(best["x"])), int(np.floor(best["y"])), int(np.floor(best["width"])), int(np.floor(best["height"]))]])
            return best
        return ctx.camshift_track(1, calc_angles=True)[0]

    def frame(i):  # strictly in turn: upload, process, result — the latency of one frame on an idle pipeline
        t0 = time.perf_counter()
        ctx.upload_ptr(host.data_ptr() + (i % nuniq) * fbytes, 1)
        out = process(i)
        lat["detect" if i % 30 == 0 else "track"].append((time.perf_counter() - t0) * 1e3)
        return out

    for i in range(max(a.warmup, 1) * 30 + 1):
        frame(i)
    steps = a.steps
    lat = {"detect": [], "track": []}
    for i in range(steps):  # latency pass (not the timed region)
        frame(i)
    # timed region: the same frames with double-buffered ingest (ht_upload_frames_async / ht_swap_frames): frame i+1
    # crosses PCIe on the copy stream while frame i is processed
    ctx.upload_async_ptr(host.data_ptr(), 1)
    ctx.swap_frames()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        ctx.upload_async_ptr(host.data_ptr() + ((i + 1) % nuniq) * fbytes, 1)
        last = process(i)
        ctx.swap_frames()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        allv = np.array(lat["detect"] + lat["track"])
        pct = lambda v, q: round(float(np.percentile(np.array(v), q)), 4) if len(v) else None  # noqa: E731
        print(json.dumps({
            "metric": "frames/sec streaming 1920x1080 feeds (detect every 30th frame, camshift between), end to end incl. PCIe (double-buffered ingest)",
            "value": round(world * steps / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": a.warmup,
            "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "C5: one 1920x1080 RGBA feed per GPU, host->GPU every frame, detect on frames 0,30,60,... + camshift.track otherwise",
                       "feeds_per_gpu": 1, "width": W, "height": H, "parallelism": f"{world} feed(s), one per GPU, no collective"},
            "latency_note": "latency_ms: upload + process + result of one frame strictly in turn (separate untimed pass over the same frames)",
            "latency_ms": {"p50": pct(allv, 50), "p99": pct(allv, 99), "detect_p50": pct(lat["detect"], 50), "detect_max": pct(lat["detect"], 100),
                           "track_p50": pct(lat["track"], 50), "track_p99": pct(lat["track"], 99)},
            "last_track": [float(last["x"]), float(last["y"]), float(last["width"]), float(last["height"])],
            "roofline": None, "cpu_baseline": None}), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from headtrackr_amd import distributed as hd
    from headtrackr_amd import synth
    from headtrackr_amd.api import Context

    if a.workload == "c5":
        return stream_bench(a, torch, dist, rank, world, local)

    if a.workload == "c4":
        W, H, nf = 1280, 720, a.frames or 128
        uniq = a.unique or 12
    else:
        W, H, nf = 320, 240, a.frames or 256
        uniq = a.unique or nf
    uniq = min(uniq, nf)
    # frame i of rank r is synthetic frame (r*nf + i) mod uniq of the N/S/F mix
    if a.workload == "c3":
        # C3: family F only — every stream has one face; 4 versions of each stream's frame with the face
        # moved by a seeded <= 3 px walk; track() call i sees version i % 4
        NV = 4
        walk = synth.lcg_stream(4242 + rank, 2 * NV * nf).astype(np.int64) >> 20
        vers = np.empty((NV, nf, H, W, 4), dtype=np.uint8)
        for f in range(nf):
            s0 = 48 + (f * 7) % 80
            x, y = 20 + (f * 13) % (W - s0 - 40), 16 + (f * 29) % (H - s0 - 32)
            for v in range(NV):
                vers[v, f] = synth.face_frame(W, H, [(x, y, s0)])
                x += int(walk[2 * (f * NV + v)] % 7) - 3
                y += int(walk[2 * (f * NV + v) + 1] % 7) - 3
        frames = vers[0]
        dev_vers = [torch.from_numpy(vers[v]).cuda() for v in range(NV)]
        dev = dev_vers[0]
    else:
        base = synth.mixed_batch(uniq, W, H, seed0=1234 + 1000 * rank)
        frames = base[np.arange(nf) % uniq]
        dev = torch.from_numpy(frames).cuda()  # resident in HBM before the timed region
    # every step is one full pass over the batch with its results collected on the host; with --pipeline 2 the next
    # step is enqueued (on a second context / HIP stream) before the previous one is collected, so the host-side
    # collect + sort of step i overlaps the GPU work of step i+1.  All K steps are collected inside the timed region.
    depth = 1 if a.workload == "c3" else max(1, a.pipeline)
    ctxs = []
    for _ in range(depth):
        cx = Context(device=local)
        cx.set_geometry(W, H, nf)
        cx.bind_device(dev.data_ptr(), nf, W * H * 4)
        ctxs.append(cx)
    ctx = ctxs[0]
    if a.workload == "c3":
        ctx.camshift_reserve(nf)

    rec_local = torch.zeros((nf, hd.RECORD_F64), dtype=torch.float64, device="cuda")

    def finish(cx):
        hits, counts = cx.detect_collect(cap=1 << 16)
        if world > 1:  # the path's one exchange step: every rank ends up with every frame's best-face record
            rec_local.copy_(torch.from_numpy(hd.pack_records(hits, counts, nf)), non_blocking=False)
            hd.allgather_records(rec_local, world, nf)
        return hits, counts

    def run_steps(k):
        inflight = []
        last = None
        for i in range(k):
            cx = ctxs[i % depth]
            if len(inflight) == depth:
                last = finish(inflight.pop(0))
            cx.detect_enqueue(a.flags)
            inflight.append(cx)
        while inflight:
            last = finish(inflight.pop(0))
        return last

    c3_state = {}

    def step():
        ctx.detect_enqueue(a.flags)
        hits, counts = ctx.detect_collect(cap=1 << 16)
        if a.workload == "c3":
            # detect once, initTracker on the best face, then 60 camshift track() calls on the moving frames
            best = ctx.best_faces(hits, counts, 1)  # facetrackr.js:147-175 for the whole batch
            fl = np.floor(np.stack([best["x"], best["y"], best["width"], best["height"]], axis=1)).astype(np.int64)  # facetrackr.js:101-106
            rects = [tuple(fl[f]) if best["neighbors"][f] > 0 else (W // 4, H // 4, W // 2, H // 2) for f in range(nf)]
            ctx.camshift_init(rects)
            for it in range(60):
                ctx.bind_device(dev_vers[(it + 1) % NV].data_ptr(), nf, W * H * 4)
                tracked = ctx.camshift_track(nf, calc_angles=True, fetch=(it == 59))
            ctx.bind_device(dev.data_ptr(), nf, W * H * 4)
            c3_state["detected"] = int((best["neighbors"] > 0).sum())
            c3_state["alive"] = int((tracked["width"] > 0).sum())
        if world > 1:  # the path's one exchange step: every rank ends up with every frame's best-face record
            rec_local.copy_(torch.from_numpy(hd.pack_records(hits, counts, nf)), non_blocking=False)
            hd.allgather_records(rec_local, world, nf)
        return hits, counts

    # setup, before the W warm-up steps: ~0.2 s of the same work so that clocks, the allocator and the page tables are in
    # their steady state whatever W the caller chose (a 3-step warm-up is 1 ms of GPU time; a cold first run measured
    # up to 10 % slower)
    if a.workload != "c3" and a.prewarm > 0:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < a.prewarm:
            run_steps(8 * depth)
    if a.workload == "c3":
        for _ in range(a.warmup):
            hits, counts = step()
    else:
        hits, counts = run_steps(max(a.warmup, depth))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    if a.workload == "c3":
        for _ in range(a.steps):
            hits, counts = step()
    else:
        hits, counts = run_steps(a.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # C3 counts every processed frame: 1 detected + 60 tracked per stream and step
    total_frames = world * nf * a.steps * (61 if a.workload == "c3" else 1)
    fps = total_frames / dt

    # ---- roofline of the dominant kernel: live HIP-event timing on the ctx stream (rank 0) ----------------------
    roofline = None
    extra = {}
    if rank == 0:
        ctx.profile(True)
        ctx.kernel_times(reset=True)
        psteps = max(3, min(10, a.steps))
        for _ in range(psteps):
            ctx.detect_enqueue(a.flags)
            ctx.detect_collect(cap=1 << 16)
        kt = ctx.kernel_times(reset=True)
        ctx.profile(False)
        per_step = {k: v["ms"] / psteps for k, v in kt.items()}
        per_launch = {k: v["ms"] / v["launches"] for k, v in kt.items()}
        # "dominant kernel" = the longest single launch (the unit the roofline formula is written in).  k_resample runs 7
        # dependent launches per step, each over a different slice of the pyramid; its total per step is listed in
        # kernel_ms_per_step and its own-bytes roofline in kernel_rooflines.
        dom = max(per_launch, key=per_launch.get)
        P = ctx.pyramid_bytes_per_frame
        b_detect = 4 * W * H + 2 * P  # read RGBA once, write each gray plane once, read it once in the scan
        launches_per_step = kt[dom]["launches"] / psteps
        avg_launch_ms = per_launch[dom]
        achieved = b_detect * nf / launches_per_step / (avg_launch_ms * 1e-3) / 1e9
        traffic = None
        all_traffic = {}
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                all_traffic = json.load(open(tf)).get(a.workload, {})
                traffic = all_traffic.get(dom)
            except Exception:
                traffic = None
        roofline = dict(bound="hbm", kernel=dom, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic,
                        algorithmic_bytes_per_frame=b_detect, avg_launch_ms=round(avg_launch_ms, 5))
        # each kernel against its OWN algorithmic bytes (per step): gray 5*W*H, pyramid build 2*(P - W*H) (every derived plane
        # written once, its source read once), tile scan P (every plane read once)
        own = {"gray": 5 * W * H * nf, "resample": 2 * (P - W * H) * nf, "scan_tiles": P * nf}
        kernel_rooflines = {k: dict(own_bytes_per_step=own[k], gbs=round(own[k] / (per_step[k] * 1e-3) / 1e9, 1),
                                    frac=round(own[k] / (per_step[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    hbm_traffic_per_launch=all_traffic.get(k)) for k in own if k in per_step}
        dev_ms = sum(per_step.values())
        ctx.detect_enqueue(a.flags | 16)  # one extra untimed pass with HT_SCAN_STATS for the survival curve
        ctx.detect_collect(cap=1 << 16)
        sc = ctx.stage_counts()
        # a device-copy ceiling measured in the same run (what a kernel that only reads and writes HBM
        # reaches on this box), and feature evaluations per second from the survival curve
        buf = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
        dst = torch.empty_like(buf)
        dst.copy_(buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(8):
            dst.copy_(buf)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2.0 * buf.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del buf, dst
        roofline["device_copy_gbs"] = round(copy_gbs, 1)
        roofline["frac_of_device_copy"] = round(achieved / copy_gbs, 5)
        per_stage = [int(v) for v in ctx.cascade.stages["count"]]
        feat_evals = sum(int(sc[j]) * per_stage[j] for j in range(len(per_stage)))
        extra = dict(feature_evals_per_s=round(feat_evals / (dev_ms * 1e-3), 1),
                     kernel_ms_per_step={k: round(v, 5) for k, v in per_step.items()}, kernel_rooflines=kernel_rooflines,
                     device_ms_per_step=round(dev_ms, 5),
                     path_hbm_gbs=round(b_detect * nf / (dev_ms * 1e-3) / 1e9, 2),
                     path_hbm_frac=round(b_detect * nf / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     windows_per_frame=int(ctx.windows_per_frame),
                     windows_per_s=round(float(sc[0]) / (dev_ms * 1e-3), 1),
                     hits_per_step=int(len(hits)), stage_in=[int(v) for v in sc])

    # ---- CPU baselines on the box's host cores (rank 0, N = 1 only), bounded samples of the same frames ------------------
    #  * "reference": the UNMODIFIED reference JS, single-threaded Node (its own execution model), from oracle/_ref
    #  * "port":      the plain-C oracle restatement, 1 thread
    cpu = None
    cpu_port = None
    if rank == 0 and world == 1 and a.cpu_seconds > 0:
        import shutil
        import subprocess
        import tempfile

        from oracle import ht_oracle as ho

        blob = ctx.cascade.blob
        ho.detect_raw(frames[0], blob)  # warm
        t0 = time.perf_counter()
        done = 0
        while done < nf and (done < 4 or time.perf_counter() - t0 < a.cpu_seconds):
            ho.detect_raw(frames[done], blob)
            done += 1
        cdt = time.perf_counter() - t0
        cpu_port = dict(value=round(done / cdt, 3), unit="frames/s", cores=1, kind="port",
                        sample=f"first {done} of the {nf} {W}x{H} frames of this workload, oracle/ht_oracle.c detect (gray+pyramid+scan), 1 thread",
                        host_cpus=os.cpu_count())
        cpu = cpu_port
        gz = os.path.join(ROOT, "oracle", "_ref", "headtrackr_ref.js.gz")
        node = shutil.which("node")
        if node and os.path.exists(gz):
            try:
                ns = min(nf, 64)
                with tempfile.NamedTemporaryFile(suffix=".raw") as tf:
                    frames[:ns].tofile(tf.name)
                    r = subprocess.run([node, os.path.join(ROOT, "oracle", "ref_bench.js"), tf.name, str(ns), str(W), str(H), str(a.cpu_seconds)],
                                       capture_output=True, text=True, timeout=a.cpu_seconds * 6 + 120)
                j = json.loads(r.stdout.strip().splitlines()[-1])
                cpu = dict(value=round(j["fps"], 3), unit="frames/s", cores=1, kind="reference",
                           sample=f"first {j['frames']} of the {nf} {W}x{H} frames of this workload: unmodified reference JS (ccv.grayscale + ccv.detect_objects(..., 5, 1)) "
                                  f"on oracle/canvas_shim.js, {j['node']} single thread, median {j['ms_median']:.1f} ms/frame, {100 * j['shim_fraction']:.0f}% of it inside the canvas shim",
                           host_cpus=j["cpus"], cpu_model=j["cpu_model"])
            except Exception as e:  # the port baseline stands in
                cpu = dict(cpu_port, note=f"reference JS baseline unavailable: {e}")

    if rank == 0:
        line = {
            "metric": f"frames/sec {'(1 full-cascade detect + 60 camshift track per stream)' if a.workload == 'c3' else 'full-cascade detect'} at {W}x{H}",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": {"c2": "C2: 256 x 320x240 RGBA frames per GPU, full BBF cascade detect (interval 5), raw hits to host",
                                    "c3": "C3: 256 streams of 320x240 per GPU (one moving face each): detect once, initTracker, then 60 camshift track() calls; every processed frame counts",
                                    "c4": "C4: 1280x720 frames, 128 per GPU (1024 on 8 GPUs), full cascade detect + all-gather of best-face records"}[a.workload],
                       "frames_per_gpu": nf, "batches_in_flight": depth, "width": W, "height": H, "unique_frames": uniq, "frame_mix": "1/3 LCG noise, 1/3 smooth, 1/3 faces",
                       "parallelism": f"frames sharded over {world} GPU(s), all-gather of {nf}x64B records" if world > 1 else "1 GPU"},
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_port": cpu_port,
        }
        line.update(extra)
        if c3_state:
            line["c3"] = c3_state
        print(json.dumps(line), flush=True)
    for cx in ctxs:
        cx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
