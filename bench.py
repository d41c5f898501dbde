#!/usr/bin/env python3
"""bench.py — frames/s of the detect(+camshift) hot path on N MI355X, one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--scaling weak|strong] [--feeds K] [--no-sub]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (torch.distributed.run, one
rank per GPU over RCCL) and fails loudly when fewer than N GPUs are visible; under a launcher WORLD_SIZE must equal --gpus.

A "step" is one pass of the hot path over one batch of synthetic frames that are already resident in HBM:
gray -> 39-level pyramid -> full BBF cascade scan -> raw hits copied back, sorted, converted to rects, grouped and reduced
to the best face per frame (ht_detect_enqueue + ht_detect_collect + ht_best_faces = ccv.detect_objects(..., 5, 1) +
facetrackr's selection, /root/reference/src/ccv.js:109-333, facetrackr.js:147-175), plus, for N > 1, one RCCL all-gather of
the fixed-size per-frame best-face rectangles.

Workloads (BASELINE.json configs):
  c2  256 x 320x240 detect per GPU — the headline `value` (the configuration the metric is quoted on);
  c4  1280x720 detect, 128 frames per GPU (weak) or 1024 frames in total (--scaling strong: 1024 / N per GPU);
  c3  256 streams of 320x240: detect once + initTracker + 60 camshift track() calls per step (ht_camshift_track_sequence);
  c5  --feeds K live 1920x1080 feeds per GPU (one batch of K frames per time step), detect every 30th frame; resident and PCIe-inclusive.
The default run (c2) also measures c4 (weak + strong), c3 and c5 (one feed: latency; 8 feeds: the N = 1 point of configs[4]) with
their own bounded budgets and reports them as sub-records of the same JSON line ("sub": {"c4_1gpu", "c4_strong", "c3", "c5",
"js_host"}), each with its own `roofline` and, at N = 1, the unmodified reference JS timed on the host cores as `cpu_baseline`.
--no-sub skips them (profiler runs).

Timing: the K timed steps (barrier + synchronize on both sides, max over ranks) are repeated R times back to back ("rounds"; R is
chosen so that the timed work is ~0.4 s, K stays what the caller passed) and the MEDIAN block is reported, with the spread
(`ms_per_step_min` / `ms_per_step_max`): a 20-step block of C2 is 6 ms of GPU time, far too short to quote on its own.
"dominant kernel" of a roofline = the kernel with the largest device time per step (the sum over its launches).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
DEFAULT_STEPS = {"c2": 1000, "c4": 300, "c3": 20, "c5": 300}
SUB_STEPS = {"c4": 100, "c4_strong": 24, "c3": 12, "c5": 90}
GEOM = {"c2": (320, 240, 256), "c3": (320, 240, 256), "c4": (1280, 720, 128)}
WORKLOAD_TEXT = {
    "c2": "C2: 256 x 320x240 RGBA frames per GPU, full BBF cascade detect (interval 5) incl. grouping + best face per frame on the host",
    "c3": "C3: 256 streams of 320x240 per GPU (one moving face each): detect once, initTracker, then 60 camshift track() calls; every processed frame counts",
    "c4": "C4: 1280x720 frames, full cascade detect incl. grouping + best face per frame, all-gather of best-face rects for N > 1",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 1000 for c2, 300 for c4 / c5, 20 for c3: ~0.3-0.5 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed warm-up steps (default: a tenth of --steps, at least 3)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="strong: total frames fixed at 8 x the per-GPU default (c4: 1024) and split over the ranks")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (default: 256 for c2/c3, 128 for c4)")
    ap.add_argument("--unique", type=int, default=0, help="distinct synthetic frames generated (tiled to --frames)")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="budget of each cpu_baseline leg (0 = skip)")
    ap.add_argument("--flags", type=int, default=0, help="ht_detect flags (A/B of scan schedules)")
    ap.add_argument("--pipeline", type=int, default=0, help="batches in flight (contexts on their own HIP streams); 1 = enqueue+collect strictly in turn; "
                    "0 = auto: 3 at 320x240, 2 at 1280x720 (measured, see detect_bench)")
    ap.add_argument("--prewarm", type=float, default=0.2, help="seconds of untimed steady-state work before the warm-up steps (0 for profiler runs)")
    ap.add_argument("--no-sub", action="store_true", help="only the primary workload (no c4 / c3 sub-records)")
    ap.add_argument("--no-requeue", action="store_true", help="A/B: enqueue a context's next batch only after its results were post-processed")
    ap.add_argument("--feeds", type=int, default=1, help="c5: live feeds per GPU (own contexts / HIP streams); 8 on one GPU is the N = 1 point of BASELINE.json configs[4]")
    ap.add_argument("--rounds", type=int, default=0, help="repetitions of the K-step timed block (median reported); 0 = auto: ~0.4 s of timed work, 3..25 rounds")
    a = ap.parse_args()
    if a.steps <= 0:
        a.steps = DEFAULT_STEPS[a.workload]
    if a.warmup < 0:
        a.warmup = max(3, a.steps // 10) if a.workload != "c5" else 1
    return a


class Env:
    def __init__(self, torch, dist, rank, world, local, stub=False):
        self.torch, self.dist, self.rank, self.world, self.local, self.stub = torch, dist, rank, world, local, stub
        self.dev = "cpu" if stub else "cuda"

    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        if not self.stub:
            self.torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        if self.world > 1:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed_rounds(self, run_block, steps, rounds=0, target_s=0.4):
        """R back-to-back timed blocks of `steps` steps each, every block bracketed by barrier + synchronize on both sides and
        max-reduced over the ranks.  rounds == 0: R from the first block so that the timed work is ~target_s (3..25; every rank
        derives the same R from the same max-reduced time).  Returns the list of block times in seconds."""
        dts = []
        r = 0
        while True:
            self.fence()
            t0 = time.perf_counter()
            run_block(steps)
            self.fence()
            dts.append(self.max_over_ranks(time.perf_counter() - t0))
            r += 1
            if rounds <= 0:
                rounds = int(min(25, max(3, -(-target_s // max(dts[0], 1e-6)))))
            if r >= rounds:
                return dts


def round_stats(dts, steps):
    """median block -> the reported time; min / max -> the spread"""
    med = float(np.median(dts))
    return med, dict(rounds=len(dts), ms_per_step=round(med / steps * 1e3, 4), ms_per_step_min=round(min(dts) / steps * 1e3, 4),
                     ms_per_step_max=round(max(dts) / steps * 1e3, 4))


def dominant_roofline(per_step_ms, launches_per_step, bytes_per_step, extra=None):
    """SURVEY.md §8(d) roofline of the DOMINANT kernel = the one with the largest device time per step (sum of its launches):
    achieved = algorithmic bytes of a step / that kernel's time per step (for a kernel with one launch per step this is bytes per
    launch / average launch duration)."""
    dom = max(per_step_ms, key=per_step_ms.get)
    ach = bytes_per_step / (per_step_ms[dom] * 1e-3) / 1e9
    # kernels within 5 % of the dominant one's time per step are named with it (C2: resample's launches and scan_tiles trade places from run to run)
    co = {k: dict(kernel_ms_per_step=round(v, 5), frac=round(bytes_per_step / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 5))
          for k, v in per_step_ms.items() if k != dom and v >= 0.95 * per_step_ms[dom]}
    r = dict(bound="hbm", kernel=dom, co_dominant=co, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5), traffic=None,
             kernel_ms_per_step=round(per_step_ms[dom], 5), launches_per_step=round(launches_per_step[dom], 2),
             avg_launch_ms=round(per_step_ms[dom] / max(launches_per_step[dom], 1e-9), 5), dominant="largest device time per step (sum of its launches)")
    if extra:
        r.update(extra)
    return r


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only): bounded samples of the same frames on the box's host cores
#  * "reference": the UNMODIFIED reference JS, single-threaded Node (its own execution model), from oracle/_ref
#  * "port":      the plain-C oracle restatement, 1 thread


def cpu_detect_baseline(frames, W, H, blob, seconds):
    from oracle import ht_oracle as ho

    nf = len(frames)
    ho.detect_raw(frames[0], blob)  # warm
    t0 = time.perf_counter()
    done = 0
    while done < nf and (done < 2 or time.perf_counter() - t0 < seconds / 2):
        ho.detect_raw(frames[done], blob)
        done += 1
    cdt = time.perf_counter() - t0
    port = dict(value=round(done / cdt, 3), unit="frames/s", cores=1, kind="port",
                sample=f"first {done} of the {nf} {W}x{H} frames of this workload, oracle/ht_oracle.c detect (gray+pyramid+scan), 1 thread",
                host_cpus=os.cpu_count())
    cpu = port
    gz = os.path.join(ROOT, "oracle", "_ref", "headtrackr_ref.js.gz")
    node = shutil.which("node")
    if node and os.path.exists(gz):
        try:
            ns = min(nf, 64)
            with tempfile.NamedTemporaryFile(suffix=".raw") as tf:
                np.ascontiguousarray(frames[:ns]).tofile(tf.name)
                r = subprocess.run([node, os.path.join(ROOT, "oracle", "ref_bench.js"), tf.name, str(ns), str(W), str(H), str(seconds)],
                                   capture_output=True, text=True, timeout=seconds * 6 + 120)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            cpu = dict(value=round(j["fps"], 3), unit="frames/s", cores=1, kind="reference",
                       sample=f"first {j['frames']} of the {nf} {W}x{H} frames of this workload: unmodified reference JS (ccv.grayscale + ccv.detect_objects(..., 5, 1)) "
                              f"on oracle/canvas_shim.js, {j['node']} single thread, median {j['ms_median']:.1f} ms/frame, {100 * j['shim_fraction']:.0f}% of it inside the canvas shim",
                       host_cpus=j["cpus"], cpu_model=j["cpu_model"])
        except Exception as e:  # the port baseline stands in
            cpu = dict(port, note=f"reference JS baseline unavailable: {e}")
    return cpu, port


def cpu_camshift_baseline(versions, rect, W, H, seconds):
    """camshift.Tracker.initTracker + track() (camshift.js:198-312) of ONE stream on its moving frames: the unmodified
    reference JS (kind "reference") or, without Node / the bundle, the C port."""
    gz = os.path.join(ROOT, "oracle", "_ref", "headtrackr_ref.js.gz")
    node = shutil.which("node")
    nv = len(versions)
    if node and os.path.exists(gz):
        try:
            with tempfile.NamedTemporaryFile(suffix=".raw") as tf:
                np.ascontiguousarray(versions).tofile(tf.name)
                r = subprocess.run([node, os.path.join(ROOT, "oracle", "ref_bench.js"), tf.name, str(nv), str(W), str(H), str(seconds), "camshift"] + [str(int(v)) for v in rect],
                                   capture_output=True, text=True, timeout=seconds * 6 + 120)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            return dict(value=round(j["fps"], 3), unit="track() calls/s", cores=1, kind="reference",
                        sample=f"{j['calls']} camshift.Tracker.track() calls of one {W}x{H} stream (initTracker on rect {list(map(int, rect))}, its {nv} moving frames in turn): unmodified reference JS on "
                               f"oracle/canvas_shim.js, {j['node']} single thread, median {j['ms_median']:.2f} ms/call",
                        host_cpus=j["cpus"], cpu_model=j["cpu_model"])
        except Exception as e:
            note = f"reference JS baseline unavailable: {e}"
    else:
        note = "node or oracle/_ref missing"
    from oracle import ht_oracle as ho

    st = ho.cs_init(versions[0], *[int(v) for v in rect], calc_angles=True)
    t0 = time.perf_counter()
    calls = 0
    while calls < 8 or time.perf_counter() - t0 < seconds / 2:
        ho.cs_track(st, versions[(calls + 1) % nv])
        calls += 1
    return dict(value=round(calls / (time.perf_counter() - t0), 3), unit="track() calls/s", cores=1, kind="port",
                sample=f"{calls} track() calls of one {W}x{H} stream, oracle/ht_oracle.c, 1 thread", host_cpus=os.cpu_count(), note=note)


def device_copy_ceiling(torch):
    """what a kernel that only reads and writes HBM reaches on this box, measured in the same run."""
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
    gbs = 2.0 * buf.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del buf, dst
    return gbs


def load_traffic(workload):
    """profiles/traffic.json: HBM bytes per STEP and kernel timer name, summed over the launches of every kernel the timer covers
    (`resample` = the k_resample launches + k_resample_tail), from the committed rocprofv3 counter passes (tools/collect_profiles.py)."""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(tf)).get(workload, {}).get("per_step", {})
    except Exception:
        return {}


# ---------------------------------------------------------------------------------------------------------------------
# detect workloads (c2, c4)


def detect_bench(env, a, name, steps, warmup, scaling="weak", frames_per_gpu=0, cpu_seconds=0.0, prewarm=0.0, full=True, gather=None, unique=0):
    """One detect workload: K timed steps (barrier + synchronize on both sides, max over ranks), then — on rank 0 — the live
    HIP-event roofline of the dominant kernel and the CPU baseline.  Returns the record (rank 0) or None."""
    torch, dist, rank, world, local = env.torch, env.dist, env.rank, env.world, env.local
    from headtrackr_amd import distributed as hd
    from headtrackr_amd import native, synth
    from headtrackr_amd.api import Context

    W, H, nf_default = GEOM[name]
    if scaling == "strong":  # fixed total = 8 x the per-GPU default (C4: the 1024 frames of BASELINE.json configs[3]), block-sharded
        total = 8 * nf_default
        f0, f1 = hd.shard_range(total, rank, world)
        nf = f1 - f0
    else:
        nf = frames_per_gpu or nf_default
        total = nf * world
        f0 = rank * nf
    nf_max = -(-total // world)
    uniq = min(unique or a.unique or (128 if name == "c4" else 256), nf)  # C4: every frame of the per-GPU batch is distinct (round 2 tiled 12 unique frames); the 1024-frame strong-scaling batch repeats the 128
    # frame g of the job is synthetic frame g mod uniq' of the N/S/F mix, seeded per rank
    base = synth.mixed_batch(uniq, W, H, seed0=1234 + 1000 * rank)
    dev_uniq = torch.from_numpy(base).cuda()
    idx = torch.arange(nf, device="cuda") % uniq
    dev = dev_uniq[idx].contiguous() if nf != uniq else dev_uniq  # resident in HBM before the timed region
    del dev_uniq
    # batches in flight: 3 at 320x240, 2 at 1280x720.  Round 3 measured 2 = 3 at C2 with the deep kernel on 512 workgroups (each holding 77 KB of
    # LDS: it shut the other batches out of every CU).  With that kernel on 192 workgroups a third batch has something to overlap with: C2
    # 0.2449 / 0.2317 / 0.2598 ms per step at 2 / 3 / 4 in flight (1000-step blocks), 0.2489 / 0.2400 at 2 / 3 in the driver's 20-step blocks;
    # C4 1.113 / 1.112 at 2 / 3 (a third 700 MB arena buys nothing there).  LABLOG.md round 4.
    # c2_large (1024 frames per batch, 760 MB per context) is back at 2: 1 152 k frames/s at 2, 1 111 k at 3.
    depth = a.pipeline if a.pipeline > 0 else (3 if nf * (4 * W * H + 440000 * (W * H) // 76800) <= 256 * 1024 * 1024 else 2)
    ctxs = []
    for _ in range(depth):
        cx = Context(device=local)
        cx.set_geometry(W, H, nf)
        cx.bind_device(dev.data_ptr(), nf, W * H * 4)
        ctxs.append(cx)
    ctx = ctxs[0]
    # the exchange step's buffers, one set per batch in flight: pinned host records -> device records -> gathered table.  Nothing in it
    # blocks the host: the copy is asynchronous, the collective is enqueued on RCCL's stream, and a set is only reused `depth` steps later
    # (its event is checked first — by then it has long completed).
    gather_on = world > 1 or bool(gather)  # gather=True runs the exchange step on one GPU too (sub.gather_n1: what a step pays for it)
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
        # raw hits -> sorted -> seq rects -> ccv's grouping -> facetrackr's best face per frame: all inside the timed step
        # (one C-ABI call, ht_detect_collect_best: the Python host was the limiter of a 0.3 ms step with three calls and copies).
        # requeue: the context's next batch is enqueued inside that call, right after the raw hits reached the host and before they
        # are sorted and grouped — `depth` batches stay in flight while the host post-processes
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
            state["gathered"] = hd.allgather_records(x["dev"], world, nf_max, out=x["out"], force_collective=bool(gather))
            state["rec"] = rec
        return best

    def run_steps(k):
        # k batches in all: the first min(depth, k) are enqueued up front, every collected batch re-enqueues its context while
        # batches remain to be started, the last ones are only collected
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
    # state whatever W the caller chose (a 3-step warm-up is 1 ms of GPU time; a cold first run measured up to 10 % slower)
    if prewarm > 0:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < prewarm:
            run_steps(8 * depth)
    run_steps(max(warmup, depth))
    dts = env.timed_rounds(run_steps, steps, a.rounds)
    dt, spread = round_stats(dts, steps)
    fps = total * steps / dt

    gather_ok = None
    if gather_on:  # outside the timed region: the gathered tensor must be the concatenation of every rank's own records
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

    # ---- roofline of the dominant kernel: live HIP-event timing on the ctx stream --------------------------------------
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
    all_traffic = load_traffic(name)
    roofline = dominant_roofline(per_step, {k: v["launches"] / psteps for k, v in kt.items()}, b_detect * nf,
                                 dict(algorithmic_bytes_per_frame=b_detect, frames_per_step=nf))
    # PMC traffic per step (every launch's own counters summed), from the committed rocprofv3 counter passes — only for the shape they were taken at
    if (nf, scaling) != (nf_default, "weak"):
        all_traffic = {}
    roofline["traffic"] = all_traffic.get(roofline["kernel"])
    for k in roofline["co_dominant"]:
        roofline["co_dominant"][k]["traffic"] = all_traffic.get(k)
    dev_ms = sum(per_step.values())
    rec = {
        "value": round(fps, 2), "unit": "frames/s", "steps": steps, "warmup": warmup, **spread, "scaling": scaling,
        "config": {"workload": WORKLOAD_TEXT[name], "frames_per_gpu": nf, "frames_total": total, "batches_in_flight": depth, "width": W, "height": H,
                   "unique_frames": uniq, "frame_mix": "1/3 LCG noise, 1/3 smooth, 1/3 faces",
                   "parallelism": f"frames block-sharded over {world} GPU(s), all-gather of {nf_max}x64B best-face rect records (verified against the per-rank results)" if world > 1 else "1 GPU"},
        "roofline": roofline,
    }
    if gather_ok is not None:
        rec["allgather_verified"] = bool(gather_ok)
    # each kernel against its OWN algorithmic bytes (per step): gray 5*W*H, pyramid build 2*(P - W*H) (every derived plane
    # written once, its source read once), tile scan P (every plane read once)
    own = {"gray": 5 * W * H * nf, "resample": 2 * (P - W * H) * nf, "scan_tiles": P * nf}
    rec["kernel_ms_per_step"] = {k: round(v, 5) for k, v in per_step.items()}
    rec["kernel_rooflines"] = {k: dict(own_bytes_per_step=own[k], gbs=round(own[k] / (per_step[k] * 1e-3) / 1e9, 1),
                                       frac=round(own[k] / (per_step[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), traffic=all_traffic.get(k)) for k in own if k in per_step}
    if all_traffic:
        rec["path_traffic_over_algorithmic"] = round(sum(v for v in all_traffic.values() if v) / (b_detect * nf), 3)
    rec["device_ms_per_step"] = round(dev_ms, 5)
    # whole-path figures (every kernel of a step): device time, and the wall clock of the timed region
    rec["path_hbm_gbs"] = round(b_detect * nf / (dev_ms * 1e-3) / 1e9, 2)
    rec["path_hbm_frac"] = round(b_detect * nf / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    rec["wall_hbm_frac"] = round(b_detect * total / world / (dt / steps) / 1e9 / HBM_PEAK_GBS, 5)
    rec["hits_per_step"] = int(state["nhits"])
    rec["faces_per_step"] = int((state["best"]["neighbors"] > 0).sum())
    if full:
        ctx.detect_enqueue(a.flags | 16)  # one extra untimed pass with HT_SCAN_STATS for the survival curve
        ctx.detect_collect(cap=1 << 17)
        sc = ctx.stage_counts()
        per_stage = [int(v) for v in ctx.cascade.stages["count"]]
        feat_evals = sum(int(sc[j]) * per_stage[j] for j in range(len(per_stage)))
        rec.update(feature_evals_per_s=round(feat_evals / (dev_ms * 1e-3), 1), windows_per_frame=int(ctx.windows_per_frame),
                   windows_per_s=round(float(sc[0]) / (dev_ms * 1e-3), 1), stage_in=[int(v) for v in sc])
    if world == 1 and cpu_seconds > 0:
        frames = base[np.arange(min(nf, 64)) % uniq]
        cpu, port = cpu_detect_baseline(frames, W, H, ctx.cascade.blob, cpu_seconds)
        rec["cpu_baseline"], rec["cpu_baseline_port"] = cpu, port
        rec["vs_cpu"] = round(fps / cpu["value"], 1)
    else:
        rec["cpu_baseline"] = None
    for cx in ctxs:
        cx.close()
    return rec


# ---------------------------------------------------------------------------------------------------------------------
# C3: detect once + 60 camshift track() calls per stream


def c3_bench(env, a, steps, warmup, cpu_seconds=0.0):
    torch, rank, world, local = env.torch, env.rank, env.world, env.local
    from headtrackr_amd import synth
    from headtrackr_amd.api import Context

    W, H, nf = GEOM["c3"]
    nf = a.frames or nf
    # family F only: every stream has one face; NV versions of each stream's frame with the face moved by a
    # seeded <= 3 px walk; track() call i sees version (i + 1) % NV
    NV, CALLS = 4, 60
    walk = synth.lcg_stream(4242 + rank, 2 * NV * nf).astype(np.int64) >> 20
    vers = np.empty((NV, nf, H, W, 4), dtype=np.uint8)
    for f in range(nf):
        s0 = 48 + (f * 7) % 80
        x, y = 20 + (f * 13) % (W - s0 - 40), 16 + (f * 29) % (H - s0 - 32)
        for v in range(NV):
            vers[v, f] = synth.face_frame(W, H, [(x, y, s0)])
            x += int(walk[2 * (f * NV + v)] % 7) - 3
            y += int(walk[2 * (f * NV + v) + 1] % 7) - 3
    dev_vers = [torch.from_numpy(vers[v]).cuda() for v in range(NV)]
    # Three contexts take the steps in turn (own HIP streams, own tracker states): while one batch of streams is in its 60 track()
    # calls (one launch, one workgroup per stream: half of every CU idle) the next steps' detects run on the other contexts.
    # --pipeline 1 keeps the steps strictly in turn.
    depth = a.pipeline if a.pipeline > 0 else 3  # steps in flight, measured (round 4): 6.23 / 6.82 / 6.41 M frames/s at 2 / 3 / 4
    ctxs = []
    for _ in range(depth):
        cx = Context(device=local)
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
        fl = np.floor(np.stack([best["x"], best["y"], best["width"], best["height"]], axis=1)).astype(np.int64)  # facetrackr.js:101-106
        rects = [tuple(fl[f]) if best["neighbors"][f] > 0 else (W // 4, H // 4, W // 2, H // 2) for f in range(nf)]
        cx.camshift_init(rects)
        cx.camshift_track_sequence(seq_ptrs, nf, calc_angles=True, fetch="none")  # 60 calls, one host call, enqueue only
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

    dts = env.timed_rounds(block, steps, a.rounds)
    dt, spread = round_stats(dts, steps)
    total_frames = world * nf * steps * (CALLS + 1)  # every processed frame: 1 detected + 60 tracked per stream and step
    for cx in ctxs[1:]:
        cx.close()
    if rank != 0:
        ctx.close()
        return None
    # camshift roofline: HIP-event timing of the two track kernels + the window pixels actually visited
    ctx.camshift_stats(nf, reset=True)
    ctx.profile(True)
    ctx.kernel_times(reset=True)
    ctx.camshift_init(state["rects"])
    ctx.camshift_track_sequence(seq_ptrs, nf, calc_angles=True)
    kt = ctx.kernel_times(reset=True)
    ctx.profile(False)
    px, calls = ctx.camshift_stats(nf, reset=True)
    win_px_per_call = float(px.sum()) / max(float(calls.sum()), 1.0)
    b_track = 4 * W * H + 4 * win_px_per_call  # one full-frame histogram pass + the window passes, per stream and call
    # >= 192 streams: ONE kernel per call (k_cs_track_fused: histogram + LUT + mean-shift); fewer: k_cs_hist + k_cs_meanshift
    # (with >= 192 streams ht_camshift_track_sequence puts up to 64 calls of every stream into one launch: times are per CALL below)
    launches = {k: v["launches"] for k, v in kt.items() if k in ("cs_hist", "cs_lut", "cs_meanshift", "cs_track")}
    per_launch = {k: v["ms"] / CALLS for k, v in kt.items() if k in launches}  # device time per track() CALL of the 256 streams
    call_ms = sum(per_launch.values())
    own = {"cs_hist": 4 * W * H * nf, "cs_meanshift": 4 * win_px_per_call * nf, "cs_track": b_track * nf}
    croof = dominant_roofline(per_launch, {k: launches[k] / CALLS for k in per_launch}, b_track * nf,
                              dict(algorithmic_bytes_per_stream_call=round(b_track, 1), window_pixels_per_call=round(win_px_per_call, 1), streams_per_launch=nf,
                                   per="track() call of all streams (a launch carries up to 64 calls of every stream)"))
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
            g = got[k, f]
            tot += 1
            exact += int([int(g["sw_x"]), int(g["sw_y"]), int(g["sw_width"]), int(g["sw_height"])] == list(sw) and
                         all(float(g[q]) == to[q] for q in ("x", "y", "width", "height")) and abs(float(g["angle"]) - to["angle"]) < 1e-6)
    rec = {
        "value": round(total_frames / dt, 2), "unit": "frames/s", "steps": steps, "warmup": warmup, **spread, "scaling": "weak",
        "parity_exact": f"{exact}/{tot}", "parity_note": f"track() calls of the first {PAR} streams of this run vs oracle/ht_oracle.c: search window, x, y, width, height bit-exact, angle to 1e-6 rad "
                                                       "(all 256 x 60 calls: tests/test_gpu_shapes.py, profiles/r03_camshift_parity.json)",
        "config": {"workload": WORKLOAD_TEXT["c3"], "streams_per_gpu": nf, "track_calls_per_step": CALLS, "width": W, "height": H,
                   "frame_mix": "family F only: one vote-image face per stream, moved by a seeded <= 3 px walk over 4 frame versions",
                   "steps_in_flight": depth,
                   "host_calls_per_step": "ht_detect_enqueue/collect + ht_best_faces + ht_camshift_init_batch + ONE ht_camshift_track_sequence (60 calls) + ht_camshift_sequence_collect"},
        "roofline": croof,
        "kernel_ms_per_track_call": {k: round(v, 5) for k, v in per_launch.items()},
        "kernel_rooflines": {k: dict(own_bytes_per_call=round(own[k]), gbs=round(own[k] / (per_launch[k] * 1e-3) / 1e9, 1),
                                     frac=round(own[k] / (per_launch[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)) for k in per_launch},
        "track_path_hbm_frac": round(b_track * nf / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
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


# ---------------------------------------------------------------------------------------------------------------------
# C5: streaming feeds


def stream_bench(env, a, feeds=1, steps=300, warm_cycles=1, cpu_seconds=0.0):
    """C5 (BASELINE.json configs[4]): `feeds` live 1920x1080 feeds per GPU.  The feeds of a GPU are frame-synchronous cameras: their
    frames of one time step form ONE batch of `feeds` frames on one context (one tracker stream per feed), so a step of K feeds costs
    the host the same handful of launches as a step of one feed.  Step 0, 30, 60, ...: full-cascade detect of every feed +
    camshift.initTracker on its best face (facetrackr.js:97-108); every other step: camshift.track (main.js:168-180 is this loop for
    one feed).  Two timed variants of the same steps:
      value            frames already resident in HBM when the timed region starts (the bench contract's definition): bind, process, result;
      pcie_inclusive   every frame travels host -> GPU in the step (pinned buffer, double-buffered: step i+1 crosses PCIe on the copy
                       stream while step i is processed) — bounded by the link: 8.3 MB per frame;
    plus the per-step end-to-end latency distribution, PCIe included, strictly in turn (upload, process, result).
    Returns the record on rank 0."""
    torch, rank, world, local = env.torch, env.rank, env.world, env.local
    from headtrackr_amd import synth
    from headtrackr_amd.api import Context

    W, H, K = 1920, 1080, feeds
    nuniq = 30
    fbytes = W * H * 4
    # time step k of feed f = a face drifting 3 px / frame over a flat background, feed f running 7 f frames ahead
    uniq = synth.stream_feed_frames(nuniq, W, H, rank)
    host = torch.empty((nuniq, K, H, W, 4), dtype=torch.uint8).pin_memory()
    hv = host.numpy()
    for k in range(nuniq):
        for f in range(K):
            hv[k, f] = uniq[synth.stream_frame_index(k, f, nuniq)]
    dev = host.cuda()  # the same steps resident in HBM (nuniq x K x 8.3 MB)
    ctx = Context(device=local)
    ctx.set_geometry(W, H, K)
    ctx.camshift_reserve(K)
    sbytes = K * fbytes

    def is_detect(i):
        return i % 30 == 0

    def enqueue(i):
        if is_detect(i):
            ctx.detect_enqueue(0)
        else:
            ctx.camshift_track(K, calc_angles=True, fetch=False)

    def collect(i):
        if is_detect(i):
            best = ctx.detect_collect_best(1)[0]
            fl = np.floor(np.stack([best["x"], best["y"], best["width"], best["height"]], axis=1)).astype(np.int64)  # facetrackr.js:101-106
            ctx.camshift_init([tuple(fl[f]) if best["neighbors"][f] > 0 and best["confidence"][f] > -10 else (W // 4, H // 4, W // 2, H // 2) for f in range(K)])
            return best
        return ctx.camshift_track_collect(K)

    lat = {"detect": [], "track": []}

    def step_in_turn(i):  # the latency of one time step on an idle pipeline: upload K frames, process, results on the host
        t0 = time.perf_counter()
        ctx.upload_ptr(host.data_ptr() + (i % nuniq) * sbytes, K)
        enqueue(i)
        collect(i)
        lat["detect" if is_detect(i) else "track"].append((time.perf_counter() - t0) * 1e3)

    for i in range(max(warm_cycles, 1) * 30 + 1):
        step_in_turn(i)
    lat = {"detect": [], "track": []}
    for i in range(steps):  # latency pass (not a timed region)
        step_in_turn(i)
    last = {}

    def run_resident(k, on_result):
        """inputs resident in HBM.  A track step is enqueued before the previous step's results are waited for (two outstanding;
        the library keeps enqueue-only results in a ring of pinned slots): a feed's frame i + 1 does not depend on the HOST having
        seen the result of frame i — the search window lives on the device.  A detect step drains the pipeline first: its best faces
        come back to the host, which floors them and calls initTracker (facetrackr.js:97-108)."""
        pend = []
        for i in range(k):
            ctx.bind_device(dev.data_ptr() + (i % nuniq) * sbytes, K)
            if is_detect(i):
                while pend:
                    j = pend.pop(0)
                    on_result(j, collect(j))
                enqueue(i)
                on_result(i, collect(i))
            else:
                enqueue(i)
                pend.append(i)
                if len(pend) > 1:
                    j = pend.pop(0)
                    on_result(j, collect(j))
        while pend:
            j = pend.pop(0)
            on_result(j, collect(j))

    def block_resident(k):
        run_resident(k, lambda i, got: last.__setitem__("r", got))

    def block_pcie(k):  # double-buffered ingest: ht_upload_frames_async / ht_swap_frames
        ctx.upload_async_ptr(host.data_ptr(), K)
        ctx.swap_frames()
        for i in range(k):
            ctx.upload_async_ptr(host.data_ptr() + ((i + 1) % nuniq) * sbytes, K)
            enqueue(i)
            last["p"] = collect(i)
            ctx.swap_frames()

    rounds = a.rounds if a.workload == "c5" else 3
    block_resident(31)
    dt, spread = round_stats(env.timed_rounds(block_resident, steps, rounds, target_s=0.5), steps)
    block_pcie(31)
    dt_p, spread_p = round_stats(env.timed_rounds(block_pcie, steps, rounds, target_s=0.5), steps)
    graph_launches = ctx.graph_launches
    if rank != 0:
        ctx.close()
        return None
    allv = np.array(lat["detect"] + lat["track"])
    pct = lambda v, q: round(float(np.percentile(np.array(v), q)), 4) if len(v) else None  # noqa: E731
    # rooflines of one 30-step cycle (1 detect + 29 track), live HIP events on the ctx stream; the dominant kernel of each path =
    # the one with the largest device time
    ctx.camshift_stats(K, reset=True)
    ctx.profile(True)
    ctx.kernel_times(reset=True)
    for i in range(30):
        ctx.bind_device(dev.data_ptr() + (i % nuniq) * sbytes, K)
        enqueue(i)
        collect(i)
    kt = ctx.kernel_times(reset=True)
    ctx.profile(False)
    px, calls = ctx.camshift_stats(K, reset=True)
    det_names, cs_names = ("gray", "resample", "scan_tiles", "scan_deep"), ("cs_hist", "cs_lut", "cs_meanshift", "cs_track")
    P = ctx.pyramid_bytes_per_frame
    b_detect = 4 * W * H + 2 * P
    win = float(px.sum()) / max(float(calls.sum()), 1.0)
    b_track = 4 * W * H + 4 * win
    ncs = max(int(calls[0]), 1)  # track() steps in the cycle
    roofline = dominant_roofline({k: kt[k]["ms"] for k in det_names if k in kt}, {k: kt[k]["launches"] for k in det_names if k in kt}, b_detect * K,
                                 dict(algorithmic_bytes_per_frame=b_detect, frames_per_step=K, per="detect step",
                                      note="a few 1080p frames per launch cannot fill 256 CUs x 6 workgroups: latency-, not bandwidth-bound by construction"))
    cs_roofline = dominant_roofline({k: kt[k]["ms"] / ncs for k in cs_names if k in kt}, {k: kt[k]["launches"] / ncs for k in cs_names if k in kt}, b_track * K,
                                    dict(algorithmic_bytes_per_stream_call=round(b_track, 1), window_pixels_per_call=round(win, 1), streams_per_launch=K, per="track() step of all feeds"))
    det_ms = sum(kt[k]["ms"] for k in det_names if k in kt)
    cs_ms = sum(kt[k]["ms"] for k in cs_names if k in kt) / ncs
    dev_ms = {k: round(v["ms"], 4) for k, v in kt.items()}
    cpu = None
    if world == 1 and cpu_seconds > 0:
        # the reference JS on one feed: detect on one 1080p frame, camshift.track on the following ones; a 30-frame cycle
        # = 1 detect + 29 track calls (facetrackr's state machine after the white-balance phase)
        fr = np.ascontiguousarray(uniq[:4])
        cd, _ = cpu_detect_baseline(fr[:2], W, H, ctx.cascade.blob, cpu_seconds * 0.6)
        bx = [int(np.floor(v)) for v in (700, 300, 360, 360)]
        ct = cpu_camshift_baseline(fr, bx, W, H, cpu_seconds * 0.4)
        cyc = 1.0 / cd["value"] + 29.0 / ct["value"]
        cpu = dict(value=round(30.0 / cyc, 3), unit="frames/s", cores=1, kind=cd["kind"] if cd["kind"] == ct["kind"] else "mixed",
                   sample=f"one feed, 30-frame cycle = 1 detect ({cd['value']} frames/s: {cd['sample']}) + 29 camshift track ({ct['value']} calls/s: {ct['sample']})",
                   host_cpus=cd.get("host_cpus"))
    fps = world * K * steps / dt
    fps_p = world * K * steps / dt_p
    lr = last["r"]
    # parity in the same run (after the timed regions): one 31-step cycle exactly as timed above — bind a new set every step, detect
    # (graph replay by now) + initTracker on step 0 / 30, enqueue-only track + collect otherwise — every feed against the oracle
    from oracle import ht_oracle as ho

    exact = tot = 0
    det_ok = det_tot = 0
    oracles = [None] * K
    replays0 = ctx.graph_launches
    results = {}
    run_resident(31, lambda i, got: results.__setitem__(i, np.array(got, copy=True)))
    for i in range(31):
        got = results[i]
        for f in range(K):
            fr = uniq[synth.stream_frame_index(i, f, nuniq)]
            if is_detect(i):
                w = ho.best_faces(fr[None], ctx.cascade.blob, 1)[0]
                det_tot += 1
                det_ok += int(all(got[k][f] == w[k] for k in ("x", "y", "width", "height", "confidence", "neighbors")))
                rect = [int(np.floor(got[k][f])) for k in ("x", "y", "width", "height")] if got["neighbors"][f] > 0 and got["confidence"][f] > -10 else [W // 4, H // 4, W // 2, H // 2]
                oracles[f] = ho.Camshift(True)
                oracles[f].init_tracker(fr, rect)
            else:
                sw, to = oracles[f].track(fr)
                g = got[f]
                tot += 1
                exact += int([int(g["sw_x"]), int(g["sw_y"]), int(g["sw_width"]), int(g["sw_height"])] == list(sw) and
                             all(float(g[q]) == to[q] for q in ("x", "y", "width", "height")) and abs(float(g["angle"]) - to["angle"]) < 1e-6)
    parity = dict(parity_exact=f"{exact}/{tot}", parity_detect_exact=f"{det_ok}/{det_tot}", parity_graph_replays=int(ctx.graph_launches - replays0),
                  parity_note="one 31-step cycle of this run's own loop (bind per step, graph-replayed detect + initTracker on steps 0 / 30, enqueue-only track steps two outstanding + collect) vs oracle/ht_oracle.c: "
                              "best faces bit-exact; track(): search window, x, y, width, height bit-exact, angle to 1e-6 rad (61 steps, also from Node: tests/test_gpu_c5.py)")
    rec = {
        "value": round(fps, 2), "unit": "frames/s", "steps": steps, "warmup": warm_cycles, **spread, "scaling": "weak",
        "config": {"workload": f"C5: {K} frame-synchronous 1920x1080 RGBA feed(s) per GPU as one batch of {K} frames per time step; detect + initTracker on steps 0, 30, 60, ..., camshift.track otherwise",
                   "feeds_per_gpu": K, "width": W, "height": H, "frames": "resident in HBM before the timed region (value); host -> GPU every step in pcie_inclusive",
                   "parallelism": f"{world * K} feed(s): {K} per GPU in one context / batch, {world} GPU(s), no collective"},
        "per_feed_fps": round(fps / (world * K), 2), **parity,
        "pcie_inclusive": {"value": round(fps_p, 2), "unit": "frames/s", **spread_p, "per_feed_fps": round(fps_p / (world * K), 2),
                           "h2d_gbs": round(fps_p / world * fbytes / 1e9, 2), "note": "double-buffered pinned ingest; 8.29 MB per frame: the link (~56 GB/s measured) allows ~6.8 k frames/s per GPU whatever the kernels do"},
        "latency_note": "latency_ms: one time step strictly in turn incl. PCIe: upload the feeds' frames, process, results on the host (separate untimed pass)",
        "latency_ms": {"p50": pct(allv, 50), "p99": pct(allv, 99), "detect_p50": pct(lat["detect"], 50), "detect_max": pct(lat["detect"], 100),
                       "track_p50": pct(lat["track"], 50), "track_p99": pct(lat["track"], 99), "samples": int(len(allv))},
        "detect_graph_replays": int(graph_launches),
        "last_track": [float(lr["x"][0]), float(lr["y"][0]), float(lr["width"][0]), float(lr["height"][0])],
        "roofline": roofline, "camshift_roofline": cs_roofline,
        "device_ms": {"detect_step": round(det_ms, 4), "track_step": round(cs_ms, 4), "per_30_step_cycle": dev_ms},
        "cpu_baseline": cpu, "vs_cpu": round(fps / cpu["value"], 1) if cpu else None}
    ctx.close()
    return rec


# ---------------------------------------------------------------------------------------------------------------------
# the JavaScript host (north_star: "Host code stays JavaScript (Node)"): the same C ABI driven from Node through the N-API addon


def js_host_bench(seconds=2.0):
    """tests/js/bench_host.js on this GPU: detect frames/s at the C2 shape from Node — ccv.detect_objects_batch on host frames (PCIe
    every call) and ccv.DeviceBatch (frames resident in HBM, enqueue / collect-best / re-enqueue over 2 contexts: this file's
    headline loop, driven from JavaScript) — and the per-call latency of the drop-in facetrackr.Tracker.track() at 320x240, next to
    the unmodified reference JS on the same frames.  None when node or the addon is missing."""
    from headtrackr_amd import synth

    node = shutil.which("node")
    script = os.path.join(ROOT, "tests", "js", "bench_host.js")
    if not node or not os.path.exists(script) or not os.path.exists(os.path.join(ROOT, "headtrackr_amd", "js", "headtrackr_hip.node")):
        return None
    W, H, n, nt = 320, 240, 256, 30
    try:
        with tempfile.TemporaryDirectory() as td:
            c2 = os.path.join(td, "c2.raw")
            synth.mixed_batch(n, W, H, seed0=1234).tofile(c2)
            tr = os.path.join(td, "track.raw")
            np.stack([synth.face_frame(W, H, [(90 + 2 * k, 50 + k, 96)]) for k in range(nt)]).tofile(tr)
            r = subprocess.run([node, script, str(seconds), c2, str(n), tr, str(nt)], capture_output=True, text=True, timeout=seconds * 20 + 240)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            # C5 from the JavaScript host: 8 frame-synchronous 1080p feeds, DeviceBatch.detectStep / trackStep (tests/js/c5_stream.js)
            try:
                uq = os.path.join(td, "uniq.raw")
                synth.stream_feed_frames(30, 1920, 1080, 0).tofile(uq)
                r5 = subprocess.run([node, os.path.join(ROOT, "tests", "js", "c5_stream.js"), "bench", uq, "30", "8", str(seconds)], capture_output=True, text=True, timeout=seconds * 20 + 240)
                j["c5"] = json.loads(r5.stdout.strip().splitlines()[-1])
            except Exception as e:
                j["c5"] = {"error": f"{type(e).__name__}: {e}"}
        j["config"] = {"workload": f"JS host (Node + N-API addon): {n} x {W}x{H} detect per batch (the C2 frames), facetrackr.Tracker.track() on a {W}x{H} canvas with one drifting face, "
                                   "and the C5 loop (8 x 1080p feeds per step) through ccv.DeviceBatch.detectStep / trackStep"}
        return j
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


# ---------------------------------------------------------------------------------------------------------------------
# launcher / stub


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) under torch.distributed.run and pass their
    output through.  Fails loudly when fewer than N GPUs are visible."""
    import socket

    stub = os.environ.get("HT_BENCH_STUB") == "1"
    if not stub:
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} GPU(s) visible to this process — refusing to report an N-GPU number from fewer devices")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HT_BENCH_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def stub_bench(env, a):
    """HT_BENCH_STUB=1 (tests/test_distributed_cpu.py): the N-rank plumbing of this file — launcher, rank environment, timed rounds
    with barriers and max over ranks, the all-gather of best-face records and its verification, the one JSON line of rank 0 — on
    gloo / CPU tensors with a stand-in step.  Never a measurement: the line says so."""
    torch, dist, rank, world = env.torch, env.dist, env.rank, env.world
    from headtrackr_amd import distributed as hd
    from headtrackr_amd.native import RECT_DTYPE

    nf = a.frames or 6
    total = nf * world
    best = np.zeros(nf, dtype=RECT_DTYPE)
    best["x"] = 10.0 * rank + np.arange(nf)
    best["confidence"] = -1.0 - rank
    best["neighbors"] = 1 + (np.arange(nf) % 3)
    state = {}

    def run_steps(k):
        for _ in range(k):
            time.sleep(0.0005)  # the stand-in for a detect step
            rec = hd.pack_best_records(best, rank * nf, nf)
            state["rec"] = rec
            state["gathered"] = hd.allgather_records(torch.from_numpy(rec), world, nf)

    run_steps(max(a.warmup, 1))
    dts = env.timed_rounds(run_steps, a.steps, a.rounds, target_s=0.05)
    dt, spread = round_stats(dts, a.steps)
    everyone = [None] * world
    if world > 1:
        dist.all_gather_object(everyone, state["rec"])
    else:
        everyone = [state["rec"]]
    if rank != 0:
        return None
    got = state["gathered"].numpy()
    ok = all(np.array_equal(got[r], everyone[r]) for r in range(world))
    return {"metric": "STUB — launcher / collective plumbing only, not a measurement", "value": round(total * a.steps / dt, 2), "unit": "frames/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, **spread, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "stub",
            "config": {"workload": "stub", "frames_per_gpu": nf, "frames_total": total}, "ranks": world, "backend": "gloo", "allgather_verified": bool(ok),
            "launched_by_bench": os.environ.get("HT_BENCH_LAUNCHED") == "1"}


def main():
    a = parse()
    stub = os.environ.get("HT_BENCH_STUB") == "1"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)  # does not return
    # The ONE JSON line owns stdout: native libraries print there too (RCCL's version banner sits in C stdio's buffer until the process
    # exits and would land BEHIND the line) — file descriptor 1 is handed to stderr for everything but the line itself.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: one rank per GPU (run `python bench.py --gpus N` without a launcher, or pass matching values)")
    if stub:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        line = stub_bench(Env(torch, dist, rank, world, local, stub=True), a)
        if rank == 0:
            print(json.dumps(line), file=line_out, flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    env = Env(torch, dist, rank, world, local)
    t_run = time.perf_counter()

    if a.workload == "c5":
        prim = stream_bench(env, a, feeds=max(1, a.feeds), steps=a.steps, warm_cycles=max(a.warmup, 1), cpu_seconds=a.cpu_seconds)
        metric = "frames/sec streaming 1920x1080 feeds (detect every 30th frame, camshift between); frames resident in HBM (pcie_inclusive: host -> GPU every frame)"
        sub = {}
    else:
        if a.workload == "c3":
            prim = c3_bench(env, a, a.steps, a.warmup, a.cpu_seconds)
            metric = "frames/sec (1 full-cascade detect + 60 camshift track per stream) at 320x240"
        else:
            prim = detect_bench(env, a, a.workload, a.steps, a.warmup, scaling=a.scaling, frames_per_gpu=a.frames, cpu_seconds=a.cpu_seconds, prewarm=a.prewarm)
            W, H, _ = GEOM[a.workload]
            metric = f"frames/sec full-cascade detect at {W}x{H}"
        sub = {}
        if a.workload == "c2" and a.scaling == "weak" and not a.no_sub:
            # the rest of BASELINE.json's metric in the same line: 1280x720 detect (per-GPU batch of configs[3]; weak and strong),
            # detect + camshift (configs[2]) and the streaming config (configs[4]), each with its own bounded budget
            tag = "c4_1gpu" if world == 1 else "c4"
            sub[tag] = detect_bench(env, a, "c4", SUB_STEPS["c4"], 10, cpu_seconds=a.cpu_seconds, prewarm=0.1, full=False)
            sub["c4_strong"] = detect_bench(env, a, "c4", SUB_STEPS["c4_strong"], 4, scaling="strong", cpu_seconds=0, full=False)
            sub["c3"] = c3_bench(env, a, SUB_STEPS["c3"], 2, a.cpu_seconds * 0.7)
            # C2 with a working set far beyond the 256 MB Infinity Cache (1024 frames: 315 MB of RGBA + 450 MB of pyramid per batch in
            # flight): the "HBM" rates of the 256-frame headline are partly MALL hits, this point says what the path does without them
            sub["c2_large"] = detect_bench(env, a, "c2", 60, 6, frames_per_gpu=1024, cpu_seconds=0, prewarm=0.1, full=False, unique=256)
            # BASELINE.json configs[4] is 8 feeds over the node: 8 / N per GPU (N = 1: all eight on the one GPU)
            feeds_per_gpu = max(1, 8 // world)
            one = stream_bench(env, a, feeds=1, steps=SUB_STEPS["c5"], cpu_seconds=a.cpu_seconds * 0.7)
            many = stream_bench(env, a, feeds=feeds_per_gpu, steps=SUB_STEPS["c5"], cpu_seconds=0) if feeds_per_gpu > 1 else one
            if rank == 0:
                if many is not one:
                    many["one_feed"] = one
                    many["feeds_vs_1"] = round(many["value"] / one["value"] * 1.0, 2)
                    many["feeds_vs_1_pcie_inclusive"] = round(many["pcie_inclusive"]["value"] / one["pcie_inclusive"]["value"], 2)
                    many["cpu_baseline"] = one["cpu_baseline"]
                    many["vs_cpu"] = round(many["value"] / one["cpu_baseline"]["value"], 1) if one.get("cpu_baseline") else None
            sub["c5"] = many
            if rank == 0 and world == 1:
                sub["js_host"] = js_host_bench(2.0)
            if world == 1:
                # what the exchange step costs a step, measured where one GPU can measure it: the same C2 / C4 blocks with the all-gather of
                # best-face records forced through a 1-rank RCCL group (event wait, pack, pinned copy, H2D, ncclAllGather on RCCL's stream)
                t_init = time.perf_counter()
                try:
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                    import socket

                    sk = socket.socket()
                    sk.bind(("127.0.0.1", 0))
                    os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
                    sk.close()
                    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
                    g = {}
                    for nm, st_, key in (("c2", a.steps, None), ("c4", SUB_STEPS["c4"], "c4_1gpu")):  # the same block lengths as the records they are compared with
                        r = detect_bench(env, a, nm, st_, 10, cpu_seconds=0, prewarm=0.1, full=False, gather=True)
                        ref = sub[key] if key else None
                        g[nm] = dict(ms_per_step=r["ms_per_step"], ms_per_step_min=r["ms_per_step_min"], ms_per_step_max=r["ms_per_step_max"], value=r["value"],
                                     allgather_verified=r.get("allgather_verified"), without_exchange_ms_per_step=ref["ms_per_step"] if ref else None)
                    g["rccl_init_and_runs_s"] = round(time.perf_counter() - t_init, 1)
                    g["what"] = "the timed C2 / C4 blocks with every step's exchange forced on ONE GPU: event wait + pack + pinned copy + H2D + all_gather_into_tensor on a 1-rank RCCL group"
                    sub["gather_n1"] = g
                    dist.destroy_process_group()
                except Exception as e:  # never lose the line to the optional leg
                    sub["gather_n1"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        copy_gbs = device_copy_ceiling(torch)
        for r in [prim] + [v for v in sub.values() if v]:
            if r.get("roofline"):
                r["roofline"]["device_copy_gbs"] = round(copy_gbs, 1)
                r["roofline"]["frac_of_device_copy"] = round(r["roofline"]["achieved"] / copy_gbs, 5)
        line = {"metric": metric, "value": prim["value"], "unit": "frames/s", "n_gpus": world, "ranks": world, "steps": prim["steps"], "warmup": prim["warmup"],
                "ms_per_step": prim["ms_per_step"], "higher_is_better": True, "scaling": prim["scaling"], "vs_baseline": None, "dtype": "u8",
                "data": "synthetic", "launched_by_bench": os.environ.get("HT_BENCH_LAUNCHED") == "1"}
        line.update({k: v for k, v in prim.items() if k not in line})
        if sub.get("gather_n1") and "c2" in sub["gather_n1"]:
            g = sub["gather_n1"]
            g["c2"]["without_exchange_ms_per_step"] = prim["ms_per_step"]
            for nm in ("c2", "c4"):
                if g[nm].get("without_exchange_ms_per_step"):
                    g[nm]["exchange_cost_frac"] = round(g[nm]["ms_per_step"] / g[nm]["without_exchange_ms_per_step"] - 1.0, 4)
        if sub:
            line["sub"] = sub
            if sub.get("c4_1gpu") and sub["c4_1gpu"].get("vs_cpu"):
                line["north_star_720p_vs_reference_js"] = sub["c4_1gpu"]["vs_cpu"]  # target: >= 30x on 1280x720 detect at 1 GPU
        line["bench_wall_s"] = round(time.perf_counter() - t_run, 1)
        print(json.dumps(line), file=line_out, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
