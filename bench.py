#!/usr/bin/env python3
"""bench.py — frames/s of the detect(+camshift) hot path on N MI355X, ONE compact JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--scaling weak|strong] [--feeds K]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(torch.distributed.run, one rank per GPU over RCCL) and fails loudly when fewer than N GPUs are visible; under a
launcher WORLD_SIZE must equal --gpus.  --force-launcher does the same for N = 1 (torchrun + a 1-rank RCCL group + the
verified all-gather on one GPU).

A "step" is one pass of the hot path over one batch of synthetic frames that are already resident in HBM:
gray -> 39-level pyramid -> full BBF cascade scan -> raw hits copied back, sorted, converted to rects, grouped and
reduced to the best face per frame (ht_detect_enqueue + ht_detect_collect_best = ccv.detect_objects(..., 5, 1) +
facetrackr's selection, /root/reference/src/ccv.js:109-333, facetrackr.js:147-175), plus, for N > 1, one RCCL
all-gather of the fixed-size per-frame best-face rectangles.

Workloads (BASELINE.json configs) — benchlib/detect.py, c3.py, c5.py:
  c2  256 x 320x240 detect per GPU — the headline `value` (the configuration the metric is quoted on);
  c4  1280x720 detect, 128 frames per GPU (weak) or 1024 frames in total (--scaling strong: 1024 / N per GPU);
  c3  256 streams of 320x240: detect once + initTracker + 60 camshift track() calls per step;
  c5  --feeds K live 1920x1080 feeds per GPU (one batch of K frames per time step), detect every 30th frame.
The default run (c2) also measures c4 (weak + strong), c3, c5, the depth-1 / PCIe-inclusive / single-frame variants, the
JavaScript host and the exchange step's cost, each with its own bounded budget.  The LINE carries the contract's keys
for the headline plus scalars of the sub-records (value_720p, c3_value, c5_value, ...: benchlib/line.py, < 4 KB,
asserted); the full record tree — every sub-record with its own `roofline`, `cpu_baseline`, configuration and
notes — is written to bench_sub.json (and gpurun_out/bench_sub.json).  --no-sub skips the sub-records (profiler runs).

Timing: the K timed steps (barrier + synchronize on both sides, max over ranks) are repeated R times back to back
("rounds"; R is chosen so that the timed work is ~0.4 s, K stays what the caller passed) and the MEDIAN block is
reported, with the spread (`ms_per_step_min` / `ms_per_step_max`): a 20-step block of C2 is 5 ms of GPU time, far too
short to quote on its own.  "dominant kernel" of a roofline = the kernel with the largest device time per step (the
sum over its launches).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import DEFAULT_STEPS, GEOM, SUB_STEPS, Env, device_copy_ceiling, free_port  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0,
                    help="timed steps (default: 1000 for c2, 300 for c4 / c5, 20 for c3: ~0.3-0.5 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed warm-up steps (default: a tenth of --steps, >= 3)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="strong: total frames fixed at 8 x the per-GPU default (c4: 1024) and split over the ranks")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (default: 256 for c2/c3, 128 for c4)")
    ap.add_argument("--unique", type=int, default=0, help="distinct synthetic frames generated (tiled to --frames)")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="budget of each cpu_baseline leg (0 = skip)")
    ap.add_argument("--flags", type=int, default=0, help="ht_detect flags (A/B of scan schedules)")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="batches in flight (contexts on their own HIP streams); 1 = enqueue+collect strictly in turn; "
                         "0 = auto: 3 at 320x240, 2 at 1280x720 (measured, see benchlib/detect.py)")
    ap.add_argument("--prewarm", type=float, default=0.2,
                    help="seconds of untimed steady-state work before the warm-up steps (0 for profiler runs)")
    ap.add_argument("--no-sub", action="store_true", help="only the primary workload (no sub-records, no side file)")
    ap.add_argument("--no-requeue", action="store_true",
                    help="A/B: enqueue a context's next batch only after its results were post-processed")
    ap.add_argument("--feeds", type=int, default=1,
                    help="c5: live feeds per GPU; 8 on one GPU is the N = 1 point of BASELINE.json configs[4]")
    ap.add_argument("--rounds", type=int, default=0,
                    help="repetitions of the K-step timed block (median reported); 0 = auto: ~0.4 s, 3..25 rounds")
    ap.add_argument("--options", default="", help="ht_config.options of every context (result-preserving schedule A/B)")
    ap.add_argument("--force-launcher", action="store_true",
                    help="start the rank(s) through torch.distributed.run even for --gpus 1 (RCCL init + all-gather)")
    a = ap.parse_args()
    if a.steps <= 0:
        a.steps = DEFAULT_STEPS[a.workload]
    if a.warmup < 0:
        a.warmup = max(3, a.steps // 10) if a.workload != "c5" else 1
    return a


def sub_records(env, a, prim, t_budget):
    """the rest of BASELINE.json's metric next to the C2 headline: 1280x720 detect (per-GPU batch of configs[3]; weak
    and strong), detect + camshift (configs[2]), the streaming config (configs[4]); at N = 1 also the single-frame
    latencies, the JavaScript host and the exchange step's cost on one GPU.  Every rank runs the GPU legs."""
    from benchlib.c3 import c3_bench
    from benchlib.c5 import stream_bench
    from benchlib.detect import detect_bench, single_frame_latency
    from benchlib.js_host import js_host_bench

    torch, dist, rank, world, local = env.torch, env.dist, env.rank, env.world, env.local
    sub = {}
    one_gpu = world == 1
    tag = "c4_1gpu" if one_gpu else "c4"
    sub[tag] = detect_bench(env, a, "c4", SUB_STEPS["c4"], 10, cpu_seconds=a.cpu_seconds, prewarm=0.1, full=False,
                            extras=one_gpu)
    sub["c4_strong"] = detect_bench(env, a, "c4", SUB_STEPS["c4_strong"], 4, scaling="strong", cpu_seconds=0, full=False,
                                    extras=False)
    sub["c3"] = c3_bench(env, a, SUB_STEPS["c3"], 2, a.cpu_seconds * 0.7)
    if one_gpu:
        # C2 with a working set far beyond the 256 MB Infinity Cache (1024 frames: 315 MB of RGBA + 450 MB of pyramid
        # per batch in flight): what the path does without MALL hits
        sub["c2_large"] = detect_bench(env, a, "c2", 60, 6, frames_per_gpu=1024, cpu_seconds=0, prewarm=0.1, full=False,
                                       unique=256, extras=False)
    # BASELINE.json configs[4] is 8 feeds over the node: 8 / N per GPU (N = 1: all eight on the one GPU)
    feeds_per_gpu = max(1, 8 // world)
    one = stream_bench(env, a, feeds=1, steps=SUB_STEPS["c5"], cpu_seconds=a.cpu_seconds * 0.7)
    many = stream_bench(env, a, feeds=feeds_per_gpu, steps=SUB_STEPS["c5"], cpu_seconds=0) if feeds_per_gpu > 1 else one
    if rank == 0 and many is not one:
        many["one_feed"] = one
        many["feeds_vs_1"] = round(many["value"] / one["value"] * 1.0, 2)
        many["feeds_vs_1_pcie_inclusive"] = round(many["pcie_inclusive"]["value"] / one["pcie_inclusive"]["value"], 2)
        many["cpu_baseline"] = one["cpu_baseline"]
        if one.get("cpu_baseline"):
            many["vs_cpu"] = round(many["value"] / one["cpu_baseline"]["value"], 1)
    sub["c5"] = many
    if not one_gpu:
        return sub
    sub["latency_1frame"] = {f"{w}x{h}": single_frame_latency(env, w, h, a.flags)
                             for w, h in ((320, 240), (1280, 720), (1920, 1080))}
    sub["js_host"] = js_host_bench(2.0)
    # what the exchange step costs a step, measured where one GPU can measure it: the same C2 / C4 blocks with the
    # all-gather of best-face records forced through a 1-rank RCCL group (event wait, pack, pinned copy, H2D,
    # ncclAllGather on RCCL's stream).  Skipped when the process already runs under a launcher's group.
    if dist.is_initialized():
        return sub
    try:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        t_init = time.perf_counter()
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        t = torch.zeros(8, device="cuda")
        dist.all_reduce(t)  # the communicator is created lazily: the first collective pays for it
        torch.cuda.synchronize()
        g = {"rccl_init_s": round(time.perf_counter() - t_init, 2)}
        # the same block lengths as the records they are compared with
        for nm, st_, ref in (("c2", a.steps, prim), ("c4", SUB_STEPS["c4"], sub[tag])):
            r = detect_bench(env, a, nm, st_, 10, cpu_seconds=0, prewarm=0.1, full=False, gather=True, extras=False)
            g[nm] = dict(ms_per_step=r["ms_per_step"], ms_per_step_min=r["ms_per_step_min"],
                         ms_per_step_max=r["ms_per_step_max"], value=r["value"],
                         allgather_verified=r.get("allgather_verified"), without_exchange_ms_per_step=ref["ms_per_step"],
                         exchange_cost_frac=round(r["ms_per_step"] / ref["ms_per_step"] - 1.0, 4))
        g["what"] = ("the timed C2 / C4 blocks with every step's exchange forced on ONE GPU: event wait + pack + pinned "
                     "copy + H2D + all_gather_into_tensor on a 1-rank RCCL group")
        sub["gather_n1"] = g
        dist.destroy_process_group()
    except Exception as e:  # never lose the line to the optional leg
        sub["gather_n1"] = {"error": f"{type(e).__name__}: {e}"}
    return sub


RCCL_INIT_BUDGET_S = 240.0  # communicator creation + first collective; 8 ranks of one node take a few seconds, a 1-rank ncclCommInitAll
                            # took ~50 s on one test box (tests/test_js_host.py)


def init_group(torch, dist, backend, rank, world, local):
    """init_process_group + the first collective (communicators are created lazily), bounded: a rendezvous or a communicator that does
    not come up inside RCCL_INIT_BUDGET_S ends the run with a message and a non-zero exit code instead of hanging the driver's N-GPU
    slot.  Returns the MAX over the ranks of the seconds it took (every rank's line would carry the same figure)."""
    import datetime

    budget = float(os.environ.get("HT_BENCH_INIT_BUDGET_S", RCCL_INIT_BUDGET_S))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    t_init = time.perf_counter()
    try:
        kw = dict(device_id=torch.device("cuda", local)) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=budget), **kw)
        t = torch.zeros(8, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        if backend == "nccl":
            torch.cuda.synchronize()
    except Exception as e:
        sys.stderr.write(f"bench.py rank {rank}/{world}: {backend} process group did not come up within {budget:.0f} s "
                         f"({type(e).__name__}: {e})\n")
        raise SystemExit(3)
    mine = time.perf_counter() - t_init
    if mine > budget:
        sys.stderr.write(f"bench.py rank {rank}/{world}: {backend} init + first collective took {mine:.1f} s (budget {budget:.0f} s)\n")
        raise SystemExit(3)
    tm = torch.tensor([mine], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    return round(float(tm.item()), 2)


def main():
    a = parse()
    stub = os.environ.get("HT_BENCH_STUB") == "1"
    if (a.gpus > 1 or a.force_launcher) and "WORLD_SIZE" not in os.environ:
        from benchlib.launch import launch_ranks

        launch_ranks(a, __file__)  # does not return
    # The ONE JSON line owns stdout: native libraries print there too (RCCL's version banner sits in C stdio's buffer
    # until the process exits and would land BEHIND the line) — file descriptor 1 is handed to stderr for everything
    # but the line itself.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    from benchlib import line as bl

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = os.environ.get("HT_BENCH_LAUNCHED") == "1" or "TORCHELASTIC_RUN_ID" in os.environ
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: one rank per GPU (run `python bench.py --gpus N` "
                         "without a launcher, or pass matching values)")
    if stub:
        from benchlib.launch import stub_bench

        init_s = init_group(torch, dist, "gloo", rank, world, local) if world > 1 else None
        prim = stub_bench(Env(torch, dist, rank, world, local, stub=True), a)
        if rank == 0:
            line = bl.compose("STUB — launcher / collective plumbing only, not a measurement", prim, {}, world,
                              dict(backend="gloo", rccl_init_s=init_s))
            line["data"] = "stub"
            bl.emit(line, prim, line_out, write_sub=False)
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    rccl_init_s = None
    if world > 1 or (launched and "MASTER_PORT" in os.environ):  # under a launcher even ONE rank gets its RCCL group
        rccl_init_s = init_group(torch, dist, "nccl", rank, world, local)
    env = Env(torch, dist, rank, world, local)
    t_run = time.perf_counter()
    gather = True if (world == 1 and dist.is_initialized()) else None

    from benchlib.c3 import c3_bench
    from benchlib.c5 import stream_bench
    from benchlib.detect import detect_bench

    sub = {}
    if a.workload == "c5":
        prim = stream_bench(env, a, feeds=max(1, a.feeds), steps=a.steps, warm_cycles=max(a.warmup, 1),
                            cpu_seconds=a.cpu_seconds)
        metric = ("frames/sec streaming 1920x1080 feeds (detect every 30th frame, camshift between); frames resident in "
                  "HBM")
    elif a.workload == "c3":
        prim = c3_bench(env, a, a.steps, a.warmup, a.cpu_seconds)
        metric = "frames/sec (1 full-cascade detect + 60 camshift track per stream) at 320x240"
    else:
        prim = detect_bench(env, a, a.workload, a.steps, a.warmup, scaling=a.scaling, frames_per_gpu=a.frames,
                            cpu_seconds=a.cpu_seconds, prewarm=a.prewarm, gather=gather,
                            extras=world == 1 and not a.no_sub)
        W, H, _ = GEOM[a.workload]
        metric = f"frames/sec full-cascade detect at {W}x{H}"
        if a.workload == "c2" and a.scaling == "weak" and not a.no_sub:
            sub = sub_records(env, a, prim, t_run)
    if rank == 0:
        copy_gbs = device_copy_ceiling(torch)
        for r in [prim] + [v for v in sub.values() if isinstance(v, dict)]:
            if r.get("roofline"):
                r["roofline"]["device_copy_gbs"] = round(copy_gbs, 1)
                r["roofline"]["frac_of_device_copy"] = round(r["roofline"]["achieved"] / copy_gbs, 5)
        extra = dict(rccl_init_s=rccl_init_s, device_copy_gbs=round(copy_gbs, 1),
                     bench_wall_s=round(time.perf_counter() - t_run, 1))
        line = bl.compose(metric, prim, sub, world, extra)
        full = dict(line=dict(line), primary=prim, sub=sub)
        bl.emit(line, full, line_out, write_sub=not a.no_sub)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
