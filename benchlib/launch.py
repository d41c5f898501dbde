"""`python bench.py --gpus N` without a launcher, and the CPU stand-in that tests the N-rank plumbing."""
import os
import subprocess
import sys
import time

import numpy as np

from .common import free_port, round_stats


def launch_ranks(a, script):
    """Start a.gpus ranks (one per GPU) under torch.distributed.run and pass their output through.  Fails loudly when
    fewer than N GPUs are visible.  Does not return."""
    stub = os.environ.get("HT_BENCH_STUB") == "1"
    if not stub:
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} GPU(s) visible to this process — refusing to "
                             "report an N-GPU number from fewer devices")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(script)]
    cmd += [x for x in sys.argv[1:] if x != "--force-launcher"]
    env = dict(os.environ, HT_BENCH_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def stub_bench(env, a):
    """HT_BENCH_STUB=1 (tests/test_distributed_cpu.py): the N-rank plumbing of bench.py — launcher, rank environment,
    timed rounds with barriers and max over ranks, the all-gather of best-face records and its verification, the one
    JSON line of rank 0 — on gloo / CPU tensors with a stand-in step.  Never a measurement: the line says so."""
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
    rank_ms = env.gather_scalar(float(np.median(env.last_own)) / a.steps * 1e3)
    everyone = [None] * world
    if world > 1:
        dist.all_gather_object(everyone, state["rec"])
    else:
        everyone = [state["rec"]]
    if rank != 0:
        return None
    got = state["gathered"].numpy()
    ok = all(np.array_equal(got[r], everyone[r]) for r in range(world))
    return {"value": round(total * a.steps / dt, 2), "unit": "frames/s", "steps": a.steps, "warmup": a.warmup, **spread,
            "scaling": "weak", "config": {"workload": "stub", "frames_per_gpu": nf, "frames_total": total},
            "allgather_verified": bool(ok), "rank_ms_per_step_min": round(min(rank_ms), 4),
            "rank_ms_per_step_max": round(max(rank_ms), 4), "roofline": None, "cpu_baseline": None}
