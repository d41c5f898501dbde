"""The N > 1 path on CPU: 2 processes, gloo backend.  Each rank produces the detection records of its block of frames
(here with the CPU oracle standing in for the GPU — this is a test of the sharding + all-gather plumbing only), the
records are all-gathered, and every rank must hold exactly the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from headtrackr_amd import distributed as hd
from headtrackr_amd import synth
from headtrackr_amd.cascade import load_cascade

N_FRAMES, W, H = 7, 160, 120


def _records_for(frames, first):
    """the all-gathered record = the frame's bounding box as facetrackr selects it (grouped rect of highest confidence)"""
    from oracle import ht_oracle as ho

    best = ho.best_faces(frames, load_cascade().blob, 1)
    return hd.pack_best_records(best, first)


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = [synth.make(dict(family="mixed", index=i, seed0=1234), W, H) if i % 3 else synth.face_frame(W, H, [(30 + i, 20, 64)]) for i in range(N_FRAMES)]
    a, b = hd.shard_range(N_FRAMES, rank, world)
    maxper = max(hd.shard_range(N_FRAMES, r, world)[1] - hd.shard_range(N_FRAMES, r, world)[0] for r in range(world))
    rec = np.zeros((maxper, hd.RECORD_F64))
    rec[: b - a] = _records_for(frames[a:b], a)
    gathered = hd.allgather_records(torch.from_numpy(rec), world, maxper).numpy()
    merged = hd.unshard(gathered, N_FRAMES, world)
    full = _records_for(frames, 0)
    ret[rank] = (bool(np.array_equal(merged, full)) and int((merged[:, 5] > 0).sum()) >= 2 and
                 bool(np.array_equal(merged[:, 6], np.arange(N_FRAMES))) and bool(np.all(merged[:, 7] == 1.0)))
    dist.destroy_process_group()


def test_pack_best_records_layout():
    from headtrackr_amd.native import RECT_DTYPE

    best = np.zeros(3, dtype=RECT_DTYPE)
    best[0] = (10.5, 20.25, 90.0, 90.0, 3.5, 9, 0)
    best[1] = (0, 0, 0, 0, -10000.0, 0, 0)
    best[2] = (1, 2, 3, 4, -1.0, 2, 0)
    rec = hd.pack_best_records(best, first_frame=128, rows=5)
    assert rec.shape == (5, hd.RECORD_F64)
    assert rec[0].tolist() == [10.5, 20.25, 90.0, 90.0, 3.5, 9.0, 128.0, 1.0]
    assert rec[1].tolist() == [0, 0, 0, 0, -10000.0, 0, 129.0, 1.0]
    assert not rec[3:].any()


def test_shard_ranges_cover_everything():
    for n in (1, 7, 256, 1024):
        for world in (1, 2, 3, 8):
            spans = [hd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_allgather_equals_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def _bench(args, env_extra, timeout=180):
    import json
    import subprocess

    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_n_launches_n_ranks():
    """`python bench.py --gpus 2` — the one command the driver uses — must start 2 ranks itself: launcher -> torch.distributed.run ->
    rank environment -> barriers / max over ranks -> all-gather of the best-face records (verified) -> ONE line with n_gpus 2.
    HT_BENCH_STUB=1 swaps RCCL for gloo and the detect step for a stand-in, so the plumbing runs without GPUs."""
    r, line = _bench(["--gpus", "2", "--steps", "4", "--warmup", "1"], {"HT_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["launched_by_bench"] is True
    assert line["allgather_verified"] is True and line["steps"] == 4 and line["rounds"] >= 3
    assert line["ms_per_step_min"] <= line["ms_per_step"] <= line["ms_per_step_max"]
    assert line["config"]["frames_total"] == 2 * line["config"]["frames_per_gpu"]
    assert line["rank_ms_per_step_min"] <= line["rank_ms_per_step_max"] and line["data"] == "stub"
    assert len(r.stdout.strip().splitlines()[-1]) < 2000  # the one line stays a headline (benchlib/line.py)


def test_bench_gpus_8_stub_is_cheap_and_loud():
    """The driver's N = 8 run: `python bench.py --gpus 8` must come up, finish well inside a minute of plumbing and print ONE line with
    ranks 8 (stand-in step, gloo); the line carries the group's init time (max over the ranks).  And a group that cannot come up inside
    the budget ends with a message and a non-zero exit code, not a hang: rank 1 of a 2-rank world that never gets its peer."""
    import subprocess
    import time

    t0 = time.time()
    r, line = _bench(["--gpus", "8", "--steps", "3", "--warmup", "1"], {"HT_BENCH_STUB": "1"}, timeout=170)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    assert wall < 60.0, f"bench.py --gpus 8 (stub) took {wall:.0f} s"
    assert line["n_gpus"] == 8 and line["ranks"] == 8 and line["allgather_verified"] is True
    assert line["config"]["frames_total"] == 8 * line["config"]["frames_per_gpu"]
    assert 0.0 <= line["rccl_init_s"] < 60.0
    assert len([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]) == 1
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HT_BENCH_STUB="1", WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HT_BENCH_INIT_BUDGET_S="3")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "did not come up within 3 s" in r.stderr, r.stderr[-1500:]
    assert time.time() - t0 < 60.0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_to_mislabel_the_gpu_count():
    """fewer GPUs than --gpus (here: none) is an error, not a silent 1-GPU run labelled n_gpus 1; so is a launcher whose
    WORLD_SIZE disagrees with --gpus"""
    r, line = _bench(["--gpus", "2", "--steps", "2"], {"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode != 0 and line is None and "--gpus 2" in r.stderr and "visible" in r.stderr
    env = dict(os.environ, HT_BENCH_STUB="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
