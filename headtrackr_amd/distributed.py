"""Frame sharding across the GPUs of one node and the one exchange step of the path: an all-gather of fixed-size
per-frame detection records (RCCL over xGMI when the tensors live on GPUs; the same code runs on gloo/CPU tensors in the
tests).  One process per GPU (torch.distributed); frames are independent, so there is no other collective.

Record layout (float64 x 8 per frame, 64 B): [raw hit count, scale, q, x, y, confidence of the best raw hit, 0, 0]
— the strict-'>' arg-max of facetrackr.js:157-165 applied to the raw hits.  Payload is KBs: latency-bound.
"""
from __future__ import annotations

import numpy as np

RECORD_F64 = 8


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous block of frames owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_records(hits: np.ndarray, counts: np.ndarray, nframes: int) -> np.ndarray:
    rec = np.zeros((nframes, RECORD_F64), dtype=np.float64)
    k = 0
    for f in range(nframes):
        c = int(counts[f])
        if c:
            h = hits[k : k + c]
            b = h[int(np.argmax(h["sum"]))]  # first maximum, like the reference's strict '>' scan
            rec[f, :6] = (c, b["scale"], b["q"], b["x"], b["y"], b["sum"])
        k += c
    return rec


def allgather_records(rec_local, world: int, max_frames_per_rank: int):
    """All-gathers one [max_frames_per_rank, 8] float64 tensor per rank into [world, max_frames_per_rank, 8].
    rec_local must already be padded to max_frames_per_rank rows and live on the device the backend expects."""
    import torch
    import torch.distributed as dist

    assert rec_local.shape == (max_frames_per_rank, RECORD_F64) and rec_local.dtype == torch.float64
    out = torch.empty((world, max_frames_per_rank, RECORD_F64), dtype=torch.float64, device=rec_local.device)
    if world == 1:
        out[0].copy_(rec_local)
        return out
    dist.all_gather_into_tensor(out.view(world * max_frames_per_rank, RECORD_F64), rec_local.contiguous())
    return out


def unshard(gathered: np.ndarray, n_total: int, world: int) -> np.ndarray:
    """[world, max_per_rank, 8] -> [n_total, 8] in global frame order (drops the padding rows)."""
    parts = []
    for r in range(world):
        a, b = shard_range(n_total, r, world)
        parts.append(gathered[r, : b - a])
    return np.concatenate(parts, axis=0)
