"""Frame sharding across the GPUs of one node and the one exchange step of the path: an all-gather of fixed-size
per-frame detection records (RCCL over xGMI when the tensors live on GPUs; the same code runs on gloo/CPU tensors in the
tests).  One process per GPU (torch.distributed); frames are independent, so there is no other collective.

Record layout (float64 x 8 per frame, 64 B): the frame's bounding box as facetrackr.Tracker.doVJDetection selects it
(facetrackr.js:147-175: ccv.detect_objects(..., 5, 1) grouped rects, strict-'>' arg-max of confidence) —
[x, y, width, height, confidence, neighbors, global frame index, 1.0]; a frame without a detection carries
neighbors = 0 and confidence = -10000 (facetrackr.js:239); padding rows (ranks holding fewer frames) are all zero.
Payload is KBs: latency-bound.  (pack_records — the best RAW hit per frame — is kept for hosts that gather before grouping.)
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
    """Per-frame best raw hit (first maximum of `sum`, like the reference's strict '>' scan) as [nframes, 8] float64.
    `hits` is sorted by frame, `counts[f]` hits belong to frame f.  Vectorised: this runs once per step on every rank."""
    rec = np.zeros((nframes, RECORD_F64), dtype=np.float64)
    cnt = np.asarray(counts[:nframes], dtype=np.int64)
    nz = np.nonzero(cnt)[0]
    if nz.size == 0:
        return rec
    starts = np.cumsum(cnt) - cnt
    total = int(cnt.sum())
    s = np.asarray(hits["sum"][:total], dtype=np.float64)
    seg = np.repeat(np.arange(nz.size), cnt[nz])      # hit -> index of its (non-empty) frame
    mx = np.maximum.reduceat(s, starts[nz])            # per-frame maximum
    cand = np.nonzero(s == mx[seg])[0]                 # hits that reach it, in hit order
    first = np.full(nz.size, total, dtype=np.int64)
    np.minimum.at(first, seg[cand], cand)              # the first one per frame
    b = hits[first]
    rec[nz, 0] = cnt[nz]
    rec[nz, 1] = b["scale"]
    rec[nz, 2] = b["q"]
    rec[nz, 3] = b["x"]
    rec[nz, 4] = b["y"]
    rec[nz, 5] = b["sum"]
    return rec


def pack_best_records(best: np.ndarray, first_frame: int, rows: int | None = None) -> np.ndarray:
    """ht_best_faces output (one ht_rect per local frame) -> [rows, 8] float64 records; rows >= len(best) pads with zeros."""
    n = len(best)
    rec = np.zeros((rows if rows is not None else n, RECORD_F64), dtype=np.float64)
    rec[:n, 0] = best["x"]
    rec[:n, 1] = best["y"]
    rec[:n, 2] = best["width"]
    rec[:n, 3] = best["height"]
    rec[:n, 4] = best["confidence"]
    rec[:n, 5] = best["neighbors"]
    rec[:n, 6] = first_frame + np.arange(n)
    rec[:n, 7] = 1.0
    return rec


def allgather_records(rec_local, world: int, max_frames_per_rank: int, out=None, force_collective: bool = False):
    """All-gathers one [max_frames_per_rank, 8] float64 tensor per rank into [world, max_frames_per_rank, 8].
    rec_local must already be padded to max_frames_per_rank rows and live on the device the backend expects.
    out: optional preallocated result tensor (a steady-state caller reuses one per batch in flight).
    force_collective: with one rank, still issue the collective on the initialised process group (measures the exchange step's cost on one GPU)."""
    import torch
    import torch.distributed as dist

    assert rec_local.shape == (max_frames_per_rank, RECORD_F64) and rec_local.dtype == torch.float64
    if out is None:
        out = torch.empty((world, max_frames_per_rank, RECORD_F64), dtype=torch.float64, device=rec_local.device)
    assert out.shape == (world, max_frames_per_rank, RECORD_F64) and out.dtype == torch.float64
    if world == 1 and not (force_collective and dist.is_initialized()):  # force_collective: a 1-rank group still goes through the backend
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
