#!/usr/bin/env python3
"""tools/sim_scan_lds2.py — CPU model of the LDS bank conflicts of k_scan_tiles AS IT IS NOW (pair-mode stage 0 on aligned dword
reads, wave-private queues, byte reads in the stages after it) for different orders in which stage 0 walks a tile's windows.

A wave64 LDS read is serviced in two groups of 32 lanes; a group takes max over the 32 banks of the number of DISTINCT dwords asked
of that bank (MI355X_MICROARCH.md, LDS).  All reads of a stage are `window base + constant`: for the dword reads of stage 0 a
constant only rotates the banks, so one evaluation per group stands for all 30 reads; for the byte reads of later stages
(base is even) the pattern depends on the constant mod 4 only.

    python tools/sim_scan_lds2.py [nframes] [w h]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from headtrackr_amd import synth  # noqa: E402
from headtrackr_amd.cascade import load_cascade  # noqa: E402
from oracle import ht_oracle as ho  # noqa: E402
import sim_scan_lds as s1  # noqa: E402

NST = 8
TXH, TYH = 64, 32
PITCH0 = 152
ROWS0 = 2 * TYH + 22
P12 = PITCH0 * ROWS0


def group_cycles_dw(dw, valid):
    """dw, valid: [G, 32] -> cycles per group (>= 1)"""
    G = dw.shape[0]
    cyc = np.ones(G, dtype=np.int64)
    bank = dw & 31
    for g in range(G):
        v = valid[g]
        if not v.any():
            continue
        key = np.unique(dw[g][v])
        cyc[g] = np.bincount(key & 31, minlength=32).max()
    return cyc


def tile_shapes(W2, H2):
    ntx = -(-W2 // TXH)
    tw2 = (-(-W2 // ntx) + 7) & ~7
    nty = -(-H2 // TYH)
    th2 = -(-H2 // nty)
    th2 += th2 & 1
    return tw2, th2, -(-W2 // tw2), -(-H2 // th2)


def enum_pairs(scheme, tw2, th, tw):
    """order in which stage 0 walks the tile's window PAIRS: arrays (row, pair index, valid0, valid1) in lane order"""
    pw = tw2 // 2
    if scheme == "rowmajor":
        n = np.arange(pw * th)
        row, p = n // pw, n % pw
    elif scheme == "strip4":
        n = np.arange(pw * th)
        strip, rem = n // (4 * th), n % (4 * th)
        row, p = rem >> 2, strip * 4 + (rem & 3)
    elif scheme == "strip12":  # strips of 12 pairs, the remainder in strips of 4
        rows, ps = [], []
        p0 = 0
        while p0 < pw:
            wdt = 12 if pw - p0 >= 12 else 4
            n = np.arange(wdt * th)
            rows.append(n // wdt)
            ps.append(p0 + n % wdt)
            p0 += wdt
        row, p = np.concatenate(rows), np.concatenate(ps)
    else:
        raise ValueError(scheme)
    return row, p, 2 * p < tw, 2 * p + 1 < tw


def wave_pair_ranges(npairs, wv, nw, assign):
    """[lo, hi) ranges of walking-order pair indices wavefront wv evaluates in stage 0: "block" = a contiguous 1/nw of the 32-pair batches
    (the kernel as it is), "interleave" = every nw-th pass of 64 pairs"""
    nbat = -(-npairs // 32)
    if assign == "block":
        per = -(-nbat // nw)
        lo, hi = min(wv * per * 32, npairs), min((wv * per + per) * 32, npairs)
        return [(b, min(b + 64, hi)) for b in range(lo, hi, 64)]
    return [(b, min(b + 64, npairs)) for b in range(64 * wv, npairs, 64 * nw)]


def simulate(frames, c, scheme, assign="block"):
    offs = [np.array(s1.stage_offsets(c, j, PITCH0, P12)) for j in range(NST)]
    cls = [np.bincount(o & 3, minlength=4) for o in offs]  # byte-read constants per (offset mod 4)
    act = np.zeros(NST)
    issued = np.zeros(NST)
    for fr in frames:
        levels, arena = ho.pyramid(fr)
        for i in range(len(levels) - 12):
            death = s1.survivors_for_scale(c, levels, arena, i)
            if death is None:
                continue
            H2, W2 = death.shape
            tw2, th2, ntx, nty = tile_shapes(W2, H2)
            for ty in range(nty):
                for tx in range(ntx):
                    X0, Y0 = tx * tw2, ty * th2
                    tw, th = min(tw2, W2 - X0), min(th2, H2 - Y0)
                    d = death[Y0:Y0 + th, X0:X0 + tw]
                    row, p, v0, v1 = enum_pairs(scheme, tw2, th, tw)
                    npairs = len(row)
                    # waves: contiguous quarters of the 32-pair batches (64 ids), a pass = 64 pairs
                    queues = []
                    for wv in range(4):
                        q_ids = []
                        for b, e in wave_pair_ranges(npairs, wv, 4, assign):
                            r_, p_, a0, a1 = row[b:e], p[b:e], v0[b:e], v1[b:e]
                            n = e - b
                            pad = 64 - n
                            dw = np.concatenate([np.where(a0, r_ * (PITCH0 // 2) + p_, 0), np.zeros(pad, dtype=np.int64)])  # invalid lanes read at base 0
                            vv = np.ones(64, dtype=bool)  # every lane issues its read (invalid ones at the tile's base)
                            cyc = group_cycles_dw(dw.reshape(2, 32), vv.reshape(2, 32)).sum()
                            act[0] += 30 * cyc
                            issued[0] += 30 * 2
                            # survivors: even windows of the pass first, then the odd ones
                            for a, dx in ((a0, 0), (a1, 1)):
                                keep = a & (d[np.minimum(r_, th - 1), np.minimum(2 * p_ + dx, tw - 1)] > 0)
                                q_ids.append(np.stack([r_[keep], 2 * p_[keep] + dx], axis=1))
                        queues.append(np.concatenate(q_ids) if q_ids else np.zeros((0, 2), dtype=np.int64))
                    # later stages on the wave-private queues (the <=64 merge is modelled as the same byte reads)
                    for j in range(1, NST):
                        for wv in range(4):
                            q = queues[wv]
                            if len(q) == 0:
                                continue
                            B = 2 * (q[:, 0] * PITCH0 + q[:, 1])
                            n = len(B)
                            pad = (-n) % 64
                            Bp = np.concatenate([B, np.zeros(pad, dtype=np.int64)]).reshape(-1, 32)
                            vp = np.concatenate([np.ones(n, dtype=bool), np.ones(pad, dtype=bool)]).reshape(-1, 32)
                            for k in range(4):
                                if cls[j][k] == 0:
                                    continue
                                cyc = group_cycles_dw((Bp + k) >> 2, vp).sum()
                                act[j] += cls[j][k] * cyc
                                issued[j] += cls[j][k] * Bp.shape[0]
                            keep = d[q[:, 0], q[:, 1]] > j
                            queues[wv] = q[keep]
    return act, issued


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "lanes":
        nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
        for assign in ("block", "interleave"):
            print(f"--- stage-0 passes dealt out: {assign}")
            lane_model(synth.mixed_batch(nframes, 320, 240, seed0=1234), load_cascade(), assign=assign)
        return
    nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (320, 240)
    c = load_cascade()
    frames = synth.mixed_batch(nframes, w, h, seed0=1234)
    for scheme, assign in (("rowmajor", "block"), ("strip4", "block"), ("strip4", "interleave"), ("strip12", "block")):
        act, issued = simulate(frames, c, scheme, assign)
        print(f"{scheme:9s} {assign:10s}: LDS cycles {act.sum():.3e} for {issued.sum():.3e} conflict-free -> conflict share {100 * (1 - issued.sum() / act.sum()):.1f} % of all cycles")
        print("    per stage cycles / conflict-free:", " ".join(f"{a / max(b, 1):.2f}" for a, b in zip(act, issued)), "| share of cycles:", " ".join(f"{100 * a / act.sum():.0f}%" for a in act))




def lane_model(frames, c, scheme="strip4", assign="block"):
    """Wave passes of the stages after stage 0 as the kernel runs them (every wavefront walks its own queue in chunks of 64; once <= 64
    windows are left in the tile all four wavefronts take a quarter of the stage's features for all of them = one pass in all) against
    two alternatives: the queues balanced over the wavefronts whenever 64 < survivors <= 256, and two wavefronts per tile.
    Weighted with the stage's sample count (~ its VALU + LDS instructions per pass)."""
    pts = []
    for j in range(NST):
        st = c.stages[j]
        n = 0
        for k in range(int(st["count"])):
            f = c.features[int(st["first"]) + k]
            n += sum(1 for t in range(int(f["size"])) if f["pz"][t] >= 0) + sum(1 for t in range(int(f["size"])) if f["nz"][t] >= 0)
        pts.append(n)
    nfeat = [int(c.stages[j]["count"]) for j in range(NST)]
    FP_PASS = 16  # sample-units of one feature-parallel pass: 64 (window, feature) lanes x (10 padded points + the per-lane record)
    fpar = np.zeros(NST)   # as it runs, but a tile with few survivors evaluates (window, feature) pairs across the lanes when that is cheaper
    both = np.zeros(NST)   # ... and the queues packed into full wavefronts for 64 < T <= 256
    hist_T = np.zeros((NST, 8), dtype=np.int64)  # tiles by survivor count at stage entry: 1-4, 5-8, 9-16, 17-32, 33-64, 65-128, 129-256, > 256
    now = np.zeros(NST)
    bal = np.zeros(NST)
    two = np.zeros(NST)
    ideal = np.zeros(NST)
    for fr in frames:
        levels, arena = ho.pyramid(fr)
        for i in range(len(levels) - 12):
            death = s1.survivors_for_scale(c, levels, arena, i)
            if death is None:
                continue
            H2, W2 = death.shape
            tw2, th2, ntx, nty = tile_shapes(W2, H2)
            for ty in range(nty):
                for tx in range(ntx):
                    X0, Y0 = tx * tw2, ty * th2
                    tw, th = min(tw2, W2 - X0), min(th2, H2 - Y0)
                    d = death[Y0:Y0 + th, X0:X0 + tw]
                    row, p, v0, v1 = enum_pairs(scheme, tw2, th, tw)
                    npairs = len(row)
                    nbat = -(-npairs // 32)
                    for nw, acc in ((4, None), (2, two)):
                        cnt = np.zeros((nw, NST + 1), dtype=np.int64)  # windows of wavefront w alive at the entry of stage j
                        for wv in range(nw):
                            sel = np.concatenate([np.arange(b, e) for b, e in wave_pair_ranges(npairs, wv, nw, assign)] + [np.zeros(0, dtype=np.int64)]).astype(np.int64)
                            r_, p_ = row[sel], p[sel]
                            for a, dx in ((v0[sel], 0), (v1[sel], 1)):
                                dd = d[np.minimum(r_, th - 1), np.minimum(2 * p_ + dx, tw - 1)][a]
                                for j in range(1, NST):
                                    cnt[wv, j] += int((dd >= j).sum())
                        for j in range(1, NST):
                            T = int(cnt[:, j].sum())
                            if T == 0:
                                break
                            if nw == 4:
                                ideal[j] += pts[j] * T / 64.0
                                passes = 1 if T <= 64 else int(sum(-(-int(x) // 64) for x in cnt[:, j]))
                                now[j] += pts[j] * passes
                                bal[j] += pts[j] * (1 if T <= 64 else (-(-T // 64) if T <= 256 else passes))
                                fp = -(-T * nfeat[j] // 64) * FP_PASS
                                fpar[j] += min(pts[j] * passes, fp) if T <= 64 else pts[j] * passes
                                both[j] += min(pts[j], fp) if T <= 64 else pts[j] * (-(-T // 64) if T <= 256 else passes)
                                hist_T[j, min(7, max(0, int(np.ceil(np.log2(max(T, 1) / 4.0 + 1e-9))) if T > 4 else 0))] += 1
                            else:
                                acc[j] += pts[j] * (1 if T <= 64 else int(sum(-(-int(x) // 64) for x in cnt[:, j])))
    print("stage:                ", " ".join(f"{j:8d}" for j in range(1, NST)))
    print("tiles by survivors at stage entry (1-4, 5-8, 9-16, 17-32, 33-64, 65-128, 129-256, > 256):")
    for j in range(1, NST):
        print(f"  stage {j}:", " ".join(f"{int(x):6d}" for x in hist_T[j]))
    for name, v in (("as it runs", now), ("balanced 64<T<=256", bal), ("two wavefronts/tile", two), ("feature-parallel sparse", fpar), ("packed + feature-par.", both),
                    ("full lanes", ideal)):
        print(f"{name:22s}", " ".join(f"{x / 1e3:8.0f}" for x in v[1:]), f"  total {v.sum() / 1e3:.0f} k sample-passes ({100 * v.sum() / now.sum():.0f} %)")


if __name__ == "__main__":
    main()
