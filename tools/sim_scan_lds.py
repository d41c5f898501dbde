#!/usr/bin/env python3
"""tools/sim_scan_lds.py — CPU model of k_scan_tiles' LDS traffic: bank-conflict cycles per cascade stage for a given tile
layout, on real survivor sets (the oracle's pyramid + a numpy evaluation of the first stages).  ds_read_u8 is serviced in two
groups of 32 lanes; a group takes max(#distinct dwords per bank) cycles (MI355X_MICROARCH.md, LDS).

    python tools/sim_scan_lds.py [nframes] [w h]
"""
import sys
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from headtrackr_amd import synth  # noqa: E402
from headtrackr_amd.cascade import load_cascade  # noqa: E402
from oracle import ht_oracle as ho  # noqa: E402

NST = 8
TXH, TYH, NT = 64, 32, 256


def stage_fires(c, planes, j):
    """planes: (p0, p1, p2q) arrays for one scale & all 4 phases handled by caller; returns pass mask for stage j"""
    raise NotImplementedError


def survivors_for_scale(c, levels, arena, i):
    """death stage (0..NST, NST = survived all simulated stages) for every half-step window (Y', X') of scale i"""
    w0, h0, off0 = levels[i]
    w1, h1, off1 = levels[i + 6]
    w2, h2, off2 = levels[i + 12]
    qw, qh = w2 - 6, h2 - 6
    if qw <= 0 or qh <= 0:
        return None
    P0 = arena[off0[0]:off0[0] + w0 * h0].reshape(h0, w0).astype(np.int32)
    P1 = arena[off1[0]:off1[0] + w1 * h1].reshape(h1, w1).astype(np.int32)
    P2 = [arena[off2[q]:off2[q] + w2 * h2].reshape(h2, w2).astype(np.int32) for q in range(4)]
    H2, W2 = 2 * qh, 2 * qw
    Yp, Xp = np.meshgrid(np.arange(H2), np.arange(W2), indexing="ij")
    q = (Yp & 1) * 2 + (Xp & 1)
    alive = np.ones((H2, W2), dtype=bool)
    death = np.full((H2, W2), NST, dtype=np.int32)

    def px(z, x, y):
        if z == 0:
            return P0[np.minimum(2 * Yp + y, h0 - 1), np.minimum(2 * Xp + x, w0 - 1)]
        if z == 1:
            return P1[np.minimum(Yp + y, h1 - 1), np.minimum(Xp + x, w1 - 1)]
        out = np.zeros((H2, W2), dtype=np.int32)
        for qq in range(4):
            m = q == qq
            out[m] = P2[qq][np.minimum((Yp >> 1) + y, h2 - 1), np.minimum((Xp >> 1) + x, w2 - 1)][m]
        return out

    for j in range(NST):
        st = c.stages[j]
        s = np.zeros((H2, W2))
        for k in range(int(st["count"])):
            f = c.features[int(st["first"]) + k]
            pmin = np.full((H2, W2), 255, dtype=np.int32)
            nmax = np.zeros((H2, W2), dtype=np.int32)
            for t in range(int(f["size"])):
                if f["pz"][t] >= 0:
                    pmin = np.minimum(pmin, px(int(f["pz"][t]), int(f["px"][t]), int(f["py"][t])))
                if f["nz"][t] >= 0:
                    nmax = np.maximum(nmax, px(int(f["nz"][t]), int(f["nx"][t]), int(f["ny"][t])))
            s += np.where(pmin > nmax, float(f["alpha"][1]), float(f["alpha"][0]))
        dead = alive & (s < float(st["threshold"]))
        death[dead] = j
        alive &= ~dead
    return death


def stage_offsets(c, j, PITCH0, P12_BASE):
    st = c.stages[j]
    offs = {}
    for k in range(int(st["count"])):
        f = c.features[int(st["first"]) + k]
        for xs, ys, zs in ((f["px"], f["py"], f["pz"]), (f["nx"], f["ny"], f["nz"])):
            for t in range(int(f["size"])):
                z = int(zs[t])
                if z < 0:
                    continue
                x, y = int(xs[t]), int(ys[t])
                o = y * PITCH0 + x if z == 0 else (P12_BASE + y * 2 * PITCH0 + 2 * x if z == 1 else P12_BASE + 1 + 4 * y * PITCH0 + 4 * x)
                offs[(z, x, y)] = o
    return list(offs.values())


def group_cycles(addr, valid):
    """addr: [ngroups, 32] byte addresses (invalid lanes read address of lane base 0 -> we model them as inactive)"""
    dw = addr >> 2
    bank = dw & 31
    cyc = np.zeros(addr.shape[0], dtype=np.int64)
    for g in range(addr.shape[0]):
        v = valid[g]
        if not v.any():
            cyc[g] = 1
            continue
        pairs = np.unique(np.stack([bank[g][v], dw[g][v]], axis=1), axis=0)
        cyc[g] = np.bincount(pairs[:, 0], minlength=32).max()
    return cyc


def simulate(frames, c, PITCH0, mode="wrap"):
    ROWS0 = 2 * TYH + 22
    P12_BASE = PITCH0 * ROWS0
    offs = [np.array(stage_offsets(c, j, PITCH0, P12_BASE)) for j in range(NST)]
    ideal = np.zeros(NST)
    actual = np.zeros(NST)
    instr = np.zeros(NST)
    for fr in frames:
        levels, arena = ho.pyramid(fr)
        for i in range(len(levels) - 12):
            death = survivors_for_scale(c, levels, arena, i)
            if death is None:
                continue
            H2, W2 = death.shape
            ntx = -(-W2 // TXH)
            tw2 = (-(-W2 // ntx) + 7) & ~7
            nty = -(-H2 // TYH)
            th2 = -(-H2 // nty)
            th2 += th2 & 1
            ntx, nty = -(-W2 // tw2), -(-H2 // th2)
            for ty in range(nty):
                for tx in range(ntx):
                    X0, Y0 = tx * tw2, ty * th2
                    tw, th = min(tw2, W2 - X0), min(th2, H2 - Y0)
                    d = death[Y0:Y0 + th, X0:X0 + tw]
                    # stage 0 enumeration
                    stride = tw2 if mode == "wrap" else 64
                    n_in = stride * th
                    ids = np.arange(n_in)
                    yy, xx = ids // stride, ids % stride
                    valid = xx < tw
                    cur_ids = [(yy, xx, valid)]
                    for j in range(NST):
                        yy, xx, valid = cur_ids[-1]
                        n = len(yy)
                        if n == 0:
                            break
                        B = 2 * (yy * PITCH0 + xx)
                        B = np.where(valid, B, 0)
                        # lanes: position p -> wave/lane.  Stage 0/1 x2 loops interleave, modelled as plain order (same groups of 32)
                        pad = (-n) % 32
                        Bp = np.concatenate([B, np.zeros(pad, dtype=B.dtype)])
                        vp = np.concatenate([valid, np.zeros(pad, dtype=bool)])
                        Bg = Bp.reshape(-1, 32)
                        vg = vp.reshape(-1, 32)
                        # sparse phase (n <= 64): 4 waves each a quarter of the offsets -> same total group-instructions
                        # idle waves in the general loop (256-thread passes): groups with no valid lane still issue
                        ngroups_issued = (-(-n // NT)) * (NT // 32) if n > 64 else Bg.shape[0]
                        for o in offs[j]:
                            cyc = group_cycles(Bg + o, vg)
                            actual[j] += cyc.sum() + (ngroups_issued - Bg.shape[0])
                            ideal[j] += vg.any(axis=1).sum()
                            instr[j] += ngroups_issued
                        # survivors keep enumeration order (compaction by ballot/prefix in wave order)
                        keep = valid & (d[np.minimum(yy, th - 1), np.minimum(xx, tw - 1)] > j)
                        cur_ids.append((yy[keep], xx[keep], np.ones(int(keep.sum()), dtype=bool)))
    return ideal, actual, instr


def main():
    nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (320, 240)
    c = load_cascade()
    frames = synth.mixed_batch(nframes, w, h, seed0=1234)
    for PITCH0, mode in ((160, "wrap"), (160, "rows64"), (168, "wrap"), (176, "wrap"), (164, "wrap"), (162, "wrap")):
        ideal, actual, instr = simulate(frames, c, PITCH0, mode)
        print(f"PITCH0={PITCH0} {mode}: total cycles {actual.sum():.3e} (ideal {ideal.sum():.3e}, issued group-instr {instr.sum():.3e}) conflict overhead {100 * (actual.sum() / instr.sum() - 1):.1f} %")
        print("   per stage actual/issued:", " ".join(f"{a / max(b, 1):.2f}" for a, b in zip(actual, instr)), "| share of cycles:", " ".join(f"{100 * a / actual.sum():.0f}%" for a in actual))


if __name__ == "__main__":
    main()
