#!/usr/bin/env python3
"""tools/model_scan_derive.py [W H N] — arithmetic for round-5 verdict item 2: "derive what only the scan reads inside the scan".

Levels >= upto and all 81 variant canvases are never a plane 0: they exist only as plane 1 / plane 2 of some scale.  k_scan_tiles holds
a scale's plane-0 tile in LDS, so its plane-1 cells (level i + 6 = resample of level i) and the four plane-2 variants (level i + 12 =
resample of level i + 6 shifted by (dx, dy)) could be built during staging and the pyramid would stop building them.  This script
counts, with the library's own geometry and tile plan (ht_context.hip set_geometry_impl, ht_scan.hip ht_scan_plan_tiles restated), the
pixels either side and prices them with the measured instruction figures of the round (PMC, wave instructions):

  * pyramid pixels that would no longer be built (variants + levels that are never a plane 0), at k_resample_bands' measured VALU
    wave-instructions per destination pixel (launch total / pixels written), plus the tail kernel's share of them;
  * pixels the tile kernel would have to derive, per tile, INCLUDING the halo overlap between neighbouring tiles (a tile of tw x th
    half-steps holds (tw + 11) x (th + 11) plane-1 cells and as many plane-2 cells spread over the four variants), at the resampler's
    instruction cost per pixel in its straight-line inner loop (57 VALU per 4 pixels: tools/disasm.py k_resample_bands).

Two variants of the idea: (a) derive plane 1 AND plane 2 from plane 0 (nothing but level i is read), (b) keep reading plane 1 from HBM
and derive only the four variants from it (the only planes the pyramid could then really drop: plane 1 of scale i is plane 0 of scale
i + 6 and has to exist anyway for i + 6 < upto)."""
import math
import sys

W, H, N = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (320, 240, 256)
INTERVAL, NEXT, CW = 5, 6, 24
TXH, TYH = 64, 32
# measured this round (gpurun_out/pmc_ab.txt, profiles/r06_*_pmc_per_launch.csv): VALU wave instructions per launch
VALU_RESAMPLE = {(320, 240, 256): 3 * 10.091e6, (1280, 720, 128): 5 * 30.785e6}
VALU_TAIL = {(320, 240, 256): 5.428e6, (1280, 720, 128): 1.874e6}
VALU_TILES = {(320, 240, 256): 47.426e6, (1280, 720, 128): 274.115e6}
INNER_VALU_PER_4PX = 57.0

scale = 2.0 ** (1.0 / (INTERVAL + 1))
levels = []
i = 0
upto = int(math.floor(math.log(CW) / math.log(scale)))  # ccv.js:111: from the cascade's size, not the frame's
nlev = upto + 2 * NEXT
for i in range(nlev):
    if i == 0:
        levels.append((W, H))
    elif i <= INTERVAL:
        levels.append((int(W / scale ** i), int(H / scale ** i)))
    else:
        w, h = levels[i - NEXT]
        levels.append((w // 2, h // 2))
P = sum(w * h for w, h in levels) + 3 * sum(w * h for w, h in levels[2 * NEXT:])
px_variants = 3 * sum(w * h for w, h in levels[2 * NEXT:])
px_never_p0 = sum(w * h for w, h in levels[upto:])
px_derived_total = P - W * H
tail_px = 0  # pixels of the generations the tail kernel builds: its cap is 32 768 destination pixels per frame, last generations first
gen = [0 if i == 0 else (1 if i <= INTERVAL else None) for i in range(nlev)]
for i in range(NEXT, nlev):
    gen[i] = gen[i - NEXT] + 1
ngen = max(gen) + 1
gpx = [0] * ngen
for i in range(1, nlev):
    gpx[gen[i]] += levels[i][0] * levels[i][1] * (4 if i >= 2 * NEXT else 1)
tail_gens = []
acc = 0
for g in range(ngen - 1, 0, -1):
    if acc + gpx[g] > 32768:
        break
    acc += gpx[g]
    tail_gens.append(g)
tail_px = acc
tail_drop = sum(levels[i][0] * levels[i][1] * (3 if i < upto else 4) for i in range(2 * NEXT, nlev) if gen[i] in tail_gens)
tiles = 0
cells = 0
for i in range(upto):
    qw, qh = levels[i + 2 * NEXT][0] - CW // 4, levels[i + 2 * NEXT][1] - CW // 4
    if qw <= 0 or qh <= 0:
        continue
    ntx = (2 * qw + TXH - 1) // TXH
    tw2 = ((2 * qw + ntx - 1) // ntx + 7) & ~7
    nty = (2 * qh + TYH - 1) // TYH
    th2 = (2 * qh + nty - 1) // nty
    th2 += th2 & 1
    ntx, nty = (2 * qw + tw2 - 1) // tw2, (2 * qh + th2 - 1) // th2
    for y in range(nty):
        for x in range(ntx):
            tw, th = min(tw2, 2 * qw - x * tw2), min(th2, 2 * qh - y * th2)
            tiles += 1
            cells += (tw + 11) * (th + 11)
key = (W, H, N)
print(f"{W}x{H}, {N} frames: {nlev} levels, upto {upto}, P = {P} px, derived {px_derived_total} px per frame, {tiles} tiles per frame")
print(f"  never a plane 0: variants {px_variants} px + levels >= upto {px_never_p0} px = {(px_variants + px_never_p0) / P * 100:.1f} % of P;"
      f" tail kernel builds {tail_px} px of which {tail_drop} would go")
print(f"  plane-1 / plane-2 cells held by the tiles of a frame: {cells} each ({cells / max(sum(levels[i + NEXT][0] * levels[i + NEXT][1] for i in range(upto)), 1):.2f} x the plane-1 pixels: halo overlap)")
if key in VALU_RESAMPLE:
    valu_per_px = VALU_RESAMPLE[key] / ((px_derived_total - tail_px) * N)
    drop = px_variants + px_never_p0
    saved = (drop - tail_drop) * N * valu_per_px + VALU_TAIL[key] * tail_drop / max(tail_px, 1)
    cost_a = 2 * cells * N * INNER_VALU_PER_4PX / 4 / 64
    cost_b = cells * N * INNER_VALU_PER_4PX / 4 / 64
    print(f"  k_resample_bands: {valu_per_px * 64:.1f} VALU lane-instructions per destination pixel (launch totals / pixels); inner loop {INNER_VALU_PER_4PX / 4:.1f}")
    print(f"  pyramid VALU saved per step: {saved / 1e6:.1f} M wave instructions")
    print(f"  (a) derive plane 1 + plane 2 in the tile kernel: + {cost_a / 1e6:.1f} M  ({cost_a / VALU_TILES[key] * 100:.0f} % of k_scan_tiles' {VALU_TILES[key] / 1e6:.1f} M)  -> net {(cost_a - saved) / 1e6:+.1f} M")
    print(f"  (b) derive only the four variants (plane 1 still read): + {cost_b / 1e6:.1f} M  -> net {(cost_b - saved) / 1e6:+.1f} M, before the halo / odd-parent"
          " fallback code and its registers (the kernel holds 79 of 80 VGPRs)")
