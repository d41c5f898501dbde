// ht_scan.hip — the BBF cascade scan of ccv.detect_objects (reference: /root/reference/src/ccv.js:150-246) for gfx950.
//
// What the reference does per pyramid scale i and half-pixel phase q=(dx,dy): slide a 24x24 window over level i in
// steps of 4 px (origin (4x+2dx, 4y+2dy)); a feature compares single pixels taken from three planes — level i at full
// window resolution, level i+6 at half, level i+12 (variant q) at quarter resolution — and fires iff
// min(positive pixels) > max(negative pixels) (ccv.js:189-220, the early exits there never change the outcome);
// a stage sums alpha[2k + fired_k] sequentially in binary64 and rejects the window if sum < threshold (ccv.js:222).
//
// MI355X mapping (no MFMA: this is byte gathers + compares, LDS-issue bound):
//   * k_scan_tiles — one workgroup per (frame, scale, tile).  The 4 phases are just the half-step grid
//     (X', Y') = (2x+dx, 2y+dy), so a tile is a dense rectangle of X' x Y' windows.  The three planes are staged in
//     LDS in a "unified-base" layout: plane 0 at 1 B/px with row pitch P, plane 1 at 2 B/px with row pitch 2P, and the
//     four quarter-res variants interleaved into one half-step grid in the odd bytes of the plane-1 cells.  Then
//     every window has ONE base address B = 2*(Y'*P + X') and every feature point is base + a per-feature constant,
//     so the inner loop is {s_load feature (uniform), v_add, ds_read_u8, v_min/v_max}; adjacent lanes (adjacent X')
//     read LDS at a 2-byte stride on every plane: conflict-free.  Windows that pass a stage are compacted so later stages
//     run on dense lanes.  Built-in cascade: the first 8 stages are generated straight-line code with exact integer decisions
//     and run "wave-private" (every wavefront owns the survivors of its windows: no atomics, no barriers; see the comment in
//     the kernel); other cascades and later stages: table-driven, shared queue, per-lane sequential binary64 sums in the
//     reference's order.  Both are bit-exact.
//   * k_scan_deep — survivors of the first `split` stages (a fraction of a percent of all windows) go through a
//     global queue to one wavefront per window with the stage's features spread across lanes.  Stage decisions use
//     exact integer sums of alpha*1e8 (the trained alphas/thresholds are 8-digit decimals; a different summation
//     order cannot change an integer sum), falling back to the sequential binary64 sum on an exact tie and for the
//     reported confidence of full survivors.
//   * k_scan_simple — one thread per window straight from HBM; slow, kept as an independent cross-check.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "ht_internal.h"

namespace {

// tile geometry (window = 24x24, see ht_scan_tile_tables for the check)
constexpr int TXH = 64;                 // tile width  in half-window steps X'
#ifndef HT_TILE_TYH
#define HT_TILE_TYH 32
#endif
#ifndef HT_TILE_WPS
#define HT_TILE_WPS 6  // waves per SIMD the register allocator must leave room for: 6 workgroups per CU (80 VGPRs, no spills; 26.4 KB of LDS
                       // each now that the survivor queue lives in the unused tails of the plane-1/2 rows).  Measured: 4 -> 5 workgroups
                       // -15 %, 5 -> 6 another -7 % (C2) / -6 % (C4) on the tile kernel.
#endif
constexpr int TYH = HT_TILE_TYH;        // tile height in half-window steps Y'
#ifndef HT_TILE_NT
#define HT_TILE_NT 256
#endif
constexpr int NT = HT_TILE_NT;          // threads per workgroup (256: 4 waves/SIMD at <=128 VGPRs; 512: 8 waves/SIMD at <=64)
#ifndef HT_TILE_PITCH
#define HT_TILE_PITCH 152
#endif
// plane-0 bytes per LDS row: 2*TXH + 24 = 152 are needed.  A window's base address steps by 2*PITCH0 bytes per half-step row, i.e.
// PITCH0/2 dwords: 152 -> 76 = 12 (mod 32 banks), so 8 consecutive rows start in 8 different 4-bank slots (0,12,24,4,16,28,8,20).
// With 160 the step was 16 (mod 32): rows r and r+2 aliased, and compacted survivors — which cluster in 2-D blobs — hit the
// same banks from every other row (36 % of the kernel's LDS cycles were bank conflicts; tools/sim_scan_lds.py models
// 61 % -> 44 % overhead over conflict-free).  Multiple of 8: rows are staged as 16-byte loads split into two 8-byte LDS writes.
constexpr int PITCH0 = HT_TILE_PITCH;
static_assert(PITCH0 >= 2 * TXH + 24 && PITCH0 % 8 == 0, "PITCH0");
constexpr int ROWS0 = 2 * TYH + 22;     // 86
constexpr int P0_BYTES = PITCH0 * ROWS0;  // 13760
constexpr int GH = TYH + 11;               // plane-1 / plane-2 half-step grid: 75 x 43 cells of 2 bytes
constexpr int G_PITCH = 2 * PITCH0;          // 320
constexpr int P12_BASE = P0_BYTES;
constexpr int LDS_TILE_BYTES = P0_BYTES + GH * G_PITCH;  // 27520
constexpr int MAXWIN = TXH * TYH;            // 2048 windows per tile
static_assert(P0_BYTES % 16 == 0, "alignment");

// unified-base LDS offsets of a feature point (x, y) on plane 0 / 1 / 2, relative to the window base B
#define HT_O0(x, y) ((y) * PITCH0 + (x))                      // level i:        1 B/px, row pitch P
#define HT_O1(x, y) (P12_BASE + (y) * G_PITCH + 2 * (x))      // level i+6:      2 B/px, row pitch 2P
#define HT_O2(x, y) (P12_BASE + 1 + 4 * (y) * PITCH0 + 4 * (x))  // level i+12 q: odd bytes, 4 B/px, row pitch 4P

#include "ht_cascade_gen.inc"  // straight-line code for the first HT_GEN_STAGES stages of the built-in cascade

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// offset j of a packed u16x2 table held in scalar registers
#define HT_OFF(tab, j) (((j) & 1) ? ((tab)[(j) >> 1] >> 16) : ((tab)[(j) >> 1] & 0xffffu))
#define HT_RD(tab, j) ((uint32_t)lds[B + HT_OFF(tab, j)])

// One stage for one window held by this lane; F/count are wave-uniform, so the feature record is fetched with scalar
// loads and every branch below is a scalar branch.  Sequential binary64 accumulation in feature order == ccv.js:186-221.
__device__ __forceinline__ double eval_stage_lds(const uint8_t *lds, uint32_t B, const HtTileFeature *__restrict__ F, uint32_t count) {
    double sum = 0.0;
    for (uint32_t k = 0; k < count; k++) {
        const uint4 P4 = *reinterpret_cast<const uint4 *>(F[k].po);
        const uint4 N4 = *reinterpret_cast<const uint4 *>(F[k].no);
        const uint4 A4 = *reinterpret_cast<const uint4 *>(F[k].a);
        const uint32_t np = F[k].np, nn = F[k].nn;
        const uint32_t po[4] = {P4.x, P4.y, P4.z, P4.w}, no[4] = {N4.x, N4.y, N4.z, N4.w};
        uint32_t pmin = HT_RD(po, 0), nmax = HT_RD(no, 0);
        if (np > 1) {
            pmin = min(pmin, HT_RD(po, 1));
            if (np > 2) {
                pmin = min(pmin, HT_RD(po, 2));
                if (np > 3) {
                    pmin = min(pmin, HT_RD(po, 3));
                    if (np > 4) {
                        pmin = min(pmin, HT_RD(po, 4));
                        if (np > 5) {
                            pmin = min(pmin, HT_RD(po, 5));
                            if (np > 6) {
                                pmin = min(pmin, HT_RD(po, 6));
                                if (np > 7) pmin = min(pmin, HT_RD(po, 7));
                            }
                        }
                    }
                }
            }
        }
        if (nn > 1) {
            nmax = max(nmax, HT_RD(no, 1));
            if (nn > 2) {
                nmax = max(nmax, HT_RD(no, 2));
                if (nn > 3) {
                    nmax = max(nmax, HT_RD(no, 3));
                    if (nn > 4) {
                        nmax = max(nmax, HT_RD(no, 4));
                        if (nn > 5) {
                            nmax = max(nmax, HT_RD(no, 5));
                            if (nn > 6) {
                                nmax = max(nmax, HT_RD(no, 6));
                                if (nn > 7) nmax = max(nmax, HT_RD(no, 7));
                            }
                        }
                    }
                }
            }
        }
        const bool fire = pmin > nmax;
        sum = __dadd_rn(sum, __hiloint2double((int)(fire ? A4.w : A4.y), (int)(fire ? A4.z : A4.x)));
    }
    return sum;
}

// -DHT_TILE_TIMELINE (tools/gpu_tile_timeline.py): shader-clock time a workgroup spends in each phase, summed per phase into the
// statistics rows (entries 32.. = cycles, 48.. = workgroups that went through the phase); needs HT_SCAN_STATS at run time
#ifdef HT_TILE_TIMELINE
#define TL_STAMP(p)                                                                              \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (my_stats && tid == 0) {                                                              \
            const unsigned long long now_ = __builtin_readcyclecounter();                        \
            atomicAdd(&my_stats[32 + (p)], now_ - tl_prev);                                      \
            atomicAdd(&my_stats[48 + (p)], 1ull);                                                \
            tl_prev = now_;                                                                      \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
#else
#define TL_STAMP(p)
#endif

template <bool GEN>
__global__ __launch_bounds__(NT, HT_TILE_WPS) void k_scan_tiles(const uint8_t *__restrict__ arena, uint64_t arena_stride,
                                                   const HtTileRec *__restrict__ tile_recs, const HtTileFeature *__restrict__ feats,
                                                   const HtPackedFeature *__restrict__ fp_feats, const HtDevStage *__restrict__ stages, int nstages, int split, uint32_t deep_bias,
                                                   int stop_stage, int force_exact, uint32_t tiles_per_frame, uint32_t total_tiles, HtQueueEntry *__restrict__ queue,
                                                   uint32_t queue_cap, ht_hit *__restrict__ hits, uint32_t hit_cap,
                                                   HtCounters *__restrict__ ctr, unsigned long long *__restrict__ stats) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_TILE_BYTES];
    // Survivor queue of window ids, compacted in place (a thread reads its entry, the workgroup synchronises, then survivors are
    // written to the front).  The unified-base layout gives the plane-1 / plane-2 rows a pitch of 2 * PITCH0 = 304 bytes of which only
    // the first 160 hold cells: the queue lives in the unused tail of those rows (64 u16 entries per row, 43 rows = 2752 >= MAXWIN)
    // instead of 4 KB of its own — 26.2 KB of LDS per workgroup = 6 workgroups per CU.
    static_assert(G_PITCH - 160 >= 128 && GH * 64 >= MAXWIN, "queue does not fit the row tails");
#define QB(qi_, i_) (*reinterpret_cast<uint16_t *>(&lds[P12_BASE + 160 + ((uint32_t)(i_) >> 6) * G_PITCH + (((uint32_t)(i_)&63u) << 1)]))  // qi_: always 0 (one queue)
    __shared__ uint32_t s_nout;
    __shared__ uint32_t s_qbase;
#ifndef HT_TILE_WAVEQ
#define HT_TILE_WAVEQ 1
#endif
#ifndef HT_TILE_FPAR
#define HT_TILE_FPAR 1  // feature-parallel sparse phase (0: always the four feature slices)
#endif
#ifndef HT_TILE_MERGE_FROM
#define HT_TILE_MERGE_FROM 2  // the first stage after which the wavefronts compare their survivor counts
#endif
#if HT_TILE_WAVEQ
    // Wave-private layout of the same row tails (see "wave-private cascade" below): rows [8w, 8w + 8) = wavefront w's queue of
    // window ids (a wavefront enumerates at most MAXWIN / 4 = 512 windows), rows 32-37 = three buffers of 64 per-survivor integer
    // stage sums (36 dwords of tail per row: 32 used), row 38 = the wavefronts' survivor counts (two parities x 4).
    constexpr int WQ_ROWS = MAXWIN / NT, WQ_CAP = WQ_ROWS * 64;
    static_assert(NT != 256 || GH >= 4 * WQ_ROWS + 11, "wave-private queues do not fit the row tails");
    // rows 39-42: per wavefront the LDS bases of the tile's surviving windows in rank order (feature-parallel sparse phase)
#define FPB(w_, i_) (*reinterpret_cast<uint16_t *>(&lds[P12_BASE + 160 + (4 * WQ_ROWS + 7 + (w_)) * G_PITCH + (((uint32_t)(i_)&63u) << 1)]))
#define QW(w_, e_) QB(0, (w_) * WQ_CAP + (e_))
#define SF(b_, l_) (*reinterpret_cast<uint32_t *>(&lds[P12_BASE + 160 + (4 * WQ_ROWS + 2 * (b_) + ((l_) >> 5)) * G_PITCH + (((l_)&31u) << 2)]))
#define SCNT(p_) (reinterpret_cast<uint32_t *>(&lds[P12_BASE + 160 + (4 * WQ_ROWS + 6) * G_PITCH + (p_) * 16]))
    constexpr bool WAVEQ = GEN && NT == 256;
#else
    constexpr bool WAVEQ = false;
#define QW(w_, e_) QB(0, 0)
#define SF(b_, l_) (*SCNT(0))
#define SCNT(p_) (&s_qbase)
#define FPB(w_, i_) (*reinterpret_cast<uint16_t *>(&s_qbase))
    constexpr int WQ_CAP = 0;
#endif

    // XCD-aware tile order: consecutive tiles (same frame / scale, shared halos) stay on one XCD's L2.
    const uint32_t nb = gridDim.x, chunk = nb >> 3;  // gridDim.x is a multiple of 8
    const uint32_t t = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
    if (blockIdx.x == 0 && threadIdx.x < HT_DEEP_CTRS) reinterpret_cast<uint32_t *>(queue + queue_cap)[threadIdx.x * 64u] = 0u;  // k_scan_deep_lds' work counters
    if (t >= total_tiles) return;
    const uint32_t frame = t / tiles_per_frame, lt = t - frame * tiles_per_frame;
    const HtTileRec R = tile_recs[lt];  // one 64-byte scalar load: everything about the tile
    const int X0 = (int)(R.origin & 0xffffu), Y0 = (int)(R.origin >> 16);  // tile origin in half-window steps
    const int tw = (int)(R.size & 0xffffu), th = (int)(R.size >> 16);
    const struct { int tw2; uint32_t div_magic; uint32_t l0; } S = {(int)(R.tw2_l0 & 0xffffu), R.div_magic, R.tw2_l0 >> 16};
    struct Plane { uint32_t off[4]; int stride, h; };
    const Plane L0 = {{R.off0, 0u, 0u, 0u}, (int)(R.sh0 & 0xffffu), (int)(R.sh0 >> 16)};
    const Plane L1 = {{R.off1, 0u, 0u, 0u}, (int)(R.sh1 & 0xffffu), (int)(R.sh1 >> 16)};
    const Plane L2 = {{R.off2[0], R.off2[1], R.off2[2], R.off2[3]}, (int)(R.sh2 & 0xffffu), (int)(R.sh2 >> 16)};
    const uint8_t *fbase = arena + (uint64_t)frame * arena_stride;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    // optional survival statistics: one of HT_STAT_SHARDS counter rows per workgroup (same-address atomics from 15k
    // workgroups serialise in L2 at ~90/us, which would dominate the kernel)
    unsigned long long *my_stats = stats ? stats + (size_t)(blockIdx.x & (HT_STAT_SHARDS - 1)) * 64 : nullptr;
#ifdef HT_TILE_TIMELINE
    unsigned long long tl_prev = __builtin_readcyclecounter();
#endif
#ifndef HT_TILE_PRIO
#define HT_TILE_PRIO 1  // measured: 1 -> scan_tiles -2.1 % (C2) / -2.9 % (C4); 2 (also the sparse stages at high priority): the same
#endif
    // wave priority: the short dependent chain that gets the tile's loads out runs ahead of other wavefronts' stage-0 bursts
    if (HT_TILE_PRIO) __builtin_amdgcn_s_setprio(3);

    // ---- stage the three planes into LDS --------------------------------------------------------------------
    // All global loads of a thread are issued before the first LDS write (fixed trip counts, predicated), so one
    // memory latency is paid per tile instead of one per loop iteration.  Wide accesses: 16-byte chunks of level i, and
    // per 8 half-step cells 8 bytes of level i+6 plus 4 bytes of each of the row's two variants (tile origins are
    // multiples of 8 half-steps) — 10 loads per thread instead of 25 dword-sized ones; staging was ~30 % of the kernel's
    // VALU instructions, most of it address arithmetic and predicates per load.  Chunks may run past a row's end into
    // the next row / plane of the same arena; those bytes land in LDS columns no window of the tile reads.
    {
        // plane 0: level i, origin (2*X0, 2*Y0), PITCH0 bytes per row.  Addresses are the frame's (wave-uniform) base + a 32-bit
        // per-thread offset, and a chunk that is not needed is neither loaded nor written (its LDS bytes keep whatever the
        // previous tile left there: no valid window reads them) — staging used to be 23 % of the kernel's VALU instructions,
        // half of them 64-bit address arithmetic and zero-initialisation of the predicated loads.
        const int gx0 = 2 * X0, gy0 = 2 * Y0;
        constexpr int C0 = (PITCH0 + 15) / 16;  // 16-byte chunks per row; with PITCH0 % 16 == 8 the last one is half a chunk
        const int n0 = (2 * th + 22) * C0;
        constexpr int K0 = (ROWS0 * C0 + NT - 1) / NT;
        uint4 v0[K0];
        bool ok0[K0];
#pragma unroll
        for (int k = 0; k < K0; k++) {
            const int i = (int)tid + k * NT;
            const int r = i / C0, c16 = (i - r * C0) * 16;
            const int gy = gy0 + r, gx = gx0 + c16;
            ok0[k] = i < n0 && gy < L0.h && gx < L0.stride;
            if (ok0[k]) v0[k] = *reinterpret_cast<const uint4 *>(fbase + (L0.off[0] + (uint32_t)__mul24(gy, L0.stride) + (uint32_t)gx));
        }
        // planes 1 + 2: half-step grid cell (X, Y) = { level i+6 pixel (X0+X, Y0+Y),  variant q pixel ((X0+X)>>1, (Y0+Y)>>1) }
        // with q = ((Y0+Y)&1)*2 + ((X0+X)&1)   (ccv.js:132-146: variant q is level i+6 shifted by (dx,dy) then halved).
        // A thread builds 8 consecutive cells from 8 bytes of level i+6 and 4 bytes of each of the two variants (coordinates
        // clamped into the variant's plane: cells beyond it belong to no window).
        constexpr int GG = (TXH + 11 + 7) / 8;  // cell groups per row (10)
        const int n12 = (th + 11) * GG;
        constexpr int K12 = (GH * GG + NT - 1) / NT;
        uint2 va[K12];
        uint32_t vb[K12], vc[K12];
        bool ok12[K12];
#pragma unroll
        for (int k = 0; k < K12; k++) {
            const int g = (int)tid + k * NT;
            const int Y = g / GG, Xg = (g - Y * GG) * 8;
            const int ay = Y0 + Y, ax = X0 + Xg, y2 = min(ay >> 1, L2.h - 1), x2 = min(ax >> 1, L2.stride - 4);
            const uint32_t o2a = (ay & 1) ? L2.off[2] : L2.off[0], o2b = (ay & 1) ? L2.off[3] : L2.off[1];
            ok12[k] = g < n12 && ay < L1.h && ax < L1.stride;
            if (ok12[k]) {
                const uint32_t o2 = (uint32_t)__mul24(y2, L2.stride) + (uint32_t)x2;
                va[k] = *reinterpret_cast<const uint2 *>(fbase + (L1.off[0] + (uint32_t)__mul24(ay, L1.stride) + (uint32_t)ax));
                vb[k] = *reinterpret_cast<const uint32_t *>(fbase + (o2a + o2));
                vc[k] = *reinterpret_cast<const uint32_t *>(fbase + (o2b + o2));
            }
        }
#pragma unroll
        for (int k = 0; k < K0; k++) {
            const int i = (int)tid + k * NT;
            if (ok0[k]) {
                if (PITCH0 % 16 == 0) {
                    *reinterpret_cast<uint4 *>(&lds[i * 16]) = v0[k];  // rows are contiguous: r*PITCH0 + c16 == i*16
                } else {  // rows are 8-byte aligned: two 8-byte halves, the second one dropped where it would spill into the next row
                    const int r = i / C0, c16 = (i - r * C0) * 16;
                    uint2 *d = reinterpret_cast<uint2 *>(&lds[r * PITCH0 + c16]);
                    d[0] = make_uint2(v0[k].x, v0[k].y);
                    if (c16 + 8 < PITCH0) d[1] = make_uint2(v0[k].z, v0[k].w);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < K12; k++) {
            const int g = (int)tid + k * NT;
            if (ok12[k]) {
                const int Y = g / GG, Xg = (g - Y * GG) * 8;
                const uint32_t b = vb[k], c = vc[k];
                // cell j = { a_j, (j even ? b : c)_{j/2} }: v_perm_b32 picks bytes from {src0 = bytes 4-7, src1 = bytes 0-3}
                uint4 w;
                w.x = __builtin_amdgcn_perm(c, __builtin_amdgcn_perm(b, va[k].x, 0x00010400u), 0x04020100u);  // a0 b0 a1 c0
                w.y = __builtin_amdgcn_perm(c, __builtin_amdgcn_perm(b, va[k].x, 0x00030502u), 0x05020100u);  // a2 b1 a3 c1
                w.z = __builtin_amdgcn_perm(c, __builtin_amdgcn_perm(b, va[k].y, 0x00010600u), 0x06020100u);  // a4 b2 a5 c2
                w.w = __builtin_amdgcn_perm(c, __builtin_amdgcn_perm(b, va[k].y, 0x00030702u), 0x07020100u);  // a6 b3 a7 c3
                *reinterpret_cast<uint4 *>(&lds[P12_BASE + Y * G_PITCH + 2 * Xg]) = w;
            }
        }
    }
    if (tid == 0) s_nout = 0;
    if (WAVEQ && tid < 192u) SF(tid >> 6, tid & 63u) = 0u;
    __syncthreads();
    if (HT_TILE_PRIO) __builtin_amdgcn_s_setprio(0);
    TL_STAMP(0);

    // ---- cascade with per-stage compaction -------------------------------------------------------------------
    // ---- wave-private cascade (built-in cascade) ----------------------------------------------------------------
    // Shader-clock stamps (tools/gpu_tile_timeline.py) showed every stage costing a workgroup ~3 000 cycles whatever its survivor
    // count: not arithmetic but the DEPENDENT LDS round trips of the shared-queue machinery (queue read, feature loads, queue
    // reservation by LDS atomic, queue write, counter read — each one waits behind the ~24 wavefronts' queued LDS traffic of the
    // CU) and 4-5 workgroup barriers.  Here a wavefront keeps the survivors of ITS windows in a queue of its own: compaction is
    // ballot + popcount in registers, no atomics and no barriers; per stage it pays two round trips (ids, feature pixels).
    // After stage HT_TILE_MERGE_FROM the four wavefronts publish their counts (one barrier); once <= 64 windows are left in the
    // tile every wavefront pulls all of them into registers (lane = survivor, the same in all four wavefronts) and the
    // stages are evaluated in four feature slices whose integer partial sums meet in LDS: one atomic add, one barrier and one
    // read per stage, no queue.  Window sets and stage decisions are exactly those of the shared-queue code below, which still
    // serves other cascades, the stages past the generated ones and the hand-off to k_scan_deep.
    uint32_t n_in = (uint32_t)(S.tw2 * th);  // stage 0 enumerates id = Y'*tw2 + X' (X' >= tw is masked off)
    uint32_t qoff = 0;                        // start of the live entries inside the queue
    bool pushed = (split >= nstages);
    int s_first = 0;
    const int wq_lim = min(min(split, nstages - 1), (int)HT_GEN_STAGES);  // stages [0, wq_lim) run wave-private
    if (WAVEQ && wq_lim >= 1) {
        const uint32_t wv = tid >> 6;
        if (stop_stage == 0) return;
        uint32_t wq = 0;  // survivors of this wavefront's windows (wave-uniform)
        {
            const HtDevStage st0 = stages[0];
#ifndef HT_TILE_BLOCKASSIGN
#define HT_TILE_BLOCKASSIGN 1
#endif
            // A wavefront takes a contiguous quarter of the tile's 64-window batches (= adjacent rows).  With the batches dealt out
            // round-robin its survivors came from every 4th row, whose LDS bank phases (12 banks per row) collide more often once
            // compacted: tools/sim_scan_lds.py counts 6.6 % fewer LDS cycles for the whole kernel with contiguous rows.
            const uint32_t nbat = (n_in + 63u) >> 6, per = HT_TILE_BLOCKASSIGN ? (nbat + 3u) >> 2 : 0u;
            const uint32_t b_lo = HT_TILE_BLOCKASSIGN ? wv * per : 0u, b_hi = HT_TILE_BLOCKASSIGN ? min(b_lo + per, nbat) : nbat;
#ifndef HT_TILE_PAIR
#define HT_TILE_PAIR 1
#endif
            // pair mode: a lane evaluates the two ADJACENT windows (X', X'+1) of a pass's 128 consecutive ids from shared aligned dword
            // reads (ht_gen_stage_0_pair: 30 LDS reads per pair instead of 2 x 22; stage 0 is LDS-throughput bound)
            constexpr bool PAIR = HT_TILE_PAIR && HT_TILE_BLOCKASSIGN && PITCH0 == HT_GEN_PAIR_PITCH0 && P12_BASE == HT_GEN_PAIR_P12;
            for (uint32_t bt = b_lo; bt < b_hi; bt += (HT_TILE_BLOCKASSIGN ? 2u : 8u)) {
                uint32_t id[2], xx[2], yy[2], Fv[2];
                bool valid[2];
                if (PAIR) {
                    // Order of the walk: window PAIRS in vertical strips of 4 pairs (8 half-steps), row by row inside a strip.  A pair's
                    // dword reads are `base + constant` with base / 4 = 76 * Y' + X' / 2, and a wave64 ds_read_b32 is served in two
                    // groups of 32 lanes over 32 banks: 8 consecutive rows x 4 consecutive pairs are 32 different banks (76 = 12 mod 32:
                    // the rows start at banks 0, 12, 24, 4, 16, 28, 8, 20), so every group is conflict-free.  Walking whole rows
                    // (pair n = row-major) put rows of tw2 / 2 < 32 pairs — all but 2 of the 19 scales at 320x240 — into groups that
                    // wrap to the next row 12 banks on: 2 cycles instead of 1 for nearly every one of stage 0's reads
                    // (tools/sim_scan_lds2.py: stage 0 at 1.86 LDS cycles per conflict-free cycle, 1.14 in strips; the PMC counters
                    // had 37 % of the kernel's LDS cycles as bank conflicts).  Window ids keep their meaning (Y' * tw2 + X').
                    const uint32_t lim = min(b_hi * 64u, n_in);
                    const uint32_t pn = bt * 32u + lane;  // pair index in walking order
                    const bool in0 = 2u * pn < lim;
                    const uint32_t pn0 = in0 ? pn : 0u;
#ifndef HT_TILE_STRIPS
#define HT_TILE_STRIPS 1
#endif
                    if (HT_TILE_STRIPS) {
                        const uint32_t strip = __umul24(pn0, R.strip_magic) >> 24, rem = pn0 - __umul24(strip, 4u * (uint32_t)th);
                        yy[0] = yy[1] = rem >> 2;
                        xx[0] = 8u * strip + 2u * (rem & 3u);
                        id[0] = __umul24(yy[0], (uint32_t)S.tw2) + xx[0];
                    } else {
                        id[0] = 2u * pn0;  // ids come in even / odd pairs of one row: tw2 is even
                        yy[0] = yy[1] = __umul24(id[0], S.div_magic) >> 20;
                        xx[0] = id[0] - __umul24(yy[0], (uint32_t)S.tw2);
                    }
                    id[1] = id[0] + 1u;
                    xx[1] = xx[0] + 1u;
                    valid[0] = in0 && xx[0] < (uint32_t)tw;
                    valid[1] = in0 && xx[1] < (uint32_t)tw;
#ifndef HT_TILE_PAIRPK
#define HT_TILE_PAIRPK 1  // both windows of the pair in the two 16-bit halves of one register (v_perm_b32 gather, v_pk_min/max_u16); 0: one 32-bit chain per window
#endif
                    if (HT_TILE_PAIRPK) ht_gen_stage_0_pair(lds + (valid[0] ? 2u * (yy[0] * PITCH0 + xx[0]) : 0u), Fv[0], Fv[1]);
                    else ht_gen_stage_0_pair_u32(lds + (valid[0] ? 2u * (yy[0] * PITCH0 + xx[0]) : 0u), Fv[0], Fv[1]);
                } else {
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint32_t bu = HT_TILE_BLOCKASSIGN ? bt + (uint32_t)u : bt + 4u * (uint32_t)u + wv;
                        const uint32_t pos = bu * 64u + lane;
                        valid[u] = bu < b_hi && pos < n_in;
                        id[u] = valid[u] ? pos : 0u;
                        yy[u] = __umul24(id[u], S.div_magic) >> 20;  // 24-bit multiplies (id < 2^11, magic < 2^18); v_mul_lo_u32 measures at the same 4 cycles per wave64 (tools/micro/valu_rate_bench.hip)
                        xx[u] = id[u] - __umul24(yy[u], (uint32_t)S.tw2);
                        valid[u] = valid[u] && xx[u] < (uint32_t)tw;
                    }
                    if (!HT_TILE_BLOCKASSIGN && bt + wv >= nbat) break;  // round-robin: this wavefront's batches are exhausted
                    ht_gen_stage_0_x2(lds + (valid[0] ? 2u * (yy[0] * PITCH0 + xx[0]) : 0u), lds + (valid[1] ? 2u * (yy[1] * PITCH0 + xx[1]) : 0u), Fv[0], Fv[1]);
                }
                bool pass[2];
                pass[0] = (Fv[0] >= HT_GEN_FMIN[0]) & valid[0];
                pass[1] = (Fv[1] >= HT_GEN_FMIN[0]) & valid[1];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (valid[u] && (Fv[u] == HT_GEN_FTIE[0] || force_exact))  // exact tie: the sequential binary64 sum decides
                        pass[u] = !(eval_stage_lds(lds, 2u * (yy[u] * PITCH0 + xx[u]), feats + st0.first, st0.count) < st0.threshold);
                }
                const unsigned long long m0 = __ballot(pass[0]), m1 = __ballot(pass[1]);
                const uint32_t c0 = __popcll(m0), c1 = __popcll(m1);
                if (pass[0]) QW(wv, wq + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))) = (uint16_t)id[0];
                if (pass[1]) QW(wv, wq + c0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))) = (uint16_t)id[1];
                wq += c0 + c1;
            }
            if (tid == 0 && my_stats) atomicAdd(&my_stats[0], (unsigned long long)(uint32_t)(tw * th));
        }
        // -- stages 1.. on the wavefront's own queue, until the tile is down to one wavefront of windows
        if (HT_TILE_PRIO == 3) __builtin_amdgcn_s_setprio(3);
        int s = 1;
        uint32_t total = 0;
        uint4 cn = make_uint4(0u, 0u, 0u, 0u);
        bool gather = false;
        for (;; s++) {
            if (s > HT_TILE_MERGE_FROM || s == wq_lim) {
                if (lane == 0) SCNT(s & 1)[wv] = wq;
                __syncthreads();
                cn = *reinterpret_cast<const uint4 *>(SCNT(s & 1));
                total = cn.x + cn.y + cn.z + cn.w;
                TL_STAMP(s);
                if (total == 0) return;
                if (total <= 64u && s < wq_lim) {
                    gather = true;
                    break;
                }
                if (s == wq_lim) break;
            } else {
                TL_STAMP(s);
            }
            if (s == stop_stage) return;
            if (lane == 0 && my_stats && wq) atomicAdd(&my_stats[s], (unsigned long long)wq);
            uint32_t out = 0;
            for (uint32_t b = 0; b < wq; b += 64) {
                const bool valid = b + lane < wq;
                const uint32_t wid = valid ? (uint32_t)QW(wv, b + lane) : 0u;
                const uint32_t yy = __umul24(wid, S.div_magic) >> 20, xx = wid - __umul24(yy, (uint32_t)S.tw2);
                const uint32_t Bw = 2u * (yy * PITCH0 + xx);
                const uint32_t Fv = ht_gen_stage(s, lds + (valid ? Bw : 0u));
                bool pass = (Fv >= HT_GEN_FMIN[s]) & valid;
                if (valid && (Fv == HT_GEN_FTIE[s] || force_exact)) {  // exact tie with the threshold: the sequential binary64 sum decides
                    const HtDevStage st = stages[s];
                    pass = !(eval_stage_lds(lds, Bw, feats + st.first, st.count) < st.threshold);
                }
                const unsigned long long m = __ballot(pass);
                // in place: position out + (survivors before this lane) <= b + lane, an entry this chunk has already read
                if (pass) QW(wv, out + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) = (uint16_t)wid;
                out += (uint32_t)__popcll(m);
            }
            wq = out;
        }
        if (HT_TILE_PRIO) __builtin_amdgcn_s_setprio(HT_TILE_PRIO > 1 ? 3 : 0);
        if (gather) {
            // -- <= 64 windows left in the tile: lane = survivor in every wavefront, stages in four feature slices
            const uint32_t o1 = cn.x, o2 = o1 + cn.y, o3 = o2 + cn.z;
            const uint32_t j = (lane >= o1 ? 1u : 0u) + (lane >= o2 ? 1u : 0u) + (lane >= o3 ? 1u : 0u);
            const uint32_t e = lane - (j == 0 ? 0u : (j == 1 ? o1 : (j == 2 ? o2 : o3)));
            bool alive = lane < total;
            const uint32_t wid = alive ? (uint32_t)QW(j, e) : 0u;
            const uint32_t yy = __umul24(wid, S.div_magic) >> 20, xx = wid - __umul24(yy, (uint32_t)S.tw2);
            const uint32_t Bw = 2u * (yy * PITCH0 + xx);
            // Feature-parallel sparse phase.  Few windows left (most tiles from stage 4 on hold 1 - 16): one lane per (window, feature)
            // pair instead of one lane per window — the stage's features spread over the workgroup's 256 lanes, each lane reading its
            // feature's offsets from a packed 32-byte record and adding its alpha to the window's integer sum in LDS when the feature
            // fires.  A lane = survivor pass costs a wavefront the stage's whole feature list whatever the survivor count (1 - 20 of
            // 64 lanes busy); the pair form costs ~48 instructions per 256 pairs (HT_GEN_FP_MAXPAIRS[s] is where the two meet,
            // tools/gen_cascade_code.py).  Same integer sums, same decisions.  Measured (round 5, LABLOG.md): -5.5 % VALU and -21 % LDS
            // instructions per launch, the kernel -2 ... -4 %; evaluating SEVERAL stages' features in one phase for <= 16 windows (fewer
            // latency chains, 3 - 4 x the pairs because most windows die in the first of them) was 8 - 12 % slower: removed.
            for (; s < wq_lim; s++) {
                if (s == stop_stage) return;
                const unsigned long long am = __ballot(alive);
                const uint32_t n_alive = (uint32_t)__popcll(am);
                if (tid == 0 && my_stats) atomicAdd(&my_stats[s], (unsigned long long)n_alive);
                const uint32_t bi = (uint32_t)s % 3u;
                const uint32_t nf = HT_GEN_NFEAT[s], npairs = n_alive * nf;
                const bool fpar = HT_TILE_FPAR && fp_feats != nullptr && npairs <= HT_GEN_FP_MAXPAIRS[s];
                uint32_t fslot = lane;  // the window's entry of the sum buffer: its lane, or its rank among the survivors
                if (fpar) {
                    fslot = __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
                    if (alive) FPB(wv, fslot) = (uint16_t)Bw;  // this wavefront's own copy: every wavefront holds every survivor
                    const HtPackedFeature *frec = fp_feats + HT_GEN_FIRST[s];
                    const uint32_t magic = HT_GEN_NFMAGIC[s];
                    for (uint32_t p0 = wv * 64u; p0 < npairs; p0 += 256u) {
                        const uint32_t p = p0 + lane;
                        const bool on = p < npairs;
                        const uint32_t pc = on ? p : 0u;
                        const uint32_t si = __umul24(pc, magic) >> 20, k = pc - __umul24(si, nf);  // pc < 1024, magic < 2^18: the product needs 28 bits
                        const uint4 r0 = reinterpret_cast<const uint4 *>(frec + k)[0];
                        uint4 r1 = reinterpret_cast<const uint4 *>(frec + k)[1];
                        asm volatile("" : "+v"(r1.x), "+v"(r1.z));  // both words with the first load: the alpha must not become a load of its own behind the decision
                        const uint8_t *wb = lds + (uint32_t)FPB(wv, si);
                        const uint32_t p0v = wb[r0.x & 0xffffu], p1v = wb[r0.x >> 16], p2v = wb[r0.y & 0xffffu], p3v = wb[r0.y >> 16], p4v = wb[r0.z & 0xffffu];
                        const uint32_t n0v = wb[r0.z >> 16], n1v = wb[r0.w & 0xffffu], n2v = wb[r0.w >> 16], n3v = wb[r1.x & 0xffffu], n4v = wb[r1.x >> 16];
                        const uint32_t pmin = min(min(min(p0v, p1v), min(p2v, p3v)), p4v), nmax = max(max(max(n0v, n1v), max(n2v, n3v)), n4v);
                        if (on && pmin > nmax) atomicAdd(&SF(bi, si), r1.z);  // r1.z = a1i = alpha[2k+1] * 1e8
                    }
                } else {
                    const uint32_t part = ht_gen_stage_slice(s, (int)wv, lds + (alive ? Bw : 0u));
                    if (alive && part) atomicAdd(&SF(bi, lane), part);
                }
                __syncthreads();
                const uint32_t Fv = SF(bi, fslot);
                // the buffer of stage s-1 (= s+2 mod 3): every wavefront read it before this stage's barrier, stage s+2 adds to it after the next one
                if (wv == 0) SF((uint32_t)(s + 2) % 3u, lane) = 0u;
                bool pass = (Fv >= HT_GEN_FMIN[s]) & alive;
                if (alive && (Fv == HT_GEN_FTIE[s] || force_exact)) {
                    const HtDevStage st = stages[s];
                    pass = !(eval_stage_lds(lds, Bw, feats + st.first, st.count) < st.threshold);
                }
                alive = pass;
                TL_STAMP(1 + s);
                if (!__ballot(alive)) return;  // the same decision in all four wavefronts
            }
            // the generated stages are done: the survivors go on through the shared queue (hand-off to the deep kernel, or the table-driven stages)
            const unsigned long long m = __ballot(alive);
            if (wv == 0 && alive) QB(0, __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) = (uint16_t)wid;
            n_in = (uint32_t)__popcll(m);
            __syncthreads();
        } else {
            // more than one wavefront of windows survived every generated stage (dense face texture): the four queues become the
            // shared one, wavefront by wavefront (a wavefront moves its entries down, never past the next wavefront's first entry)
            const uint32_t off = (wv == 0 ? 0u : (wv == 1 ? cn.x : (wv == 2 ? cn.x + cn.y : cn.x + cn.y + cn.z)));
            for (uint32_t w = 1; w < 4; w++) {
                __syncthreads();
                if (wv == w) {
                    for (uint32_t b = 0; b < wq; b += 64) {
                        const bool valid = b + lane < wq;
                        const uint32_t wid = valid ? (uint32_t)QW(w, b + lane) : 0u;
                        if (valid) QB(0, off + b + lane) = (uint16_t)wid;
                    }
                }
            }
            __syncthreads();
            n_in = total;
        }
        s_first = s;
    }
    // ---- shared-queue cascade, table-driven stages: other cascades from stage 0, the built-in one past its generated stages
    for (int s = s_first; s < nstages; s++) {
        if (s == stop_stage) return;  // measurement knob (option stop_stage): results are incomplete when set
        const HtDevStage st = stages[s];
        // hand-off rule: from stage `split` on, survivors leave for k_scan_deep (one wavefront per window, features
        // across lanes) as soon as that is cheaper than keeping them on a few lanes of this workgroup
        if (s >= split && !pushed && n_in * deep_bias * ((st.count + 63u) >> 6) <= st.count) {
            // hand the survivors of the first `split` stages to k_scan_deep through the global queue
            pushed = true;
            if (tid == 0) s_qbase = atomicAdd(&ctr->nqueue, n_in);
            __syncthreads();
            const uint32_t qb = s_qbase;
            const uint32_t room = qb < queue_cap ? queue_cap - qb : 0u;
            const uint32_t npush = min(n_in, room);
            // the plane offsets / strides for the entries: the tile record is read AGAIN here (one scalar load, only tiles with survivors get
            // this far) — kept alive from the staging code they cost 11 scalar registers spilled across the whole cascade
            uint32_t lt2 = lt;
            asm volatile("" : "+s"(lt2));
            const HtTileRec Rq = tile_recs[lt2];
            const uint32_t q_s0 = Rq.sh0 & 0xffffu, q_s1 = Rq.sh1 & 0xffffu, q_s2 = Rq.sh2 & 0xffffu;
            for (uint32_t i = tid; i < npush; i += NT) {
                const uint32_t id = QB(0, qoff + i);
                const uint32_t yy = __umul24(id, S.div_magic) >> 20, xx = id - __umul24(yy, (uint32_t)S.tw2);
                const uint32_t ax = (uint32_t)X0 + xx, ay = (uint32_t)Y0 + yy;
                // (built as two 16-byte words: as a struct with 1-, 2- and 4-byte members the entry went through scratch memory)
                const uint32_t q = ((ay & 1u) << 1) | (ax & 1u);
                uint4 w0, w1;
                w0.x = frame;
                w0.y = (ax >> 1) | (ay >> 1) << 16;                        // x, y
                w0.z = S.l0 | q << 8 | (uint32_t)s << 16;                  // scale, q, first stage the deep kernel has to run
                // window origins on the three planes (HT_WIN_SETUP's expressions with 4y + 2dy = 2 ay, 2y + dy = ay)
                w0.w = Rq.off0 + 2u * ay * q_s0 + 2u * ax;  // o0
                w1.x = Rq.off1 + ay * q_s1 + ax;            // o1
                w1.y = (q == 0 ? Rq.off2[0] : (q == 1 ? Rq.off2[1] : (q == 2 ? Rq.off2[2] : Rq.off2[3]))) + (ay >> 1) * q_s2 + (ax >> 1);  // o2
                w1.z = q_s0 | q_s1 << 16;                   // s0, s1
                w1.w = q_s2;                                // s2, pad
                uint4 *qe = reinterpret_cast<uint4 *>(queue + qb + i);
                qe[0] = w0, qe[1] = w1;
            }
            TL_STAMP(10);
            if (npush == n_in) return;  // the common case
            if (tid == 0) atomicAdd(&ctr->queue_inline, n_in - npush);
            qoff += npush;  // queue full: finish the remaining survivors right here
            n_in -= npush;
        }
        const HtTileFeature *F = feats + st.first;
        const bool last = (s == nstages - 1);
        for (uint32_t base = 0; base < n_in; base += NT) {
            const uint32_t pos = base + tid;
            bool valid = pos < n_in;
            uint32_t id = 0;
            if (valid) id = (s == 0) ? pos : (uint32_t)QB(0, qoff + pos);
            if (s > 0) __syncthreads();  // every entry of this chunk is in a register before survivors overwrite the queue
            // Survivors are compacted, so the waves behind the last survivor have no valid lane: they skip the stage body
            // (wave-uniform branch; the barrier above and the one after the loop are still hit by every wave).  With 65..255
            // survivors — the usual case in stages 2-3 — up to three of the four waves used to walk the whole stage on dead lanes,
            // 16 % of the kernel's LDS instructions (tools/sim_scan_lds.py).
            if (base + (tid & ~63u) >= n_in) continue;
            const uint32_t yy = __umul24(id, S.div_magic) >> 20, xx = id - __umul24(yy, (uint32_t)S.tw2);
            valid = valid && xx < (uint32_t)tw;
            const uint32_t B = 2u * (yy * PITCH0 + xx);
            const double sum = eval_stage_lds(lds, valid ? B : 0u, F, st.count);
            const bool pass = valid && !(sum < st.threshold);  // ccv.js:222
            const unsigned long long m = __ballot(pass);
            if (m) {
                const uint32_t cnt = __popcll(m);
                const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                uint32_t b0 = 0;
                if (!last) {
                    if (lane == 0) b0 = atomicAdd(&s_nout, cnt);
                    b0 = __builtin_amdgcn_readfirstlane(b0);
                    if (pass) QB(0, b0 + pre) = (uint16_t)id;
                } else {
                    if (lane == 0) b0 = atomicAdd(&ctr->nhits, cnt);
                    b0 = __builtin_amdgcn_readfirstlane(b0);
                    if (pass && b0 + pre < hit_cap) {
                        const uint32_t ax = (uint32_t)X0 + xx, ay = (uint32_t)Y0 + yy;
                        ht_hit h;
                        h.frame = frame;
                        h.x = (uint16_t)(ax >> 1);
                        h.y = (uint16_t)(ay >> 1);
                        h.scale = (uint8_t)S.l0;
                        h.q = (uint8_t)(((ay & 1u) << 1) | (ax & 1u));
                        h.reserved0 = 0;
                        h.reserved1 = 0;
                        h.sum = sum;  // ccv.js:233
                        hits[b0 + pre] = h;
                    }
                    if (lane == 0 && my_stats) atomicAdd(&my_stats[nstages], (unsigned long long)cnt);
                }
            }
        }
        if (tid == 0 && my_stats) atomicAdd(&my_stats[s], (unsigned long long)(s == 0 ? (uint32_t)(tw * th) : n_in));
        __syncthreads();
        n_in = s_nout;
        __syncthreads();
        if (tid == 0) s_nout = 0;
        qoff = 0;
        TL_STAMP(1 + min(s, 8));
        if (n_in == 0) return;
        // s_nout reset is ordered before its next use by the __syncthreads at the end of the next stage's loop body:
        // the next atomicAdd(&s_nout) can only come after every thread passed this point, but tid 0 might still be
        // about to write 0 -> make the reset visible first.
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------------------------------
// deep kernel: one wavefront per surviving window, features across lanes

// A window's three plane origins as 32-bit offsets from the frame's arena base, plus the row strides.  Plain scalars
// passed by value on purpose: a struct of arrays indexed by the per-lane plane number gets demoted to LDS scratch.
#define HT_WIN_ARGS const uint8_t *fb, uint32_t o0, uint32_t o1, uint32_t o2, int s0, int s1, int s2
#define HT_WIN_PASS fb, o0, o1, o2, s0, s1, s2
#define HT_WIN_SETUP(fbase_, levels_, next_, scale_, q_, x_, y_)                                                      \
    const uint8_t *fb = (fbase_);                                                                                     \
    uint32_t o0, o1, o2;                                                                                              \
    int s0, s1, s2;                                                                                                   \
    {                                                                                                                 \
        const HtDevLevel L0_ = (levels_)[(scale_)], L1_ = (levels_)[(scale_) + (next_)], L2_ = (levels_)[(scale_) + 2 * (next_)]; \
        const uint32_t dx_ = (q_)&1u, dy_ = (q_) >> 1;                                                                \
        const uint32_t v2_ = (q_) == 0 ? L2_.off[0] : ((q_) == 1 ? L2_.off[1] : ((q_) == 2 ? L2_.off[2] : L2_.off[3])); \
        s0 = L0_.stride, s1 = L1_.stride, s2 = L2_.stride;                                                            \
        o0 = L0_.off[0] + (4 * (y_) + 2 * dy_) * (uint32_t)s0 + (4 * (x_) + 2 * dx_); /* ccv.js:180,235 */            \
        o1 = L1_.off[0] + (2 * (y_) + dy_) * (uint32_t)s1 + (2 * (x_) + dx_);         /* ccv.js:180,236 */            \
        o2 = v2_ + (y_) * (uint32_t)s2 + (x_);                                        /* ccv.js:179,237 */            \
    }

__device__ __forceinline__ bool deep_fire(const HtDeepFeature *__restrict__ fp, HT_WIN_ARGS, uint32_t maxpts) {
    // six 8-byte coordinate arrays, one 64-bit load each; byte j = slot j (slots >= np/nn repeat slot 0)
    const unsigned long long *c = reinterpret_cast<const unsigned long long *>(fp);
    unsigned long long px = c[0], py = c[1], pz = c[2], nx = c[3], ny = c[4], nz = c[5];
    uint32_t pmin = 255u, nmax = 0u;
    for (uint32_t j = 0; j < maxpts; j++) {
        const uint32_t zp = (uint32_t)pz & 0xffu, zn = (uint32_t)nz & 0xffu;
        const uint32_t po = zp == 0 ? o0 : (zp == 1 ? o1 : o2), ps = (uint32_t)(zp == 0 ? s0 : (zp == 1 ? s1 : s2));
        const uint32_t no = zn == 0 ? o0 : (zn == 1 ? o1 : o2), ns = (uint32_t)(zn == 0 ? s0 : (zn == 1 ? s1 : s2));
        pmin = min(pmin, (uint32_t)fb[po + ((uint32_t)py & 0xffu) * ps + ((uint32_t)px & 0xffu)]);
        nmax = max(nmax, (uint32_t)fb[no + ((uint32_t)ny & 0xffu) * ns + ((uint32_t)nx & 0xffu)]);
        px >>= 8, py >>= 8, pz >>= 8, nx >>= 8, ny >>= 8, nz >>= 8;
    }
    return pmin > nmax;
}

__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// ---- deep kernel ------------------------------------------------------------------------------------------------
// Per-wavefront window patch in LDS: the 24x24 window of level i, its 12x12 counterpart on level i+6 and the 6x6 one on
// variant q of level i+12 = 756 bytes, loaded once per window; every later stage reads pixels from there.
constexpr int PATCH1 = 576, PATCH2 = 720, PATCH_BYTES = 768;
constexpr int DEEP_WAVES = 4;  // waves per workgroup

__device__ __forceinline__ bool patch_fire(const uint8_t *patch, const HtPatchFeature *__restrict__ fp, uint32_t maxpts) {
    const uint4 P = *reinterpret_cast<const uint4 *>(fp->poff);  // 8 x u16
    const uint4 N = *reinterpret_cast<const uint4 *>(fp->noff);
    const uint32_t pw[4] = {P.x, P.y, P.z, P.w}, nw[4] = {N.x, N.y, N.z, N.w};
    uint32_t pmin = 255u, nmax = 0u;
#pragma unroll
    for (uint32_t j = 0; j < HT_MAXPTS; j++) {
        if (j < maxpts) {  // maxpts is wave-uniform; slots >= np/nn repeat slot 0
            const uint32_t po = (j & 1) ? (pw[j >> 1] >> 16) : (pw[j >> 1] & 0xffffu);
            const uint32_t no = (j & 1) ? (nw[j >> 1] >> 16) : (nw[j >> 1] & 0xffffu);
            pmin = min(pmin, (uint32_t)patch[po]);
            nmax = max(nmax, (uint32_t)patch[no]);
        }
    }
    return pmin > nmax;
}

// sum += sel[0] + sel[1] + ... + sel[nn-1] in lane order, sequential binary64 adds (the reference's order).  Each lane holds
// the alpha its own feature selected; v_readlane broadcasts them one by one, so the only loop-carried dependency is the
// add itself (fetching the alphas from memory inside this chain cost ~50 us per full survivor).
__device__ __forceinline__ double seq_add_lanes(double sum, double sel, uint32_t nn) {
    const int lo = __double2loint(sel), hi = __double2hiint(sel);
    for (uint32_t t = 0; t < nn; t++) {
        const int l = __builtin_amdgcn_readlane(lo, (int)t), h = __builtin_amdgcn_readlane(hi, (int)t);
        sum = __dadd_rn(sum, __hiloint2double(h, l));
    }
    return sum;
}

// sequential binary64 stage sum in the reference's order: fire bits in parallel, adds in order (all lanes redundantly)
__device__ __forceinline__ double patch_stage_sum_exact(const uint8_t *patch, const HtPatchFeature *__restrict__ F, uint32_t count,
                                                        uint32_t maxpts, uint32_t lane) {
    double sum = 0.0;
    for (uint32_t kb = 0; kb < count; kb += 64) {
        const uint32_t k = kb + lane;
        double sel = 0.0;
        if (k < count) sel = patch_fire(patch, &F[k], maxpts) ? F[k].a1 : F[k].a0;
        sum = seq_add_lanes(sum, sel, min(64u, count - kb));
    }
    return sum;
}

// registers holding one lane's feature record (loaded one chunk ahead of its use)
struct PatchFeatRegs {
    uint4 P, N;        // 8 + 8 u16 offsets
    long long a0i, a1i;
};
__device__ __forceinline__ PatchFeatRegs load_feat(const HtPatchFeature *__restrict__ fp) {
    PatchFeatRegs r;
    r.P = *reinterpret_cast<const uint4 *>(fp->poff);
    r.N = *reinterpret_cast<const uint4 *>(fp->noff);
    const longlong2 A = *reinterpret_cast<const longlong2 *>(&fp->a0i);
    r.a0i = A.x;
    r.a1i = A.y;
    return r;
}
__device__ __forceinline__ bool fire_regs(const uint8_t *patch, const PatchFeatRegs &f, uint32_t maxpts) {
    const uint32_t pw[4] = {f.P.x, f.P.y, f.P.z, f.P.w}, nw[4] = {f.N.x, f.N.y, f.N.z, f.N.w};
    uint32_t pmin = 255u, nmax = 0u;
#pragma unroll
    for (uint32_t j = 0; j < HT_MAXPTS; j++) {
        if (j < maxpts) {
            const uint32_t po = (j & 1) ? (pw[j >> 1] >> 16) : (pw[j >> 1] & 0xffffu);
            const uint32_t no = (j & 1) ? (nw[j >> 1] >> 16) : (nw[j >> 1] & 0xffffu);
            pmin = min(pmin, (uint32_t)patch[po]);
            nmax = max(nmax, (uint32_t)patch[no]);
        }
    }
    return pmin > nmax;
}

__global__ __launch_bounds__(64 * DEEP_WAVES) void k_scan_deep(const uint8_t *__restrict__ arena, uint64_t arena_stride,
                                                               const HtDevLevel *__restrict__ levels, int next,
                                                               const HtPatchFeature *__restrict__ feats,
                                                               const HtDevStage *__restrict__ stages, int nstages, int use_int,
                                                               const HtQueueEntry *__restrict__ queue, uint32_t queue_cap,
                                                               ht_hit *__restrict__ hits, uint32_t hit_cap, HtCounters *__restrict__ ctr,
                                                               unsigned long long *__restrict__ stats) {
    __shared__ __attribute__((aligned(16))) uint8_t s_patch[DEEP_WAVES][PATCH_BYTES];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint8_t *patch = s_patch[wv];
    const uint32_t wave = blockIdx.x * DEEP_WAVES + wv, nwaves = gridDim.x * DEEP_WAVES;
    unsigned long long *my_stats = stats ? stats + (size_t)(wave & (HT_STAT_SHARDS - 1)) * 64 : nullptr;
    const uint32_t n = min(ctr->nqueue, queue_cap);
    for (uint32_t e = wave; e < n; e += nwaves) {
        const HtQueueEntry ent = queue[e];
        const uint8_t *fb = arena + (uint64_t)ent.frame * arena_stride;
        const uint32_t o0 = ent.o0, o1 = ent.o1, o2 = ent.o2;
        const int s0 = ent.s0, s1 = ent.s1, s2 = ent.s2;
        int j = (int)ent.stage;  // first stage to run here
        HtDevStage st = stages[j];
        // first feature chunk and the window patch are fetched together (one memory latency)
        PatchFeatRegs cur;
        if (lane < st.count) cur = load_feat(feats + st.first + lane);
        {
            uint32_t pa[5], pb[3], pc = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) {  // plane 0: 24 rows x 12 halfwords (window origin is 2-byte aligned)
                const uint32_t i = lane + 64u * k, r = i / 12u, c2 = (i - r * 12u) * 2u;
                pa[k] = 0;
                if (i < 288u) pa[k] = *reinterpret_cast<const uint16_t *>(fb + o0 + r * (uint32_t)s0 + c2);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {  // plane 1: 12 x 12 bytes
                const uint32_t i = lane + 64u * k, r = i / 12u, c = i - r * 12u;
                pb[k] = 0;
                if (i < 144u) pb[k] = fb[o1 + r * (uint32_t)s1 + c];
            }
            if (lane < 36u) {  // plane 2: 6 x 6 bytes
                const uint32_t r = lane / 6u, c = lane - r * 6u;
                pc = fb[o2 + r * (uint32_t)s2 + c];
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const uint32_t i = lane + 64u * k;
                if (i < 288u) *reinterpret_cast<uint16_t *>(&patch[2u * i]) = (uint16_t)pa[k];  // 24-byte rows are contiguous
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t i = lane + 64u * k;
                if (i < 144u) patch[PATCH1 + i] = (uint8_t)pb[k];
            }
            if (lane < 36u) patch[PATCH2 + lane] = (uint8_t)pc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool alive = true;
        double conf = 0.0;
        uint32_t kb = 0;       // chunk base inside stage j
        long long acc = 0;
        while (true) {
            // prefetch the next chunk (same stage, or speculatively the first chunk of the next stage)
            const bool stage_ends = kb + 64u >= st.count;
            int jn = j;
            uint32_t kbn = kb + 64u;
            HtDevStage stn = st;
            if (stage_ends) {
                jn = j + 1;
                kbn = 0;
                if (jn < nstages) stn = stages[jn];
            }
            PatchFeatRegs nxt = cur;
            if (jn < nstages && kbn + lane < stn.count) nxt = load_feat(feats + stn.first + kbn + lane);
            if (kb == 0 && lane == 0 && my_stats) atomicAdd(&my_stats[j], 1ull);
            if (use_int) {
                if (kb + lane < st.count) acc += fire_regs(patch, cur, st.maxpts) ? cur.a1i : cur.a0i;
            }
            if (stage_ends) {
                bool need_exact = true;
                if (use_int) {
                    const long long Ssum = wave_sum_i64(acc);
                    acc = 0;
                    if (Ssum < st.thri) {  // sum < threshold decided exactly, independent of summation order
                        alive = false;
                        break;
                    }
                    need_exact = (Ssum == st.thri) || (j == nstages - 1) || use_int == 2;
                }
                if (need_exact) {
                    const double sum = patch_stage_sum_exact(patch, feats + st.first, st.count, st.maxpts, lane);
                    if (sum < st.threshold) {  // ccv.js:222
                        alive = false;
                        break;
                    }
                    conf = sum;
                }
                if (jn >= nstages) break;
            }
            j = jn;
            kb = kbn;
            st = stn;
            cur = nxt;
        }
        if (alive && lane == 0) {
            if (my_stats) atomicAdd(&my_stats[nstages], 1ull);
            const uint32_t pos = atomicAdd(&ctr->nhits, 1u);
            if (pos < hit_cap) {
                ht_hit h;
                h.frame = ent.frame;
                h.x = ent.x;
                h.y = ent.y;
                h.scale = ent.scale;
                h.q = ent.q;
                h.reserved0 = 0;
                h.reserved1 = 0;
                h.sum = conf;
                hits[pos] = h;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the next window overwrites the patch
    }
}

// ---- deep kernel, LDS-resident feature table -------------------------------------------------------------------------------
// k_scan_deep re-reads every stage's 64-byte feature records from L2 for every window: ~100k chunk-steps x 4 KB = 400 MB per
// C2 batch, which is what it spends its time on.  Here the tail of the cascade (stages >= split, 1868 features x 32 B = 60 KB)
// is copied into LDS once per workgroup (one 1024-thread workgroup per CU); every wavefront then owns a window at a time
// exactly like k_scan_deep, but feature records and pixels both come from LDS.
constexpr uint32_t DEEP_LDS_TABLE_BYTES = 64 * 1024;
#ifndef HT_DEEPL_WAVES
#define HT_DEEPL_WAVES 12  // 12 waves x 2 workgroups per CU (2 x 77 KB of LDS) beat 16 x 1: measured 0.044 vs 0.055 ms on C2
#endif
constexpr int DEEPL_WAVES = HT_DEEPL_WAVES;

__device__ __forceinline__ bool packed_fire(const uint8_t *patch, const uint4 A, const uint32_t B0) {
    // A = off[0..7], B0 = off[8..9]
    const uint32_t p0 = A.x & 0xffffu, p1 = A.x >> 16, p2 = A.y & 0xffffu, p3 = A.y >> 16, p4 = A.z & 0xffffu;
    const uint32_t n0 = A.z >> 16, n1 = A.w & 0xffffu, n2 = A.w >> 16, n3 = B0 & 0xffffu, n4 = B0 >> 16;
    const uint32_t pmin = min(min(min((uint32_t)patch[p0], (uint32_t)patch[p1]), min((uint32_t)patch[p2], (uint32_t)patch[p3])), (uint32_t)patch[p4]);
    const uint32_t nmax = max(max(max((uint32_t)patch[n0], (uint32_t)patch[n1]), max((uint32_t)patch[n2], (uint32_t)patch[n3])), (uint32_t)patch[n4]);
    return pmin > nmax;
}

// wave64 integer sum with DPP row shifts / row broadcasts (VALU-only, ~10 instructions) instead of a ds_bpermute shuffle tree
// (6 dependent LDS round trips): the critical path of a window through the deep stages is what the deep kernel costs.
// Canonical gfx9 sequence: row_shr 1,2,3 on the input, row_shr 4 / 8 with bank masks, row_bcast 15 / 31; total in lane 63.
__device__ __forceinline__ int wave_sum_i32_dpp(int v) {
    int s = v + __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    s += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);         // row_shr:2
    s += __builtin_amdgcn_update_dpp(0, v, 0x113, 0xf, 0xf, true);         // row_shr:3
    s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xe, true);         // row_shr:4, banks 1-3
    s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xc, true);         // row_shr:8, banks 2-3
    s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, true);         // row_bcast:15 into rows 1, 3
    s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, true);         // row_bcast:31 into rows 2, 3
    return __builtin_amdgcn_readlane(s, 63);
}
// 64-bit sum of per-lane values in (-2^40, 2^40): three 21-bit limbs of (v + 2^40), each limb sum < 2^27
__device__ __forceinline__ long long wave_sum_i64_dpp(long long v) {
    const unsigned long long u = (unsigned long long)(v + (1ll << 40));
    const long long l0 = wave_sum_i32_dpp((int)(u & 0x1fffffu));
    const long long l1 = wave_sum_i32_dpp((int)((u >> 21) & 0x1fffffu));
    const long long l2 = wave_sum_i32_dpp((int)(u >> 42));
    return l0 + (l1 << 21) + (l2 << 42) - (64ll << 40);
}

// A queue entry through the SCALAR cache: the index is wave-uniform, but the kernel contains memory clobbers (the hand-written atomic), so the
// compiler does not prove the load "noclobber" and fetches the entry with vector loads into 8 VGPRs — which then live across the stage
// passes next to the prefetched next entry, and at this register budget the patch gathers were issued one by one, each waited for on the
// spot.  Read through the constant address space the entry is two s_load_dwordx4 into scalar registers.  (The tile kernel wrote the queue
// in an earlier launch: nothing in this kernel stores to it.)
struct DeepEntry {
    uint32_t frame, x, y, scale, q, stage, o0, o1, o2, s0, s1, s2;
};
__device__ __forceinline__ DeepEntry deep_entry(const HtQueueEntry *__restrict__ queue, uint32_t e) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) u32x4 *cptr4;
    cptr4 p = (cptr4)(reinterpret_cast<uintptr_t>(queue) + (uintptr_t)e * sizeof(HtQueueEntry));
    const u32x4 w0 = p[0], w1 = p[1];
    DeepEntry d;
    d.frame = w0.x, d.x = w0.y & 0xffffu, d.y = w0.y >> 16;
    d.scale = w0.z & 0xffu, d.q = (w0.z >> 8) & 0xffu, d.stage = w0.z >> 16;
    d.o0 = w0.w, d.o1 = w1.x, d.o2 = w1.y;
    d.s0 = w1.z & 0xffffu, d.s1 = w1.z >> 16, d.s2 = w1.w & 0xffffu;
    return d;
}

#ifdef HT_DEEP_TIMELINE  // tools/gpu_deep_timeline.py: shader-clock stamps per wavefront of k_scan_deep_lds (first window of each wavefront)
__device__ unsigned long long g_deep_tl[8192][8];  // entry, table copied, patch loaded, window done, last stage run, exact-sum start
#define DL_STAMP(i)                                                                                                  \
    do {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if ((threadIdx.x & 63u) == 0 && dl_first) g_deep_tl[(blockIdx.x * DEEPL_WAVES + (threadIdx.x >> 6)) & 8191u][i] = __builtin_readcyclecounter(); \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#else
#define DL_STAMP(i)
#endif
__global__ __launch_bounds__(64 * DEEPL_WAVES, (2 * DEEPL_WAVES + 3) / 4) void k_scan_deep_lds(const uint8_t *__restrict__ arena, uint64_t arena_stride,
                                                                   const HtDevLevel *__restrict__ levels, int next,
                                                                   const HtPackedFeature *__restrict__ packed, uint32_t packed_count,
                                                                   uint32_t packed_first, const HtDevStage *__restrict__ stages, int nstages,
                                                                   int force_exact, const HtQueueEntry *__restrict__ queue, uint32_t queue_cap,
                                                                   ht_hit *__restrict__ hits, uint32_t hit_cap, HtCounters *__restrict__ ctr,
                                                                   unsigned long long *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    // LDS: [packed feature table][stage table][per wave: window patch (768 B) + 64 selected alphas (512 B)]
    uint4 *tab = reinterpret_cast<uint4 *>(dyn_lds);
    HtDevStage *s_stages = reinterpret_cast<HtDevStage *>(dyn_lds + (size_t)packed_count * sizeof(HtPackedFeature));
    uint8_t *per_wave = reinterpret_cast<uint8_t *>(s_stages + 64);
    const uint32_t n = min(ctr->nqueue, queue_cap);
    if (blockIdx.x * DEEPL_WAVES >= n) return;  // nothing for this workgroup: skip the table copy
#ifdef HT_DEEP_TIMELINE
    bool dl_first = true;
#endif
    DL_STAMP(0);
    const uint32_t lane = threadIdx.x & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: the patch address, the queue index and the statistics row become scalar registers
    const uint32_t wave = blockIdx.x * DEEPL_WAVES + wv, nwaves = (uint32_t)__builtin_amdgcn_readfirstlane((int)(gridDim.x * DEEPL_WAVES));
    // The memory round trips in front of a window's first stage used to be five, one behind the other: queue count -> table copy ->
    // queue entry -> the scale's level records -> patch gathers (tools/gpu_deep_timeline.py: table copy 5 k + patch load 5.5 k cycles).
    // Now the wavefront's first entry is requested before the table copy (it needs nothing from it), entries carry their plane origins,
    // and every further entry is requested as soon as its index is known, a whole window ahead: one round trip (the gathers) per window.
    DeepEntry ent = deep_entry(queue, min(wave, n - 1u));
    // The table copy is LDS-DMA (gfx950 `global_load_lds_dwordx4`: 1 KB per instruction straight into LDS, no registers) and is NOT waited
    // for here: the first window's patch gathers are issued behind it and both arrive together — the workgroup barrier that publishes the
    // table sits in front of the first window's stage passes (wavefronts without a window take it behind the loop).
    {
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
        const uint32_t tab_bytes = packed_count * (uint32_t)sizeof(HtPackedFeature);
        for (uint32_t c0 = wv * 1024u; c0 < tab_bytes; c0 += DEEPL_WAVES * 1024u) {
            const uint32_t off = c0 + lane * 16u;
            if (off < tab_bytes)
                __builtin_amdgcn_global_load_lds((glb_void *)(reinterpret_cast<const uint8_t *>(packed) + off), (lds_void *)(dyn_lds + c0), 16, 0, 0);
        }
    }
    for (uint32_t i = threadIdx.x; i < (uint32_t)nstages * (sizeof(HtDevStage) / 16); i += blockDim.x)
        reinterpret_cast<uint4 *>(s_stages)[i] = reinterpret_cast<const uint4 *>(stages)[i];
    bool table_published = false;
    uint8_t *patch = per_wave + wv * (PATCH_BYTES + 512);
    double *sel_buf = reinterpret_cast<double *>(patch + PATCH_BYTES);
    unsigned long long *my_stats = stats ? stats + (size_t)(wave & (HT_STAT_SHARDS - 1)) * 64 : nullptr;
    // Queue entries are handed out dynamically: a wavefront's first window is entry `wave`, every further one comes from a counter.
    // A window that survives all 16 stages costs ~43 k cycles, one that dies in the stage it was handed over at ~6 k, and a C2 batch
    // queues 6 103 windows (1 392 full survivors) for the 2 304 wavefronts of the 192-workgroup grid: with the fixed stride
    // e += nwaves the launch lasted as long as the wavefront that drew three full survivors (0.055 ms).  The counter is read one
    // window ahead (the atomic's round trip is hidden behind the window being evaluated); HT_DEEP_CTRS counters, each handing out
    // every HT_DEEP_CTRS-th entry to the wavefronts that share it (see ht_internal.h).
#ifndef HT_DEEP_DYNAMIC
#define HT_DEEP_DYNAMIC 1
#endif
    const uint32_t *my_ctr = reinterpret_cast<const uint32_t *>(queue + queue_cap) + (wave & (HT_DEEP_CTRS - 1u)) * 64u;
    for (uint32_t e = wave; e < n;) {
        e = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);  // the queue entry is a scalar load
        uint32_t nxt = 0;
        if (HT_DEEP_DYNAMIC && lane == 0) {
            // written out: atomicAdd() is turned into a wave-aggregated add whose result is waited for on the spot (~2 us per window);
            // the compiler's own s_waitcnt vmcnt(k) stay correct with one more (older) operation in the in-order queue
            const uint32_t zero = 0u, one = 1u;
            // `nxt` is NOT valid until the s_waitcnt vmcnt(0) asm below: nothing may read, copy or spill its register in between.  The
            // compiler does not know that; tests/test_abi.py::test_deep_kernel_atomic_result_is_untouched_until_waited_for checks the
            // code object of every build (the register is written by the atomic and first read after that s_waitcnt).
            asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(nxt) : "v"(zero), "v"(one), "s"(my_ctr) : "memory");
        }
        uint32_t ln = lane;  // the gathers' index arithmetic from an opaque lane number: as loop invariants it would be hoisted out of the window
        asm volatile("" : "+v"(ln));  // loop and held (spilled, at this budget) across the stage passes
        const uint8_t *fb = arena + (uint64_t)ent.frame * arena_stride;
        const uint32_t o0 = ent.o0, o1 = ent.o1, o2 = ent.o2;
        const int s0 = (int)ent.s0, s1 = (int)ent.s1, s2 = (int)ent.s2;
        const DeepEntry cur = ent;  // this window (frame, x, y, scale, q, first stage); `ent` becomes the next one below
        uint32_t e_next = n;
        {
            uint32_t pa[5], pb[3], pc = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const uint32_t i = ln + 64u * k, r = i / 12u, c2 = (i - r * 12u) * 2u;
                pa[k] = 0;
                if (i < 288u) pa[k] = *reinterpret_cast<const uint16_t *>(fb + o0 + r * (uint32_t)s0 + c2);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t i = ln + 64u * k, r = i / 12u, c = i - r * 12u;
                pb[k] = 0;
                if (i < 144u) pb[k] = fb[o1 + r * (uint32_t)s1 + c];
            }
            if (ln < 36u) {
                const uint32_t r = ln / 6u, c = ln - r * 6u;
                pc = fb[o2 + r * (uint32_t)s2 + c];
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const uint32_t i = ln + 64u * k;
                if (i < 288u) *reinterpret_cast<uint16_t *>(&patch[2u * i]) = (uint16_t)pa[k];
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t i = ln + 64u * k;
                if (i < 144u) patch[PATCH1 + i] = (uint8_t)pb[k];
            }
            if (ln < 36u) patch[PATCH2 + ln] = (uint8_t)pc;
            // the next entry's index has arrived with the gathers (issued before them): parked in the patch's spare bytes, not in a
            // register that would live across the stage passes (at 80 VGPRs that one register was 9 spills)
            static_assert(PATCH2 + 36 <= 760 && PATCH_BYTES >= 764, "spare bytes of the patch");
            if (HT_DEEP_DYNAMIC) {
                if (lane == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt) : : "memory");  // the gathers before it in the queue have been consumed above
                // the k-th entry handed out by counter c is nwaves + HT_DEEP_CTRS * k + c; requested now, it arrives during the stage passes
                e_next = nwaves + HT_DEEP_CTRS * (uint32_t)__builtin_amdgcn_readlane((int)nxt, 0) + (wave & (HT_DEEP_CTRS - 1u));
            } else {
                e_next = e + nwaves;
            }
            if (e_next < n) ent = deep_entry(queue, e_next);  // two scalar loads (wave-uniform index), in flight during the stage passes
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!table_published) {  // wave-uniform; every wavefront of the workgroup passes exactly one of the two barriers
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wavefront's share of the table has landed in LDS
            __syncthreads();
            table_published = true;
            DL_STAMP(1);
        }
        DL_STAMP(2);
        bool alive = true;
        double conf = 0.0;
        for (int j = (int)cur.stage; j < nstages; j++) {
            const HtDevStage st = s_stages[j];
            const uint32_t base = st.first - packed_first;  // index of the stage's first record in the LDS table
            if (lane == 0 && my_stats) atomicAdd(&my_stats[j], 1ull);
            long long acc = 0;
            // The LAST stage keeps what every lane's features selected (alpha * 1e8, one register per 64-feature chunk, up to KEEP chunks):
            // a window that passes it needs the sequential binary64 sum of exactly these alphas for its confidence, and re-evaluating the
            // fires chunk by chunk in front of every 64 adds was a third of that sum's time.
            constexpr int KEEP = 10;
            int32_t kept[KEEP];
            const bool keep = (j == nstages - 1) && st.count <= 64u * KEEP;
            if (keep) {
#pragma unroll
                for (int cki = 0; cki < KEEP; cki++) {
                    const uint32_t k = 64u * cki + lane;
                    kept[cki] = 0;
                    if (64u * cki < st.count && k < st.count) {  // first test is wave-uniform
                        const uint4 A = tab[(base + k) * 2u], Bq = tab[(base + k) * 2u + 1u];
                        kept[cki] = packed_fire(patch, A, Bq.x) ? (int32_t)Bq.z : (int32_t)Bq.y;
                        acc += (long long)kept[cki];
                    }
                    __builtin_amdgcn_sched_barrier(0);  // one chunk at a time: interleaved, ten chunks' records and pixels do not fit the register budget
                }
            } else {
                for (uint32_t k = lane; k < st.count; k += 64) {
                    const uint4 A = tab[(base + k) * 2u], Bq = tab[(base + k) * 2u + 1u];
                    acc += packed_fire(patch, A, Bq.x) ? (long long)(int32_t)Bq.z : (long long)(int32_t)Bq.y;
                }
            }
            const long long Ssum = wave_sum_i64_dpp(acc);
            if (Ssum < st.thri) {  // sum < threshold decided exactly, independent of summation order
                alive = false;
                break;
            }
#ifdef HT_DEEP_TIMELINE
            if (lane == 0 && dl_first) g_deep_tl[(blockIdx.x * DEEPL_WAVES + wv) & 8191u][4] = (unsigned long long)j;
#endif
            if (Ssum == st.thri || j == nstages - 1 || force_exact) {
                DL_STAMP(5);
                // sequential binary64 sum in the reference's order (ccv.js:186-221): every lane parks the alpha its feature selected in LDS,
                // then the wave adds them in feature order from broadcast reads.  Only the adds form the chain: the reads of the next 8
                // alphas are in flight while 8 are added, and the last stage's selected alphas were kept in registers by its integer pass
                // (tools/gpu_deep_timeline.py: this sum was 26 k of a full survivor's 63 k cycles with 8 reads, then 8 adds, per step and
                // the fires recomputed in between).  Lanes past the stage's end park +0.0: adding +0.0 never changes a binary64 sum
                // that is not -0.0, and the sum starts at +0.0 — so every chunk is 64 adds, no trip counts.
                double sum = 0.0;
                auto sel_of = [&](uint32_t k) -> double {
                    double sel = 0.0;
                    if (k < st.count) {
                        const uint4 A = tab[(base + k) * 2u], Bq = tab[(base + k) * 2u + 1u];
                        const int32_t ai = packed_fire(patch, A, Bq.x) ? (int32_t)Bq.z : (int32_t)Bq.y;
                        sel = (double)ai / 1e8;  // == alpha exactly (checked when the table was packed)
                    }
                    return sel;
                };
                auto add64 = [&](double sel) {  // sum += sel[lane 0] + ... + sel[lane 63], in lane order
                    sel_buf[lane] = sel;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    double g[8], h[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) g[u] = sel_buf[u];
#pragma unroll
                    for (int grp = 0; grp < 8; grp++) {
                        if (grp < 7) {
#pragma unroll
                            for (int u = 0; u < 8; u++) h[u] = sel_buf[8 * (grp + 1) + u];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 8; u++) sum = __dadd_rn(sum, g[u]);
#pragma unroll
                        for (int u = 0; u < 8; u++) g[u] = h[u];
                    }
                    __builtin_amdgcn_wave_barrier();  // every lane has read the buffer before the next chunk overwrites it
                };
                if (keep) {
#pragma unroll
                    for (int cki = 0; cki < KEEP; cki++)
                        if (64u * cki < st.count) add64((double)kept[cki] / 1e8);  // lanes past the stage's end kept 0 -> +0.0
                } else {
                    for (uint32_t kb = 0; kb < st.count; kb += 64) add64(sel_of(kb + lane));
                }
                if (sum < st.threshold) {  // ccv.js:222
                    alive = false;
                    break;
                }
                conf = sum;
            }
        }
        if (alive && lane == 0) {
            if (my_stats) atomicAdd(&my_stats[nstages], 1ull);
            const uint32_t pos = atomicAdd(&ctr->nhits, 1u);
            if (pos < hit_cap) {
                ht_hit h;
                h.frame = cur.frame;
                h.x = (uint16_t)cur.x;
                h.y = (uint16_t)cur.y;
                h.scale = (uint8_t)cur.scale;
                h.q = (uint8_t)cur.q;
                h.reserved0 = 0;
                h.reserved1 = 0;
                h.sum = conf;
                hits[pos] = h;
            }
        }
        DL_STAMP(3);
#ifdef HT_DEEP_TIMELINE
        dl_first = false;
#endif
        __builtin_amdgcn_wave_barrier();
        e = e_next;
    }
    if (!table_published) {
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------------------------------
// simple kernel: one thread per window, every stage, straight from HBM (independent cross-check of the tiled path)

__global__ __launch_bounds__(256) void k_scan_simple(const uint8_t *__restrict__ arena, uint64_t arena_stride,
                                                     const HtDevLevel *__restrict__ levels, int next,
                                                     const HtScanScale *__restrict__ scales, int nscales, uint32_t windows_per_frame,
                                                     const HtDeepFeature *__restrict__ feats, const HtDevStage *__restrict__ stages,
                                                     int nstages, ht_hit *__restrict__ hits, uint32_t hit_cap,
                                                     HtCounters *__restrict__ ctr, unsigned long long *__restrict__ stats) {
    unsigned long long *my_stats = stats ? stats + (size_t)(blockIdx.x & (HT_STAT_SHARDS - 1)) * 64 : nullptr;
    const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t frame = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63u;
    bool alive = wi < windows_per_frame;
    int si = 0;
    for (int k = 1; k < nscales; k++)
        if (wi >= scales[k].win_begin) si = k;
    const HtScanScale S = scales[si];
    const uint32_t li = alive ? wi - S.win_begin : 0u;
    const uint32_t per = (uint32_t)(S.qw * S.qh);
    const uint32_t q = li / per, r = li - q * per, y = r / (uint32_t)S.qw, x = r - y * (uint32_t)S.qw;
    HT_WIN_SETUP(arena + (uint64_t)frame * arena_stride, levels, next, (uint32_t)S.l0, q, x, y)
    double sum = 0.0;
    for (int j = 0; j < nstages; j++) {
        const unsigned long long m = __ballot(alive);
        if (!m) break;
        if (lane == 0 && my_stats) atomicAdd(&my_stats[j], (unsigned long long)__popcll(m));
        const HtDevStage st = stages[j];
        const HtDeepFeature *F = feats + st.first;
        sum = 0.0;
        if (alive) {
            for (uint32_t k = 0; k < st.count; k++) {
                const HtDeepFeature &f = F[k];
                sum = __dadd_rn(sum, deep_fire(&f, HT_WIN_PASS, st.maxpts) ? f.a1 : f.a0);
            }
            if (sum < st.threshold) alive = false;
        }
    }
    if (alive) {
        if (my_stats) atomicAdd(&my_stats[nstages], 1ull);
        const uint32_t pos = atomicAdd(&ctr->nhits, 1u);
        if (pos < hit_cap) {
            ht_hit h;
            h.frame = frame;
            h.x = (uint16_t)x;
            h.y = (uint16_t)y;
            h.scale = (uint8_t)S.l0;
            h.q = (uint8_t)q;
            h.reserved0 = 0;
            h.reserved1 = 0;
            h.sum = sum;
            hits[pos] = h;
        }
    }
}

}  // namespace

// ----------------------------------------------------------------------------------------------------------------
// host side

// LDS-offset form of every feature for k_scan_tiles (unified-base layout, see the file header)
ht_status ht_scan_tile_tables(ht_ctx *c) {
    std::vector<HtTileFeature> tf(c->nfeat);
    const bool tile_ok = (c->cw == 24 && c->ch == 24);
    for (uint32_t k = 0; k < c->nfeat; k++) {
        const HtBlobFeature &f = c->h_feats[k];
        HtTileFeature &t = tf[k];
        std::memset(&t, 0, sizeof(t));
        auto off = [](int x, int y, int z) -> uint16_t {
            if (z == 0) return (uint16_t)(y * PITCH0 + x);                       // level i: 1 B/px, pitch P
            if (z == 1) return (uint16_t)(P12_BASE + y * G_PITCH + 2 * x);       // level i+6: 2 B/px, pitch 2P
            return (uint16_t)(P12_BASE + 1 + 4 * y * PITCH0 + 4 * x);            // level i+12 variants: odd bytes, 4 B/px, pitch 4P
        };
        for (int q = 0; q < f.size; q++) {
            if (f.pz[q] >= 0) {
                t.po[t.np >> 1] |= (uint32_t)off(f.px[q], f.py[q], f.pz[q]) << (16 * (t.np & 1));
                t.np++;
            }
            if (f.nz[q] >= 0) {
                t.no[t.nn >> 1] |= (uint32_t)off(f.nx[q], f.ny[q], f.nz[q]) << (16 * (t.nn & 1));
                t.nn++;
            }
        }
        std::memcpy(&t.a[0], &f.alpha[0], 8);
        std::memcpy(&t.a[2], &f.alpha[1], 8);
    }
    (void)tile_ok;
    HT_HIP(c, hipMalloc(&c->d_tile_feats, tf.size() * sizeof(HtTileFeature)));
    HT_HIP(c, hipMemcpy(c->d_tile_feats, tf.data(), tf.size() * sizeof(HtTileFeature), hipMemcpyHostToDevice));
    {   // packed per-lane form of the same offsets for the tile kernel's feature-parallel sparse phase: slots past a polarity's point
        // count repeat its first point (min / max are idempotent), alpha[2k+1] * 1e8 as the integer the generated stages add
        std::vector<HtPackedFeature> fp(c->nfeat);
        bool ok = c->decimal_alphas;
        for (uint32_t k = 0; k < c->nfeat && ok; k++) {
            const HtTileFeature &t = tf[k];
            const HtBlobFeature &f = c->h_feats[k];
            HtPackedFeature &q = fp[k];
            std::memset(&q, 0, sizeof(q));
            if (t.np == 0 || t.nn == 0 || t.np > 5 || t.nn > 5) ok = false;
            for (uint32_t j = 0; j < 5 && ok; j++) {
                const uint32_t jp = j < t.np ? j : 0u, jn = j < t.nn ? j : 0u;
                q.off[j] = (uint16_t)((t.po[jp >> 1] >> (16 * (jp & 1))) & 0xffffu);
                q.off[5 + j] = (uint16_t)((t.no[jn >> 1] >> (16 * (jn & 1))) & 0xffffu);
            }
            const double a0 = std::nearbyint(f.alpha[0] * 1e8), a1 = std::nearbyint(f.alpha[1] * 1e8);
            if (!(a1 > 0 && a1 < 2147483648.0 && a0 == -a1)) ok = false;
            q.a0i = (int32_t)a0, q.a1i = (int32_t)a1;
        }
        if (ok && c->nfeat) {
            HT_HIP(c, hipMalloc(&c->d_fp_feats, fp.size() * sizeof(HtPackedFeature)));
            HT_HIP(c, hipMemcpy(c->d_fp_feats, fp.data(), fp.size() * sizeof(HtPackedFeature), hipMemcpyHostToDevice));
        }
    }

    // patch-offset form for k_scan_deep
    std::vector<HtPatchFeature> pf(c->nfeat);
    for (uint32_t k = 0; k < c->nfeat; k++) {
        const HtBlobFeature &f = c->h_feats[k];
        HtPatchFeature &t = pf[k];
        std::memset(&t, 0, sizeof(t));
        auto poff = [&](int x, int y, int z) -> uint16_t {
            if (z == 0) return (uint16_t)(y * (int)c->cw + x);
            if (z == 1) return (uint16_t)(PATCH1 + y * (int)(c->cw / 2) + x);
            return (uint16_t)(PATCH2 + y * (int)(c->cw / 4) + x);
        };
        int np = 0, nn = 0;
        for (int q = 0; q < f.size; q++) {
            if (f.pz[q] >= 0) t.poff[np++] = poff(f.px[q], f.py[q], f.pz[q]);
            if (f.nz[q] >= 0) t.noff[nn++] = poff(f.nx[q], f.ny[q], f.nz[q]);
        }
        for (int q = np; q < HT_MAXPTS; q++) t.poff[q] = t.poff[0];
        for (int q = nn; q < HT_MAXPTS; q++) t.noff[q] = t.noff[0];
        t.a0 = f.alpha[0];
        t.a1 = f.alpha[1];
        double s0 = t.a0 * 1e8, s1 = t.a1 * 1e8;
        t.a0i = (int64_t)llround(s0);
        t.a1i = (int64_t)llround(s1);
    }
    HT_HIP(c, hipMalloc(&c->d_patch_feats, pf.size() * sizeof(HtPatchFeature)));
    HT_HIP(c, hipMemcpy(c->d_patch_feats, pf.data(), pf.size() * sizeof(HtPatchFeature), hipMemcpyHostToDevice));

    return HT_OK;
}

// LDS-resident deep table: features of stages [split, nstages) in HtPackedFeature form; leaves packed_count = 0 when the
// cascade does not fit the format (then k_scan_deep is used)
ht_status ht_scan_pack_deep(ht_ctx *c) {
    if (c->d_packed_feats) (void)hipFree(c->d_packed_feats), c->d_packed_feats = nullptr;
    c->packed_count = 0;
    if (!c->decimal_alphas || c->split_stage >= c->nstages || c->cw != 24 || c->ch != 24) return HT_OK;
    const uint32_t first = c->h_stages[c->split_stage].first;
    const uint32_t n = c->nfeat - first;
    if (n == 0 || n * sizeof(HtPackedFeature) > DEEP_LDS_TABLE_BYTES) return HT_OK;
    std::vector<HtPackedFeature> pk(n);
    for (uint32_t k = 0; k < n; k++) {
        const HtBlobFeature &f = c->h_feats[first + k];
        HtPackedFeature &t = pk[k];
        std::memset(&t, 0, sizeof(t));
        auto poff = [&](int x, int y, int z) -> uint16_t {
            if (z == 0) return (uint16_t)(y * 24 + x);
            if (z == 1) return (uint16_t)(PATCH1 + y * 12 + x);
            return (uint16_t)(PATCH2 + y * 6 + x);
        };
        int np = 0, nn = 0;
        for (int q = 0; q < f.size; q++) {
            if (f.pz[q] >= 0) {
                if (np >= 5) return HT_OK;
                t.off[np++] = poff(f.px[q], f.py[q], f.pz[q]);
            }
            if (f.nz[q] >= 0) {
                if (nn >= 5) return HT_OK;
                t.off[5 + nn++] = poff(f.nx[q], f.ny[q], f.nz[q]);
            }
        }
        for (int q = np; q < 5; q++) t.off[q] = t.off[0];
        for (int q = nn; q < 5; q++) t.off[5 + q] = t.off[5];
        const double s0 = f.alpha[0] * 1e8, s1 = f.alpha[1] * 1e8;
        if (!(std::fabs(s0) < 2.0e9) || !(std::fabs(s1) < 2.0e9)) return HT_OK;
        t.a0i = (int32_t)llround(s0);
        t.a1i = (int32_t)llround(s1);
        if ((double)t.a0i / 1e8 != f.alpha[0] || (double)t.a1i / 1e8 != f.alpha[1]) return HT_OK;
    }
    HT_HIP(c, hipMalloc(&c->d_packed_feats, pk.size() * sizeof(HtPackedFeature)));
    HT_HIP(c, hipMemcpy(c->d_packed_feats, pk.data(), pk.size() * sizeof(HtPackedFeature), hipMemcpyHostToDevice));
    c->packed_count = n;
    c->packed_first = first;
    return HT_OK;
}

// true iff blob is byte-identical to the cascade ht_cascade_gen.inc was generated from (FNV-1a 64 + length)
bool ht_scan_is_builtin_cascade(const uint8_t *blob, size_t len) {
    if (len != HT_GEN_CASCADE_LEN) return false;
    unsigned long long h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < len; i++) h = (h ^ blob[i]) * 0x100000001b3ull;
    return h == HT_GEN_CASCADE_FNV;
}

ht_status ht_scan_plan_tiles(ht_ctx *c) {
    c->h_scales.clear();
    c->windows_per_frame = 0;
    uint32_t tiles = 0;
    for (int i = 0; i < c->upto; i++) {  // ccv.js:154
        HtScanScale S;
        std::memset(&S, 0, sizeof(S));
        S.l0 = i;
        S.l1 = i + c->next;
        S.l2 = i + 2 * c->next;
        S.qw = c->h_levels[S.l2].w - (int)(c->cw / 4);  // ccv.js:155
        S.qh = c->h_levels[S.l2].h - (int)(c->ch / 4);  // ccv.js:156
        if (S.qw <= 0 || S.qh <= 0) continue;
        S.ntx = (2 * S.qw + TXH - 1) / TXH;
        S.tw2 = (2 * S.qw + S.ntx - 1) / S.ntx;
        S.tw2 = (S.tw2 + 7) & ~7;  // multiple of 8 half-steps: tile rows start on 16 / 8 / 4-byte boundaries of the three planes' rows
        S.nty = (2 * S.qh + TYH - 1) / TYH;
        S.th2 = (2 * S.qh + S.nty - 1) / S.nty;
        S.th2 += S.th2 & 1;
        S.ntx = (2 * S.qw + S.tw2 - 1) / S.tw2;
        S.nty = (2 * S.qh + S.th2 - 1) / S.th2;
        S.tile_begin = tiles;
        S.div_magic = ((1u << 20) + (uint32_t)S.tw2 - 1) / (uint32_t)S.tw2;
        if (c->windows_per_frame + 4ull * S.qw * S.qh > 0xffffffffull) return ht_fail(c, HT_ERR_INVALID, "frame too large");
        S.win_begin = (uint32_t)c->windows_per_frame;
        tiles += (uint32_t)(S.ntx * S.nty);
        c->windows_per_frame += 4ull * (uint64_t)S.qw * (uint64_t)S.qh;
        c->h_scales.push_back(S);
    }
    c->tiles_per_frame = tiles;
    if (!c->h_scales.empty()) {
        std::vector<HtTileRec> recs;
        for (const HtScanScale &S : c->h_scales) {
            const HtDevLevel &A = c->h_levels[S.l0], &B = c->h_levels[S.l1], &Cq = c->h_levels[S.l2];
            if (A.stride > 0xffff || A.h > 0xffff) return ht_fail(c, HT_ERR_INVALID, "frame too large");
            for (int y = 0; y < S.nty; y++)
                for (int x = 0; x < S.ntx; x++) {
                    HtTileRec r;
                    std::memset(&r, 0, sizeof(r));
                    const int X0 = x * S.tw2, Y0 = y * S.th2;
                    r.off0 = A.off[0], r.off1 = B.off[0];
                    for (int q = 0; q < 4; q++) r.off2[q] = Cq.off[q];
                    r.sh0 = (uint32_t)A.stride | (uint32_t)A.h << 16;
                    r.sh1 = (uint32_t)B.stride | (uint32_t)B.h << 16;
                    r.sh2 = (uint32_t)Cq.stride | (uint32_t)Cq.h << 16;
                    r.origin = (uint32_t)X0 | (uint32_t)Y0 << 16;
                    r.size = (uint32_t)std::min(S.tw2, 2 * S.qw - X0) | (uint32_t)std::min(S.th2, 2 * S.qh - Y0) << 16;
                    r.tw2_l0 = (uint32_t)S.tw2 | (uint32_t)S.l0 << 16;
                    r.div_magic = S.div_magic;
                    {  // n / (4 * th) == (n * magic) >> 24 for every pair index n of the tile (n <= 1024, 4 * th <= 128: error term n * 127 < 2^24 / 128); checked anyway
                        const uint32_t th = r.size >> 16, d = 4u * th;
                        r.strip_magic = ((1u << 24) + d - 1u) / d;
                        for (uint32_t n = 0; n < (uint32_t)(S.tw2 / 2) * th; n++)
                            if (((n * r.strip_magic) >> 24) != n / d) return ht_fail(c, HT_ERR_INVALID, "tile plan: strip_magic is not exact");
                    }
                    recs.push_back(r);
                }
        }
        HT_HIP(c, hipMalloc(&c->d_tile_recs, recs.size() * sizeof(HtTileRec)));
        HT_HIP(c, hipMemcpy(c->d_tile_recs, recs.data(), recs.size() * sizeof(HtTileRec), hipMemcpyHostToDevice));
        HT_HIP(c, hipMalloc(&c->d_scales, c->h_scales.size() * sizeof(HtScanScale)));
        HT_HIP(c, hipMemcpy(c->d_scales, c->h_scales.data(), c->h_scales.size() * sizeof(HtScanScale), hipMemcpyHostToDevice));
    }
    return HT_OK;
}

// tile kernel over tiles [first, first + count) of every frame's tile list, on `stream`
static ht_status launch_tiles(ht_ctx *c, uint32_t flags, hipStream_t stream, uint32_t first, uint32_t count) {
    unsigned long long *stats = (flags & HT_SCAN_STATS) ? c->d_stats : nullptr;
    const int split = (flags & HT_SCAN_NO_SPLIT) ? (int)c->nstages : (int)std::min<uint32_t>(c->split_stage, c->nstages);
    const uint64_t total64 = (uint64_t)count * (uint64_t)c->nframes;
    if (total64 > 0x7fffff00ull) return ht_fail(c, HT_ERR_INVALID, "ht_detect: batch too large for one launch");
    const uint32_t total = (uint32_t)total64;
    const bool gen = c->builtin_cascade && !(flags & HT_SCAN_GENERIC);
    // measurement knob: stop the tile kernel before a stage; test knob: treat every integer stage decision as an exact tie,
    // i.e. always take the sequential-binary64 fallback (both read from the environment once, in ht_create)
    const int stop_stage = c->dbg_stop_stage, force_exact = c->dbg_force_exact;
    HtProfScope ps(c, "scan_tiles", stream);
    if (gen)
        hipLaunchKernelGGL(k_scan_tiles<true>, dim3((total + 7u) & ~7u), dim3(NT), 0, stream, c->d_arena, c->arena_stride,
                           c->d_tile_recs + first, c->d_tile_feats, c->fp_sparse ? c->d_fp_feats : nullptr, c->d_stages, (int)c->nstages, split, c->deep_bias, stop_stage, force_exact, count, total,
                           c->d_queue, c->queue_capacity, c->d_hits, c->hit_capacity, c->d_counters, stats);
    else
        hipLaunchKernelGGL(k_scan_tiles<false>, dim3((total + 7u) & ~7u), dim3(NT), 0, stream, c->d_arena, c->arena_stride,
                           c->d_tile_recs + first, c->d_tile_feats, c->fp_sparse ? c->d_fp_feats : nullptr, c->d_stages, (int)c->nstages, split, c->deep_bias, stop_stage, force_exact, count, total,
                           c->d_queue, c->queue_capacity, c->d_hits, c->hit_capacity, c->d_counters, stats);
    HT_HIP(c, hipGetLastError());
    return HT_OK;
}

// Called by ht_launch_pyramid right after the generation that completes the early scales' planes: their tiles are scanned on the
// second stream while the main stream builds the remaining small generations (which are latency-, not throughput-bound).
ht_status ht_launch_scan_early(ht_ctx *c, uint32_t flags) {
    if (c->early_tiles == 0 || !c->aux_stream || (flags & HT_SCAN_SIMPLE) || c->cw != 24 || c->ch != 24) return HT_OK;
    HT_HIP(c, hipEventRecord(c->ev_early_ready, c->stream));
    HT_HIP(c, hipStreamWaitEvent(c->aux_stream, c->ev_early_ready, 0));
    ht_status st = launch_tiles(c, flags, c->aux_stream, 0, c->early_tiles);
    if (st != HT_OK) return st;
    HT_HIP(c, hipEventRecord(c->ev_early_done, c->aux_stream));
    c->early_launched = true;
    return HT_OK;
}

ht_status ht_launch_scan(ht_ctx *c, uint32_t flags) {
    if (c->h_scales.empty() || c->tiles_per_frame == 0) return HT_OK;  // image too small for any window
    const int nscales = (int)c->h_scales.size();
    unsigned long long *stats = (flags & HT_SCAN_STATS) ? c->d_stats : nullptr;
    const bool tile_ok = (c->cw == 24 && c->ch == 24);
    if ((flags & HT_SCAN_SIMPLE) || !tile_ok) {
        HtProfScope ps(c, "scan_simple");
        dim3 grid((uint32_t)((c->windows_per_frame + 255) / 256), c->nframes);
        hipLaunchKernelGGL(k_scan_simple, grid, dim3(256), 0, c->stream, c->d_arena, c->arena_stride, c->d_levels, c->next, c->d_scales,
                           nscales, (uint32_t)c->windows_per_frame, c->d_deep_feats, c->d_stages, (int)c->nstages, c->d_hits,
                           c->hit_capacity, c->d_counters, stats);
        HT_HIP(c, hipGetLastError());
        return HT_OK;
    }
    const int split = (flags & HT_SCAN_NO_SPLIT) ? (int)c->nstages : (int)std::min<uint32_t>(c->split_stage, c->nstages);
    const int force_exact = c->dbg_force_exact;
    {
        const uint32_t first = c->early_launched ? c->early_tiles : 0u;
        ht_status st = launch_tiles(c, flags, c->stream, first, c->tiles_per_frame - first);
        if (st != HT_OK) return st;
        if (c->early_launched) HT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_early_done, 0));  // the deep kernel needs every tile's survivors
    }
    if (split < (int)c->nstages) {
        HtProfScope ps(c, "scan_deep");
        const int deep_v = c->dbg_deep_v;
        if (deep_v == 4 && c->packed_count && c->h_stages[split].first >= c->packed_first) {
            const size_t lds = (size_t)c->packed_count * sizeof(HtPackedFeature) + 64 * sizeof(HtDevStage) + (size_t)DEEPL_WAVES * (PATCH_BYTES + 512);
            if (!c->deep_attr_set) {  // per context (= per device): a single-process multi-GPU host has one context per GPU
                HT_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_scan_deep_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                c->deep_attr_set = true;
            }
            // Grid: 192 workgroups (option deep_grid), not the 512 that would give every queued window of a C2 batch a wavefront of its own.
            // A workgroup holds 77 KB of LDS for as long as its slowest window lives: 512 of them take every CU's LDS for the whole launch and
            // nothing of the other batch in flight runs beside it; 192 leave a quarter of the CUs free and half of the LDS on the others.
            // Measured with two batches in flight (round 4, tools/gpu_kernel_times.py): the kernel itself 0.038 -> 0.055 ms, the C2 step
            // 0.2514 -> 0.2440 ms (96 / 128 / 160 / 192 / 224 / 256 / 320 / 384 / 512 workgroups: 0.2528 / 0.2465 / 0.2445 / 0.2439 / 0.2470 /
            // 0.2456 / 0.2474 / 0.2480 / 0.2514), C4 unchanged; 8- and 16-wavefront workgroups have the same optimum.
            // every one of the HT_DEEP_CTRS work counters needs a wavefront that draws from it (entry nwaves + 16 k + c is only ever
            // handed out by counter c): a grid below 16 wavefronts (option deep_grid=1) would silently skip queue entries
            const uint32_t deep_grid = std::max<uint32_t>((uint32_t)c->deep_grid, (HT_DEEP_CTRS + DEEPL_WAVES - 1) / DEEPL_WAVES);
            hipLaunchKernelGGL(k_scan_deep_lds, dim3(deep_grid), dim3(64 * DEEPL_WAVES), lds, c->stream, c->d_arena, c->arena_stride, c->d_levels, c->next,
                               c->d_packed_feats, c->packed_count, c->packed_first, c->d_stages, (int)c->nstages, force_exact, c->d_queue, c->queue_capacity,
                               c->d_hits, c->hit_capacity, c->d_counters, stats);
        } else
        hipLaunchKernelGGL(k_scan_deep, dim3(2048), dim3(64 * DEEP_WAVES), 0, c->stream, c->d_arena, c->arena_stride, c->d_levels, c->next,
                           c->d_patch_feats, c->d_stages, (int)c->nstages, c->decimal_alphas ? (force_exact ? 2 : 1) : 0, c->d_queue, c->queue_capacity, c->d_hits,
                           c->hit_capacity, c->d_counters, stats);
        HT_HIP(c, hipGetLastError());
    }
    return HT_OK;
}

#ifdef HT_DEEP_TIMELINE
extern "C" int ht_debug_deep_timeline(unsigned long long *out, int nmax) {  // out[nmax][8]
    static unsigned long long h[8192][8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_deep_tl), sizeof(h)) != hipSuccess) return -1;
    const int n = std::min(nmax, 8192);
    std::memcpy(out, h, (size_t)n * 8 * sizeof(unsigned long long));
    std::memset(h, 0, sizeof(h));
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_deep_tl), h, sizeof(h)) != hipSuccess) return -1;
    return n;
}
#endif
