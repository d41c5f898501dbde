// ht_pyramid.hip — grayscale + image pyramid kernels (gfx950).
//
// Reference behaviour (paths under /root/reference/src/):
//   ccv.grayscale            ccv.js:22-32     g = R*0.3 + G*0.59 + B*0.11 in binary64, Uint8ClampedArray store
//   pyramid build            ccv.js:113-147   canvas drawImage at 39 levels (+3 shifted variants for levels >= 12)
//   getWhitebalance          whitebalance.js:5-30
// drawImage's resampling filter is browser-defined; this repo declares it in oracle/canvas_shim.js (centre-aligned
// bilinear in binary64, round-half-even store) and the kernels below implement that declaration bit for bit:
// explicit __dmul_rn/__dadd_rn (never contracted into FMAs), ratios divided on the host.
//
// HBM layout: the reference keeps 4-byte RGBA copies of every level but only ever reads byte 0 (ccv.js:171,173,
// 191-192), so each plane is stored once as 1 byte/pixel (row stride = width rounded up to 4, plane base 256-B
// aligned) inside a per-frame arena; frames are arena_stride bytes apart.  A whole batch is one launch per dependency
// generation: k_resample_bands (round 6: LDS-DMA into wave-private source bands; the default) or k_resample (register-staged
// tile, option rs_bands=0), and one tail launch (k_resample_tail*) for the tiny last generations.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "ht_internal.h"

namespace {

// XCD-aware work order shared by every detect kernel (the scan uses the same formula): the dispatcher places
// workgroup b on XCD b % 8, so linear work item t = (b % 8) * (grid / 8) + b / 8 gives XCD k the contiguous range of
// frames [k*n/8, (k+1)*n/8).  A frame's gray plane, pyramid and scan then stay in one XCD's L2 across the launches
// (measured before: 8 x over-fetch of level 0 by k_resample, every XCD pulling every frame).  Speed only — nothing
// depends on the placement.  gridDim.x must be a multiple of 8; returns false for the padding blocks.
__device__ __forceinline__ bool xcd_item(uint32_t per_frame, uint32_t nframes, uint32_t *frame, uint32_t *item) {
    const uint32_t chunk = gridDim.x >> 3;
    const uint32_t t = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
    if (t >= per_frame * nframes) return false;
    *frame = t / per_frame;
    *item = t - *frame * per_frame;
    return true;
}

__device__ __forceinline__ uint32_t gray_of(uint32_t px) {  // ccv.js:29
    const double r = (double)(px & 0xffu), g = (double)((px >> 8) & 0xffu), b = (double)((px >> 16) & 0xffu);
    const double v = __dadd_rn(__dadd_rn(__dmul_rn(r, 0.3), __dmul_rn(g, 0.59)), __dmul_rn(b, 0.11));
    const int q = (int)__builtin_rint(v);  // round half to even (Uint8ClampedArray)
    return (uint32_t)min(max(q, 0), 255);
}

// RGBA -> planar gray (level 0).  ALIGNED: W % 4 == 0, so plane and frame are both linear and 16-byte aligned per
// group of 4 pixels: one dwordx4 load + one dword store per thread, fully coalesced.
// WB: the same pass also accumulates the frame's R, G, B channel sums for headtrackr.getWhitebalance (whitebalance.js:5-30) —
// the pixels are in registers anyway, so facetrackr's white-balance figure costs no second read of the frame.
// Exact: integer sums (u32 per thread and per wavefront, u64 per frame), any order gives the reference's value.
template <bool GRAY_IN_R, bool WB>
__global__ __launch_bounds__(256) void k_gray_linear(const uint8_t *__restrict__ frames, size_t frame_stride,
                                                     uint8_t *__restrict__ arena, uint64_t arena_stride, uint32_t off0,
                                                     uint32_t ngroups, uint32_t blocks_per_frame, uint32_t nframes,
                                                     unsigned long long *__restrict__ wb_sums) {
    uint32_t f, blk;
    if (!xcd_item(blocks_per_frame, nframes, &f, &blk)) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(frames + (size_t)f * frame_stride);
    uint32_t *dst = reinterpret_cast<uint32_t *>(arena + (uint64_t)f * arena_stride + off0);
    uint32_t sr = 0, sg = 0, sb = 0;
    for (uint32_t g = blk * blockDim.x + threadIdx.x; g < ngroups; g += blocks_per_frame * blockDim.x) {
        const uint4 p = src[g];
        uint32_t o;
        if (GRAY_IN_R)
            o = (p.x & 0xff) | ((p.y & 0xff) << 8) | ((p.z & 0xff) << 16) | ((p.w & 0xff) << 24);
        else
            o = gray_of(p.x) | (gray_of(p.y) << 8) | (gray_of(p.z) << 16) | (gray_of(p.w) << 24);
        dst[g] = o;
        if (WB) {
            // two channels per add: R and B sit in bits 0-7 / 16-23 of a pixel, so (px & 0x00ff00ff) summed over 4 pixels
            // cannot carry between the halves (4 * 255 < 2^16)
            const uint32_t rb = (p.x & 0x00ff00ffu) + (p.y & 0x00ff00ffu) + (p.z & 0x00ff00ffu) + (p.w & 0x00ff00ffu);
            sr += rb & 0xffffu;
            sb += rb >> 16;
            sg += ((p.x >> 8) & 0xffu) + ((p.y >> 8) & 0xffu) + ((p.z >> 8) & 0xffu) + ((p.w >> 8) & 0xffu);
        }
    }
    if (WB) {
        __shared__ uint32_t s_part[4][3];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            sr += __shfl_xor(sr, s, 64);
            sg += __shfl_xor(sg, s, 64);
            sb += __shfl_xor(sb, s, 64);
        }
        if ((threadIdx.x & 63u) == 0) s_part[threadIdx.x >> 6][0] = sr, s_part[threadIdx.x >> 6][1] = sg, s_part[threadIdx.x >> 6][2] = sb;
        __syncthreads();
        if (threadIdx.x < 3) {  // one 64-bit atomic per channel and workgroup
            const unsigned long long v = (unsigned long long)s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
            atomicAdd(&wb_sums[(size_t)f * 4 + threadIdx.x], v);
        }
    }
}

// General width: 2-D mapping, 4 pixels per thread with dword loads.
template <bool GRAY_IN_R>
__global__ __launch_bounds__(256) void k_gray_rows(const uint8_t *__restrict__ frames, size_t frame_stride,
                                                   uint8_t *__restrict__ arena, uint64_t arena_stride, uint32_t off0, int W,
                                                   int H, int stride) {
    const uint32_t f = blockIdx.z;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (y >= H || x0 >= stride) return;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(frames + (size_t)f * frame_stride) + (size_t)y * W;
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (x0 + k < W) {
            const uint32_t p = src[x0 + k];
            o |= (GRAY_IN_R ? (p & 0xff) : gray_of(p)) << (8 * k);
        }
    }
    *reinterpret_cast<uint32_t *>(arena + (uint64_t)f * arena_stride + off0 + (size_t)y * stride + x0) = o;
}

// ccv.grayscale drop-in: RGBA in place, R=G=B=gray, A kept.
__global__ __launch_bounds__(256) void k_gray_inplace(uint8_t *__restrict__ frames, size_t frame_stride, uint32_t npix) {
    uint32_t *p = reinterpret_cast<uint32_t *>(frames + (size_t)blockIdx.y * frame_stride);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const uint32_t v = p[i], g = gray_of(v);
        p[i] = g | (g << 8) | (g << 16) | (v & 0xff000000u);
    }
}

// One generation of drawImage calls.  Workgroup = RS_TW x RS_TH destination pixels of one job, thread = 4 pixels of a row.
//   1. 64 + 16 threads compute the tile's column / row taps of the declared resampler (binary64, explicit
//      __dmul_rn/__dadd_rn so nothing is contracted into an FMA) into LDS: a = floor(f), t = f - a, b = min(a+1, s-1);
//   2. the source rectangle those taps touch (<= ~133 x 35 px for ratios <= 2.04) is staged into LDS with aligned dword
//      loads, all issued before the first LDS write;
//   3. every thread produces 4 pixels from LDS bytes: top/bot/v lerps, v_rndne_f64 (round half to even), one dword store.
// Tiles whose source span does not fit (only the last 1-3 pixel levels, where the ratio can reach 6) read HBM directly.
constexpr int RS_TW = 64;                   // destination tile width; its height is 16 * RPT (RPT rows per thread)
#ifndef HT_RS_PITCH
#define HT_RS_PITCH 160
#endif
constexpr int RS_ROWB = 160;                // LDS source tile: payload bytes per row (64 * 2.04 + 2, dword aligned start; 10 threads x 16 bytes)
constexpr int RS_SP = HT_RS_PITCH;          // ... and its row pitch (a multiple of 16: rows are written as 16-byte chunks)
static_assert(RS_SP >= RS_ROWB && RS_SP % 16 == 0, "RS_SP");
// A workgroup's latency chain (taps -> barrier -> HBM loads -> barrier -> LDS reads -> store) is fixed, so the tile
// height decides how much of it is amortised: with 16 rows (1 row per thread) the 260k waves of one C2 generation run
// in ~32 occupancy rounds of ~4 us each.

typedef unsigned int rs_u32x4 __attribute__((ext_vector_type(4)));
typedef float rs_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 rs_buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const rs_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
typedef HtTap RsTap;  // {t, u: weights of b and a; a, b: source coordinates, absolute incl. the source rect origin}

__device__ __forceinline__ RsTap rs_tap(int i, double r, int s, int origin) {
    double f = __dadd_rn(__dmul_rn((double)i + 0.5, r), -0.5);
    f = f < 0.0 ? 0.0 : f;
    const double fmax = (double)(s - 1);
    f = f > fmax ? fmax : f;
    const double af = floor(f);
    RsTap tp;
    tp.a = origin + (int)af;
    tp.b = origin + min((int)af + 1, s - 1);
    tp.t = __dadd_rn(f, -af);
    tp.u = __dadd_rn(1.0, -tp.t);
    return tp;
}

// 4 destination pixels of one row from two source rows (r0/r1 either in LDS or in HBM: two instantiations, so the loads
// are ds_read_u8 / global_load_ubyte rather than flat loads).  The 4 column taps live in registers (loaded once per thread,
// reused for every row the thread produces).
template <typename PTR>
__device__ __forceinline__ uint32_t rs_pixels4(PTR r0, PTR r1, const RsTap (&cx)[4], const RsTap &ry, int xoff, int npx) {
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k < npx) {
            const int ia = cx[k].a - xoff, ib = cx[k].b - xoff;
            const double top = __dadd_rn(__dmul_rn((double)r0[ia], cx[k].u), __dmul_rn((double)r0[ib], cx[k].t));
            const double bot = __dadd_rn(__dmul_rn((double)r1[ia], cx[k].u), __dmul_rn((double)r1[ib], cx[k].t));
            const double vv = __dadd_rn(__dmul_rn(top, ry.u), __dmul_rn(bot, ry.t));
            const int q = (int)__builtin_rint(vv);  // Uint8ClampedArray: round half to even (values are within [0,255])
            o |= (uint32_t)q << (8 * k);
        }
    }
    return o;
}

// workgroup barrier that orders LDS accesses only: global loads / stores stay in flight across it
#define RS_LDS_BARRIER()                                                     \
    do {                                                                     \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");      \
        __builtin_amdgcn_s_barrier();                                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");      \
    } while (0)
// The right tap of a pair as a relaxed workgroup-scope atomic load: still a plain ds_read_u8 (with the immediate offset folded), but one
// the optimiser does not fuse with its left neighbour.  hipcc otherwise turns p[0], p[1] into ONE ds_read_u16 at an arbitrary byte
// address, and gfx950's LDS serves a misaligned access lane by lane — 64 instead of 2 cycles per wave, measured with
// tools/micro/lds_unaligned_bench.hip (the first build of the binary32 path ran 2.3x slower than the binary64 one because of it).
#define RS_LD1(ptr_) __hip_atomic_load((ptr_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

// LDS variant of a pixel, declared binary64 sequence.  Two facts of the declared resampler make the addressing trivial: b == a + 1 unless
// the coordinate was clamped to the last source sample, and then its weight t is exactly 0, so ANY finite neighbour gives the same rounded
// sum (x * 0 = +0 for every byte x).  The four taps of a pixel are therefore p[0], p[1], p[pitch], p[pitch + 1] from one address
// (ds_read_u8 with immediate offsets) whatever the clamping; s_src carries one spare row for the p[pitch] read of the last staged row.
// Store: round half to even via the 2^52 + 2^51 add (values are within [0, 255]).  p = address of the left taps, p1 = address of the
// right taps (= p + 1 through a value the optimiser cannot see, for the reason given at RS_LD1).
__device__ __forceinline__ uint32_t rs_pixel_f64(const uint8_t *p, const uint8_t *p1, double cu, double ct, double ru, double rt) {
    const double top = __dadd_rn(__dmul_rn((double)p[0], cu), __dmul_rn((double)p1[0], ct));
    const double bot = __dadd_rn(__dmul_rn((double)p[RS_SP], cu), __dmul_rn((double)p1[RS_SP], ct));
    const double vv = __dadd_rn(__dmul_rn(top, ru), __dmul_rn(bot, rt));
    return (uint32_t)__double2loint(__dadd_rn(vv, 6755399441055744.0));
}

// The declared value needs binary64 only where it decides something: the stored byte is RNE(v), and v is needed to far
// less than binary64 precision unless it sits next to a rounding boundary k + 0.5.  So every pixel is first evaluated in
// binary32 — v~ = top + ty * (bot - top), top = p00 + tx * (p01 - p00), bot likewise, three v_fma_f32 — whose distance from
// the declared binary64 value is bounded:
//     tx32 = fl32(tx): |tx32 - tx| <= 2^-25, times |p01 - p00| <= 255                      ->  7.6e-6
//     every binary32 rounding of a value <= 255.x is <= half an ulp of [128, 256) = 2^-17  ->  7.6e-6 each
//     top, bot: 2 terms each (1.52e-5); bot - top: their sum + 1 rounding (3.8e-5); times ty <= 1; + ty32's 7.6e-6
//     + top's 1.52e-5 + the last fma's rounding 7.6e-6                                      =  6.9e-5  < RS_EPS = 2^-13
// (the declared u = fl64(1 - t) differs from 1 - t by <= 2^-54 and the six binary64 roundings of the declaration by < 2e-13:
// both vanish in the margin).  If |v~ - rint(v~)| < 0.5 - RS_EPS the declared value lies strictly between the same two
// boundaries and rounds to the same integer whatever the tie rule; otherwise (2.4e-4 of the pixels of natural images,
// every exact tie of the declaration among them) that pixel is re-evaluated with the declared binary64 sequence.
// An exact 2:1 canvas (both source dimensions even: ccv.js:126-128 halves) is a 2x2 box mean, (a+b+c+d)/4 exactly, whose
// ties (a quarter of all pixels) would all take the fallback: it is computed in integers instead, RNE included.
constexpr float RS_EPS = 1.0f / 8192.0f;

#ifndef HT_RS_WPS
#define HT_RS_WPS 6  // waves per SIMD the register allocator must leave room for (6: <= 80 VGPRs, 5: <= 96)
#endif
// Measured on one box, device ms per step of the pyramid generations 1-5 at 128 x 720p / bench frames/s (profiles/r03_resample_ab.txt):
// round-2 loop 0.503 / 110.9 k; flat loop 6 waves 0.472 / 116.0 k; + packed 0.472 / 116.3 k; flat loop with the next row's taps in flight
// (needs 5 waves: 84 VGPRs) 0.490 / 112.9 k; the same + packed 0.491 / 112.3 k.
#ifndef HT_RS_ROWPF
#define HT_RS_ROWPF 0  // 1 = the 16 taps of the next row are in flight while a row is evaluated (32 tap registers: needs HT_RS_WPS 5), 0 = one row at a time
#endif
#ifndef HT_RS_PACKED
#define HT_RS_PACKED 1  // 1 = top / bot of a pixel as one packed pair (v_pk_add_f32 + v_pk_fma_f32: 4.5 cycles for two, tools/micro/valu_rate_bench.hip)
#endif
#if defined(HT_RS_EXPERIMENT) && HT_RS_EXPERIMENT && !defined(HT_DEBUG_KNOBS)
#error "HT_RS_EXPERIMENT builds compute wrong results by design: only together with -DHT_DEBUG_KNOBS (tools/build_alt.py)"
#endif
#ifndef HT_RS_EXPERIMENT
#define HT_RS_EXPERIMENT 0  // timing experiments only (results are wrong): 1 = no pixel arithmetic, 2 = no source loads after the first frame, 3 = no stores
#endif
#if defined(HT_RS_PHASES)  // tools/gpu_rs_phases.py: shader-clock stamps of every phase of every frame iteration, plain stores into per-workgroup slots
__device__ unsigned long long g_rs_tl[16384][8][8];  // [slot][frame iteration & 7][stamp]
__device__ unsigned long long g_rs_launch[1024][4];  // per launch (keyed by its tiles per frame): workgroups, sum of workgroup lifetimes
__device__ unsigned long long g_rs_tw[4096][4][6];   // per WAVE, third frame iteration of a workgroup: pixel start, pixel end, after barrier 1, after barrier 2, HW_ID
#define RS_WSTAMP(i)                                                                                                                  \
    do {                                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                            \
        if (rs_iter == 2u && (threadIdx.x & 63u) == 0) g_rs_tw[rs_slot & 4095u][threadIdx.x >> 6][i] = __builtin_readcyclecounter(); \
        __builtin_amdgcn_sched_barrier(0);                                                                                            \
    } while (0)
#define HT_RS_TIMELINE 1
#define RS_SUB(i)                                                                       \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        if (threadIdx.x == 0) g_rs_tl[rs_slot][7][i] = __builtin_readcyclecounter();    \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
#define RS_STAMP(i)                                                                     \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        if (threadIdx.x == 0) g_rs_tl[rs_slot][rs_iter & 7u][i] = __builtin_readcyclecounter(); \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
#elif defined(HT_RS_TIMELINE)  // tools/micro/resample_timeline.hip: shader-clock stamps of the phases of every workgroup
__device__ unsigned long long g_rs_timeline[1 << 16][8];
#define RS_STAMP(i)                                                                     \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        if (threadIdx.x == 0) g_rs_timeline[blockIdx.x & 0xffffu][i] = __builtin_readcyclecounter(); \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
#else
#define RS_STAMP(i)
#endif
#ifndef RS_SUB
#define RS_SUB(i)
#endif
#ifndef RS_WSTAMP
#define RS_WSTAMP(i)
#endif
// One workgroup = one tile record x one group of K consecutive frames.  A tile is 64 columns x (16 * np) rows of one
// drawImage call, np <= RPT passes chosen per tile by the host (ht_context.hip) so that (a) a level's rows are split
// evenly — 214 rows are 4 + 4 + 3 + 3 passes, not 4 x 4 with the last tile a third empty — and (b) the source rows the
// tile touches fit the fixed HT_RS_SRC_ROWS-row LDS window (np = 4 at ratio 1.12, 2 at ratio 2).
//
// Measured with tools/micro/resample_timeline.hip: a workgroup that does one tile of one frame is a pure latency chain —
// record + extent + load issue 31 %, waiting for the loads 25 %, pixels 29 % — and a CU holds too few of them to overlap
// it.  Every frame of a batch has the same geometry, so a workgroup keeps its tile for K frames instead: record, source
// extent and tap tables are computed once, and the source tile of frame f+1 is loaded (into registers) while the pixels
// of frame f are computed from LDS.  Frames of a group are consecutive, so they stay inside one XCD's share of the batch.
template <int RPT>
__global__ __launch_bounds__(256, HT_RS_WPS) void k_resample(const HtResampleJob *__restrict__ tiles, uint8_t *__restrict__ arena,
                                                  uint64_t arena_stride, uint32_t blocks_per_frame, uint32_t ngroups,
                                                  uint32_t nframes, uint32_t group_frames) {
    constexpr int TH = 16 * RPT;          // destination rows per tile (at most)
    constexpr int SR = HT_RS_SRC_ROWS;    // LDS source rows
    __shared__ __attribute__((aligned(16))) uint8_t s_src[(SR + 1) * RS_SP + 16];
    __shared__ RsTap s_col[RS_TW], s_row[TH];
    uint32_t gidx, blk;
    if (!xcd_item(blocks_per_frame, ngroups, &gidx, &blk)) return;
#ifndef HT_RS_PRIO
#define HT_RS_PRIO 1  // measured: 1 -> resample -1 % / -1.4 %; 2 (also every frame's LDS write + load issue ahead of the pixel arithmetic): no better
#endif
    if (HT_RS_PRIO) __builtin_amdgcn_s_setprio(3);  // record, extent, first loads, tap tables: ahead of other wavefronts' pixel arithmetic
#ifdef HT_RS_PHASES
    const uint32_t rs_slot = (blockIdx.x ^ (blocks_per_frame * 2654435761u)) & 16383u;  // different launches mostly land in different slots
    uint32_t rs_iter = 0;
#endif
    RS_STAMP(0);
#ifdef HT_RS_PHASES
    const unsigned long long rs_t_entry = __builtin_readcyclecounter();
    auto rs_exit = [&]() {
        if (threadIdx.x == 0) {
            const unsigned long long now = __builtin_readcyclecounter();
            unsigned long long *L = g_rs_launch[blocks_per_frame & 1023u];
            atomicAdd(&L[0], 1ull);
            atomicAdd(&L[1], now - rs_t_entry);
        }
    };
#endif
    const HtResampleJob J = tiles[blk];  // by value: the whole record in a few wide scalar loads, ONE round trip (as a reference its fields were fetched piecemeal, eight dependent s_load round trips along the prologue)
    const uint32_t f0 = gidx * group_frames, f1 = min(f0 + group_frames, nframes);
    const int tid = (int)threadIdx.x;
    const int np = (int)J.np;
    const int X0 = (int)J.bx * RS_TW, Y0 = (int)J.pass0 * 16;
    uint8_t *frame = arena + (uint64_t)f0 * arena_stride;
    const int ncols = min(RS_TW, J.dw - X0), nrows = min(16 * np, J.dh - Y0);  // drawn part of this tile (may be <= 0)
    const int x0 = X0 + (tid & 15) * 4, yt = Y0 + (tid >> 4);                  // this thread: rows yt, yt+16, ...
    const int npx = min(4, J.dw - x0);
    const bool drawn = ncols > 0 && nrows > 0;
    RS_SUB(0);
    int xa = 0, ya = 0, sw16 = 1, sh = 1;
    bool in_lds = false;
    if (drawn) {
        if (J.ex_sw16 > 0) {  // from the host (ht_context.hip)
            xa = J.ex_xa, ya = J.ex_ya, sw16 = J.ex_sw16, sh = J.ex_sh;
        } else {
            xa = rs_tap(X0, J.rx, J.sw, J.sx).a & ~15;
            ya = rs_tap(Y0, J.ry, J.sh, J.sy).a;
            sw16 = (rs_tap(X0 + ncols - 1, J.rx, J.sw, J.sx).b - xa) / 16 + 1;  // 16-byte chunks per source row
            sh = rs_tap(Y0 + nrows - 1, J.ry, J.sh, J.sy).b - ya + 1;           // source rows
        }
        // workgroup-uniform whichever branch produced them (the device path computes them with vector binary64 instructions): scalar registers
        xa = __builtin_amdgcn_readfirstlane(xa), ya = __builtin_amdgcn_readfirstlane(ya);
        sw16 = __builtin_amdgcn_readfirstlane(sw16), sh = __builtin_amdgcn_readfirstlane(sh);
        in_lds = (sw16 * 16 <= RS_ROWB) && (sh <= SR);
    }
    RS_SUB(1);
    if (in_lds) {
        // source rows as 16-byte chunks.  10 threads share a row (= the 160-byte LDS pitch), 25 rows per pass, 3 passes (SR = 75 rows).
        // Plane strides are only 4-byte multiples, so the chunks are dword- not 16-byte-aligned in HBM and may run past the row's
        // end into the next row / plane of the same arena (never used: a clamped tap has weight 0, see rs_pixel_f64).  Loads are
        // unconditional with clamped coordinates (duplicates fall into cache lines the wave fetches anyway); only the LDS writes
        // are predicated.
        //
        // Frame loop (round 3).  The loop of rounds 1-2 compiled to one basic block PER PIXEL (`if (k < npx)`, the mode tests and
        // the row tests were re-evaluated per pixel / per row): the four ds_read_u8 of a pixel were issued and waited for inside
        // that pixel's block, so a frame iteration was a chain of 16 dependent LDS round trips per thread, and the row taps were
        // re-read from LDS (with a v_cvt_f32_f64 and a v_mul_lo_u32) for every row of every frame — ~27 VALU instructions per pixel,
        // which is why the kernel "did not respond to instruction counts".  Now
        //   * everything that depends only on the tile geometry is computed ONCE per workgroup (row offsets, row weights as
        //     binary32, byte mask, store predicates);
        //   * the loop is specialised on the number of passes (NP) and on the box mode, and a row is straight-line code: every tap
        //     is loaded from CLAMPED coordinates whether or not its pixel is drawn (the byte mask zeroes what lies outside dw x dh),
        //     and the 16 taps of row q + 1 are in flight while row q is evaluated (~16 VALU instructions per pixel);
        //   * global memory is addressed through a buffer descriptor of the frame (scalar registers) + 32-bit per-thread offsets +
        //     a scalar offset: no 64-bit per-thread address arithmetic, nothing for loop strength reduction to turn into per-thread
        //     induction variables;
        //   * the staging registers of the prefetched source tile are live only inside an iteration (load issue -> LDS write), not
        //     across the tap tables and the variant switch, where the register allocator used to spill them.
        constexpr int KR = (SR + 24) / 25;
        static_assert(KR == 3, "the staging registers below are written out for three passes");
        const int r0 = (tid * 205) >> 11, c16 = tid - r0 * 10;  // tid / 10, tid % 10 for tid < 256
        const bool lane_on = (r0 < 25) && (c16 < sw16);
        // addresses = the frame's buffer descriptor (scalar registers) + 32-bit per-thread offsets
        const uint32_t soff = J.src_off + (uint32_t)(ya * J.src_stride + xa + 16 * min(c16, sw16 - 1));
        const uint32_t soff0 = soff + (uint32_t)(min(r0, sh - 1) * J.src_stride), soff1 = soff + (uint32_t)(min(r0 + 25, sh - 1) * J.src_stride);
        const uint32_t soff2 = soff + (uint32_t)(min(r0 + 50, sh - 1) * J.src_stride);
        uint8_t *const sdst = &s_src[r0 * RS_SP + 16 * c16];
        const bool w0on = lane_on && r0 < sh, w1on = lane_on && r0 + 25 < sh, w2on = lane_on && r0 + 50 < sh;
        uint64_t fbase = reinterpret_cast<uint64_t>(frame);
        // raw buffer (stride 0) over a frame's arena; 0x00020000 = the gfx9 data-format word; offsets stay inside the arena
#define RS_FRAME_RSRC(base_) __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base_), 0, (int)0xffffffffu, 0x00020000)
#define RS_TILE_TO_LDS()                                                          \
    do {                                                                          \
        if (w0on) *reinterpret_cast<uint4 *>(sdst) = v0;                          \
        if (w1on) *reinterpret_cast<uint4 *>(sdst + 25 * RS_SP) = v1;             \
        if (w2on) *reinterpret_cast<uint4 *>(sdst + 50 * RS_SP) = v2;             \
    } while (0)
        RS_SUB(2);
        {
            // the first frame's tile: loads issued, tap tables computed while they are in flight, then tile + tables behind ONE barrier
            const __amdgpu_buffer_rsrc_t fr = RS_FRAME_RSRC(fbase);
            const uint4 v0 = rs_buf_load16(fr, soff0, 0u), v1 = rs_buf_load16(fr, soff1, 0u), v2 = rs_buf_load16(fr, soff2, 0u);
            RS_SUB(3);
            if (tid < ncols) s_col[tid] = rs_tap(X0 + tid, J.rx, J.sw, J.sx);
            if (tid >= 64 && tid - 64 < nrows) s_row[tid - 64] = rs_tap(Y0 + tid - 64, J.ry, J.sh, J.sy);
            RS_SUB(4);
            RS_TILE_TO_LDS();
        }
        __syncthreads();
        if (HT_RS_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        RS_STAMP(1);
        const int cbase = x0 - X0;  // first column of this thread inside the tile
        int ia[4];
        float ctf[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const RsTap *ct = &s_col[min(cbase + k, ncols - 1)];
            ia[k] = ct->a - xa, ctf[k] = (float)ct->t;
        }
        if (HT_RS_EXPERIMENT == 4) {  // timing experiment (wrong results): conflict-free tap addresses — what would halving the LDS cycles buy?
#pragma unroll
            for (int k = 0; k < 4; k++) ia[k] = (tid & 31) * 4 + ((tid >> 5) & 1) * 1280 + k * 160 * 2;
        }
        // the right taps are read at p + 1 — with a "1" the optimiser cannot see: hipcc otherwise fuses p[0] and p[1] into ONE
        // ds_read_u16 at an arbitrary byte address (see rs_pixel_f64)
        uint32_t one = 1u;
        asm volatile("" : "+s"(one));
        const uint32_t mode = J.pad;  // bit 0: 2x2 box mean (both ratios exactly 2), bit 1: binary64 everywhere (set by the host)
        // binary64 everywhere (option rs_nofast) = every pixel takes the fallback: a threshold no distance can stay under
        const float thr = (mode & 2u) ? -1.0f : 0.5f - RS_EPS;
        const int dh = J.dh, ch = J.ch, dst_stride = J.dst_stride;
        const uint32_t doff = J.dst_off + (uint32_t)(yt * dst_stride + x0);
        const uint32_t pxmask = npx >= 4 ? 0xffffffffu : (npx <= 0 ? 0u : ((1u << (8 * npx)) - 1u));
        const int ia8 = npx > 0 ? ia[0] : 0;  // box mode: 8 consecutive source bytes of the thread's 4 pixels (8-byte aligned: x0 % 4 == 0, xa % 16 == 0, sx == 0)
        auto frames = [&](auto NPc, auto BOXc) {
            constexpr int NP = decltype(NPc)::value;
            constexpr bool BOX = decltype(BOXc)::value;
            uint32_t roff[NP];
            float rtf[NP];
            bool st[NP], rv[NP];
#pragma unroll
            for (int q = 0; q < NP; q++) {
                const int y = yt + 16 * q;
                const RsTap *ry = &s_row[min(y - Y0, nrows - 1)];
                roff[q] = HT_RS_EXPERIMENT == 4 ? (uint32_t)(q * 3 * RS_SP) : (uint32_t)((ry->a - ya) * RS_SP);
                rtf[q] = (float)ry->t;
                rv[q] = y < dh;  // a drawn row (pxmask is 0 where the thread has no drawn column)
                st[q] = y < ch && x0 < dst_stride;
            }
            for (uint32_t f = f0; f < f1; f++) {
                asm volatile("" : "+s"(fbase));  // opaque per iteration: the frame base stays ONE scalar value
                const __amdgpu_buffer_rsrc_t fr = RS_FRAME_RSRC(fbase);
                fbase += arena_stride;
                // the tap addresses roff[q] + ia[k] are loop invariants, but 32 of them do not fit the register budget: keep them from being hoisted
#pragma unroll
                for (int q = 0; q < NP; q++) asm volatile("" : "+v"(roff[q]));
                RS_STAMP(2);
                uint4 v0, v1, v2;
                if (f + 1 < f1 && HT_RS_EXPERIMENT != 2) {  // the next frame's source tile (scalar offset = one arena): in flight during this frame's pixels
                    v0 = rs_buf_load16(fr, soff0, (uint32_t)arena_stride);
                    v1 = rs_buf_load16(fr, soff1, (uint32_t)arena_stride);
                    v2 = rs_buf_load16(fr, soff2, (uint32_t)arena_stride);
                }
                RS_STAMP(3);
                RS_WSTAMP(0);
                uint32_t o[NP];
                if (HT_RS_EXPERIMENT == 1) {  // timing experiment (wrong results): no pixel arithmetic, one LDS read per row
#pragma unroll
                    for (int q = 0; q < NP; q++) o[q] = *reinterpret_cast<const uint32_t *>(s_src + roff[q] + (ia8 & ~3));
                } else if (BOX) {
                    // exact 2:1 in both directions: (a + b + c + d) / 4 with round half to even.  The 4 pixels of a thread are 8 consecutive
                    // bytes of two source rows: two ds_read_b64 per row instead of 16 byte reads; v_perm_b32 gathers a pixel's four bytes,
                    // v_sad_u8 adds them, and RNE(sum / 4) is exact in binary32 (sum <= 1020).
#pragma unroll
                    for (int q = 0; q < NP; q++) {
                        const uint8_t *row = s_src + roff[q] + ia8;
                        const uint2 A = *reinterpret_cast<const uint2 *>(row), B = *reinterpret_cast<const uint2 *>(row + RS_SP);
                        uint32_t oq = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t a = k < 2 ? A.x : A.y, b = k < 2 ? B.x : B.y;
                            // v_perm_b32(src0 = b, src1 = a): selector bytes 0-3 pick from a, 4-7 from b
                            const uint32_t g = __builtin_amdgcn_perm(b, a, (k & 1) ? 0x07060302u : 0x05040100u);
                            const uint32_t sum = __builtin_amdgcn_sad_u8(g, 0u, 0u);
                            oq = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf((float)sum * 0.25f), k, oq);
                        }
                        o[q] = rv[q] ? (oq & pxmask) : 0u;
                    }
                } else {
                    // The declared value needs binary64 only where it decides something (see RS_EPS above): every pixel is evaluated in
                    // binary32 first — top = p00 + tx * (p01 - p00), bot likewise, v = top + ty * (bot - top): three v_fma_f32 —, stored
                    // with v_cvt_pk_u8_f32 (its input is already integral), and the distance to the nearest rounding boundary is tested once
                    // per row (max of the four |v - r|).
                    uint32_t T[2][16];
#define RS_TAPS(q_, buf_)                                                                                   \
    do {                                                                                                    \
        const uint8_t *row_ = s_src + roff[q_];                                                             \
        _Pragma("unroll") for (int k = 0; k < 4; k++) {                                                     \
            const uint8_t *p_ = row_ + ia[k];                                                               \
            T[buf_][4 * k] = p_[0], T[buf_][4 * k + 1] = RS_LD1(p_ + 1), T[buf_][4 * k + 2] = p_[RS_SP], T[buf_][4 * k + 3] = RS_LD1(p_ + RS_SP + 1); \
        }                                                                                                   \
    } while (0)
                    if (HT_RS_ROWPF) {
                        RS_TAPS(0, 0);
                        __builtin_amdgcn_sched_barrier(0);  // row 0's taps first: its arithmetic only waits for them
                    }
#pragma unroll
                    for (int q = 0; q < NP; q++) {
                        if (HT_RS_ROWPF) {
                            if (q + 1 < NP) RS_TAPS(q + 1, (q + 1) & 1);
                        } else {
                            RS_TAPS(q, q & 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);  // the loads above are not sunk into the arithmetic below
                        float d[4];
                        uint32_t oq = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const float p00 = (float)T[q & 1][4 * k], p01 = (float)T[q & 1][4 * k + 1], p10 = (float)T[q & 1][4 * k + 2], p11 = (float)T[q & 1][4 * k + 3];
#if HT_RS_PACKED  // top and bot of a pixel as one packed pair: v_pk_add_f32 + v_pk_fma_f32 instead of two v_sub_f32 + two v_fma_f32
                            const rs_f2 lo2 = {p00, p10}, hi2 = {p01, p11}, ct2 = {ctf[k], ctf[k]};
                            const rs_f2 tb = __builtin_elementwise_fma(ct2, hi2 - lo2, lo2);
                            const float top = tb.x, bot = tb.y;
#else
                            const float top = __builtin_fmaf(ctf[k], p01 - p00, p00);
                            const float bot = __builtin_fmaf(ctf[k], p11 - p10, p10);
#endif
                            const float v = __builtin_fmaf(rtf[q], bot - top, top);
                            const float r = __builtin_rintf(v);
                            d[k] = v - r;
                            oq = __builtin_amdgcn_cvt_pk_u8_f32(r, k, oq);
                        }
                        const float dmq = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d[0]), __builtin_fabsf(d[1])), __builtin_fmaxf(__builtin_fabsf(d[2]), __builtin_fabsf(d[3])));
                        if (dmq >= thr) {  // rare (2.4e-4 of the pixels): the declared binary64 sequence for the pixels next to a rounding boundary
                            // everything the fallback needs is derived HERE from values made opaque inside the branch: as loop invariants its
                            // table addresses would be hoisted in front of the loop and held in registers the common path needs
                            uint32_t roff_f = roff[q];
                            int yrel = yt - Y0, cb = cbase;
                            asm volatile("" : "+v"(roff_f), "+v"(yrel), "+v"(cb));  // (also: the fallback re-reads its taps instead of keeping all 16 bytes of every row alive)
                            const RsTap *ry = &s_row[min(yrel + 16 * q, nrows - 1)];
                            const uint8_t *rowf = s_src + roff_f;
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                if (__builtin_fabsf(d[k]) >= thr) {
                                    const RsTap *ct = &s_col[min(cb + k, ncols - 1)];
                                    oq &= ~(0xffu << (8 * k));
                                    oq |= rs_pixel_f64(rowf + ia[k], rowf + ia[k] + one, ct->u, ct->t, ry->u, ry->t) << (8 * k);
                                }
                            }
                        }
                        o[q] = rv[q] ? (oq & pxmask) : 0u;
                    }
#undef RS_TAPS
                }
#ifdef HT_RS_TIMELINE
#pragma unroll
                for (int q = 0; q < NP; q++) asm volatile("" ::"v"(o[q]));
                RS_STAMP(4);
                RS_WSTAMP(1);
#endif
                // Barriers inside the loop order LDS accesses only (fence restricted to the local address space: s_waitcnt lgkmcnt(0) + s_barrier).
                // __syncthreads() also waits for vmcnt(0), i.e. for this iteration's global STORES to be acknowledged and for the prefetched
                // loads: measured (HT_RS_PHASES, 720p) a workgroup spent 8 300 of its 11 500 cycles per frame parked there.  The stores are
                // issued after the next tile has been written to LDS, so whoever waits for loads next (vmcnt cannot tell loads from older
                // stores) finds them a whole pixel phase old.
                // pixels outside the drawn dw x dh rect stay transparent black (ccv.js:135-145 draws 2 px short on the variants)
#define RS_STORE_ROWS()                                                                                                                             \
    _Pragma("unroll") for (int q = 0; q < NP; q++) if (st[q] && (HT_RS_EXPERIMENT != 3 || o[q] == 0x12345678u))                                     \
        __builtin_amdgcn_raw_buffer_store_b32(o[q], fr, doff, (uint32_t)(16 * q * dst_stride), 0)
                if (f + 1 < f1) {
                    RS_LDS_BARRIER();  // every wave is done reading this frame's tile
                    RS_STAMP(5);
                    RS_WSTAMP(2);
                    if (HT_RS_EXPERIMENT != 2) RS_TILE_TO_LDS();
                    RS_STORE_ROWS();
                    RS_LDS_BARRIER();
                    RS_WSTAMP(3);
#ifdef HT_RS_PHASES
                    if (rs_iter == 2u && (threadIdx.x & 63u) == 0) g_rs_tw[rs_slot & 4095u][threadIdx.x >> 6][4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
#endif
                } else {
                    RS_STORE_ROWS();
                    RS_STAMP(5);
                }
#undef RS_STORE_ROWS
                RS_STAMP(6);
#ifdef HT_RS_PHASES
                rs_iter++;
#endif
            }
        };
        using std::integral_constant;
        if (mode & 1u) {
            switch (np) {
                case 1: frames(integral_constant<int, 1>{}, integral_constant<bool, true>{}); break;
                case 2: frames(integral_constant<int, 2>{}, integral_constant<bool, true>{}); break;
                case 3: frames(integral_constant<int, 3>{}, integral_constant<bool, true>{}); break;
                default: frames(integral_constant<int, RPT>{}, integral_constant<bool, true>{}); break;
            }
        } else {
            switch (np) {
                case 1: frames(integral_constant<int, 1>{}, integral_constant<bool, false>{}); break;
                case 2: frames(integral_constant<int, 2>{}, integral_constant<bool, false>{}); break;
                case 3: frames(integral_constant<int, 3>{}, integral_constant<bool, false>{}); break;
                default: frames(integral_constant<int, RPT>{}, integral_constant<bool, false>{}); break;
            }
        }
#undef RS_TILE_TO_LDS
#undef RS_FRAME_RSRC
#ifdef HT_RS_PHASES
        rs_exit();
#endif
        return;
    }
    // nothing drawn in this tile (transparent black), or a source span larger than the LDS window (ratios > 2.3: the
    // last 1-3 pixel levels): taps straight from HBM
    const HtResampleJob &Jm = tiles[blk];  // this (rare) path reads the record from memory field by field: the by-value copy above would
                                             // keep 22 scalar registers alive across its binary64 loop and push it into scratch
    RsTap cx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cx[k] = rs_tap(min(x0 + k, X0 + max(ncols, 1) - 1), Jm.rx, Jm.sw, Jm.sx);
    for (uint32_t f = f0; f < f1; f++, frame += arena_stride) {
        const uint8_t *src = frame + Jm.src_off;
        int ytl = yt;
        asm volatile("" : "+v"(ytl));  // opaque per frame: the row taps and addresses of all four passes are loop invariants, and hoisted they do not fit the register budget
#pragma unroll
        for (int q = 0; q < RPT; q++) {
            const int y = ytl + 16 * q;
            uint32_t o = 0;
            if (drawn && q < np && y < Jm.dh && npx > 0) {
                const RsTap ry = rs_tap(y, Jm.ry, Jm.sh, Jm.sy);
                o = rs_pixels4<const uint8_t *>(src + (size_t)ry.a * Jm.src_stride, src + (size_t)ry.b * Jm.src_stride, cx, ry, 0, npx);
            }
            if (q < np && y < Jm.ch && x0 < Jm.dst_stride) *reinterpret_cast<uint32_t *>(frame + Jm.dst_off + (size_t)y * Jm.dst_stride + x0) = o;
        }
    }
}

// k_resample_bands (round 6): the same tiles, pixels and arithmetic as k_resample with the source staged by LDS-DMA into WAVE-PRIVATE bands.
//
// k_resample's frame iteration stages the next tile through 12 VGPRs per thread and needs two workgroup barriers (everybody done reading
// -> registers to LDS -> everybody done writing); its phase stamps (profiles/r05_rs_phases_c2.txt) put 2 580 of an iteration's 5 708 cycles
// into "load wait + registers -> LDS + barrier", with the four wavefronts reaching barrier 1 spread over ~1 190 cycles.  Here a wavefront owns
// a contiguous quarter of the tile's destination rows (4 * np rows: np passes of 4 rows x 16 column groups) and with it the band of source
// rows those rows touch (<= 24 rows of the 160-byte LDS pitch = 4 KB per wavefront; neighbouring bands overlap by 1 - 2 rows, which are
// fetched twice).  A band is filled by four `buffer_load_dwordx4 ... lds` instructions (gfx950: 16 bytes per lane, 1 KB per instruction,
// M0 = the band's LDS address; lanes outside the band are masked off) — no staging registers, no ds_write — and nobody else ever reads it:
// the frame loop has NO workgroup barrier, a wavefront only waits for its own loads (s_waitcnt vmcnt) while the CU's other wavefronts
// compute.  16 KB of bands + 3 KB of tap tables per workgroup and <= 64 VGPRs: 8 workgroups per CU (k_resample: 7 by registers).
// The host fills in each tile record's band extents (ht_context.hip, the binary64 operations of rs_tap); a tile whose bands do not fit takes
// the HBM-tap path exactly like k_resample's.
#ifndef HT_RSB_WPS
#define HT_RSB_WPS 8
#endif
constexpr int RSB_BAND = 4096;  // LDS bytes per wavefront = four 1 KB LDS-DMA instructions = 25.6 rows of RS_SP bytes
#if defined(HT_RS_PHASES)  // tools/gpu_rsb_phases.py: shader-clock sums per phase of a WAVEFRONT's frame iteration (there is no workgroup phase in the loop)
__device__ unsigned long long g_rsb_ph[256][8];  // per shard (blockIdx & 255): [0] iterations, [1] pixels, [2] DMA issue, [3] wait for the band, [4] stores, [5] workgroups, [6] record -> first loop top
// sums are kept in registers and added ONCE per wavefront at its exit: an atomic per phase and iteration sits in the vector-memory queue in front of the
// next iteration's s_waitcnt vmcnt(0) and was what the first version of these stamps measured ("wait 155 k cycles")
#define RSB_T(var_)                                     \
    do {                                                \
        __builtin_amdgcn_sched_barrier(0);              \
        var_ = __builtin_readcyclecounter();            \
        __builtin_amdgcn_sched_barrier(0);              \
    } while (0)
#else
#define RSB_T(var_)
#endif
typedef __attribute__((address_space(3))) void rs_lds_void;
template <int RPT>
__global__ __launch_bounds__(256, HT_RSB_WPS) void k_resample_bands(const HtResampleJob *__restrict__ tiles, uint8_t *__restrict__ arena,
                                                                    uint64_t arena_stride, uint32_t blocks_per_frame, uint32_t ngroups,
                                                                    uint32_t nframes, uint32_t group_frames) {
    constexpr int TH = 16 * RPT;
    __shared__ __attribute__((aligned(16))) uint8_t s_src[4 * RSB_BAND];
    __shared__ RsTap s_col[RS_TW], s_row[TH];
    uint32_t gidx, blk;
    if (!xcd_item(blocks_per_frame, ngroups, &gidx, &blk)) return;
#ifdef HT_RS_PHASES
    unsigned long long rsb_t_entry = 0, rsb_t0 = 0, rsb_t1 = 0, rsb_t2 = 0, rsb_t3 = 0, rsb_t4 = 0, rsb_acc[7] = {0, 0, 0, 0, 0, 0, 0};
    RSB_T(rsb_t_entry);
#endif
    const HtResampleJob J = tiles[blk];
    const uint32_t f0 = gidx * group_frames, f1 = min(f0 + group_frames, nframes);
    const int tid = (int)threadIdx.x;
    const int np = (int)J.np;
    const int X0 = (int)J.bx * RS_TW, Y0 = (int)J.pass0 * 16;
    uint8_t *frame = arena + (uint64_t)f0 * arena_stride;
    const int ncols = min(RS_TW, J.dw - X0), nrows = min(16 * np, J.dh - Y0);
    const uint32_t mode = J.pad;  // bit 0: 2x2 box mean, bit 1: binary64 everywhere, bit 2: the four bands fit (host)
    if ((mode & 4u) && ncols > 0 && nrows > 0) {
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
        const int xa = J.ex_xa, sw16 = J.ex_sw16;
        const int yaw = J.ex_ya + (int)((J.band_ya4 >> (8 * w)) & 0xffu);  // first source row of this wavefront's band
        const int bsh = (int)((J.band_sh4 >> (8 * w)) & 0xffu);             // ... and its rows (<= 24)
        uint8_t *const band = s_src + w * RSB_BAND;
        // lane -> 16-byte chunk of the band: instruction i fills chunks [64 i, 64 i + 64), chunk c = row c / 10, column c % 10
        uint32_t voff[4];
        bool on[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = i * 64 + l, row = (c * 205) >> 11, col = c - row * 10;
            on[i] = col < sw16 && row < bsh;
            voff[i] = J.src_off + (uint32_t)((yaw + row) * J.src_stride + xa + 16 * col);
        }
        uint64_t fbase = reinterpret_cast<uint64_t>(frame);
#define RS_FRAME_RSRC(base_) __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base_), 0, (int)0xffffffffu, 0x00020000)
#define RSB_DMA(fr_, soff_)                                                                                                              \
    do {                                                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; i++) if (on[i])                                                                         \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(fr_, (rs_lds_void *)(band + i * 1024), 16, voff[i], soff_, 0, 0);                   \
    } while (0)
        {
            const __amdgpu_buffer_rsrc_t fr = RS_FRAME_RSRC(fbase);
            RSB_DMA(fr, 0u);  // the first frame's bands: in flight while the tap tables are computed
            if (tid < ncols) s_col[tid] = rs_tap(X0 + tid, J.rx, J.sw, J.sx);
            if (tid >= 64 && tid - 64 < nrows) s_row[tid - 64] = rs_tap(Y0 + tid - 64, J.ry, J.sh, J.sy);
        }
        __syncthreads();  // tap tables (and, vmcnt(0), this wavefront's band): the only workgroup barrier
        const int cbase = (l & 15) * 4, x0 = X0 + cbase;
        const int yt = Y0 + 4 * np * w + (l >> 4);  // this thread's rows: yt, yt + 4, ... (np of them)
        const int npx = min(4, J.dw - x0);
        int ia[4];
        float ctf[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const RsTap *ct = &s_col[min(cbase + k, ncols - 1)];
            ia[k] = ct->a - xa, ctf[k] = (float)ct->t;
        }
        uint32_t one = 1u;
        asm volatile("" : "+s"(one));  // see RS_LD1
        const float thr = (mode & 2u) ? -1.0f : 0.5f - RS_EPS;
        const int dh = J.dh, ch = J.ch, dst_stride = J.dst_stride;
        const uint32_t doff = J.dst_off + (uint32_t)(yt * dst_stride + x0);
        const uint32_t pxmask = npx >= 4 ? 0xffffffffu : (npx <= 0 ? 0u : ((1u << (8 * npx)) - 1u));
        const int ia8 = npx > 0 ? ia[0] : 0;
        auto frames = [&](auto NPc, auto BOXc) {
            constexpr int NP = decltype(NPc)::value;
            constexpr bool BOX = decltype(BOXc)::value;
            uint32_t roff[NP];
            float rtf[NP];
            bool st[NP], rv[NP];
#pragma unroll
            for (int q = 0; q < NP; q++) {
                const int y = yt + 4 * q;
                const RsTap *ry = &s_row[min(y - Y0, nrows - 1)];
                roff[q] = (uint32_t)((ry->a - yaw) * RS_SP + w * RSB_BAND);
                rtf[q] = (float)ry->t;
                rv[q] = y < dh;
                st[q] = y < ch && x0 < dst_stride;
            }
            for (uint32_t f = f0; f < f1; f++) {
                asm volatile("" : "+s"(fbase));
                const __amdgpu_buffer_rsrc_t fr = RS_FRAME_RSRC(fbase);
                fbase += arena_stride;
#pragma unroll
                for (int q = 0; q < NP; q++) asm volatile("" : "+v"(roff[q]));
#ifdef HT_RS_PHASES
                if (f == f0 && w == 0) rsb_acc[5] = 1ull, rsb_acc[6] = __builtin_readcyclecounter() - rsb_t_entry;
#endif
                RSB_T(rsb_t0);
                uint32_t o[NP];
                if (BOX) {
#pragma unroll
                    for (int q = 0; q < NP; q++) {
                        const uint8_t *row = s_src + roff[q] + ia8;
                        const uint2 A = *reinterpret_cast<const uint2 *>(row), B = *reinterpret_cast<const uint2 *>(row + RS_SP);
                        uint32_t oq = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t a = k < 2 ? A.x : A.y, b = k < 2 ? B.x : B.y;
                            const uint32_t g = __builtin_amdgcn_perm(b, a, (k & 1) ? 0x07060302u : 0x05040100u);
                            const uint32_t sum = __builtin_amdgcn_sad_u8(g, 0u, 0u);
                            oq = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf((float)sum * 0.25f), k, oq);
                        }
                        o[q] = rv[q] ? (oq & pxmask) : 0u;
                    }
                } else {
                    uint32_t T[16];
#pragma unroll
                    for (int q = 0; q < NP; q++) {
                        const uint8_t *row_ = s_src + roff[q];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint8_t *p_ = row_ + ia[k];
                            T[4 * k] = p_[0], T[4 * k + 1] = RS_LD1(p_ + 1), T[4 * k + 2] = p_[RS_SP], T[4 * k + 3] = RS_LD1(p_ + RS_SP + 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        float d[4];
                        uint32_t oq = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const float p00 = (float)T[4 * k], p01 = (float)T[4 * k + 1], p10 = (float)T[4 * k + 2], p11 = (float)T[4 * k + 3];
                            const rs_f2 lo2 = {p00, p10}, hi2 = {p01, p11}, ct2 = {ctf[k], ctf[k]};
                            const rs_f2 tb = __builtin_elementwise_fma(ct2, hi2 - lo2, lo2);
                            const float top = tb.x, bot = tb.y;
                            const float v = __builtin_fmaf(rtf[q], bot - top, top);
                            const float r = __builtin_rintf(v);
                            d[k] = v - r;
                            oq = __builtin_amdgcn_cvt_pk_u8_f32(r, k, oq);
                        }
                        const float dmq = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d[0]), __builtin_fabsf(d[1])), __builtin_fmaxf(__builtin_fabsf(d[2]), __builtin_fabsf(d[3])));
                        if (dmq >= thr) {  // rare: the declared binary64 sequence next to a rounding boundary (see k_resample)
                            uint32_t roff_f = roff[q];
                            int yrel = yt - Y0, cb = cbase;
                            asm volatile("" : "+v"(roff_f), "+v"(yrel), "+v"(cb));
                            const RsTap *ry = &s_row[min(yrel + 4 * q, nrows - 1)];
                            const uint8_t *rowf = s_src + roff_f;
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                if (__builtin_fabsf(d[k]) >= thr) {
                                    const RsTap *ct = &s_col[min(cb + k, ncols - 1)];
                                    oq &= ~(0xffu << (8 * k));
                                    oq |= rs_pixel_f64(rowf + ia[k], rowf + ia[k] + one, ct->u, ct->t, ry->u, ry->t) << (8 * k);
                                }
                            }
                        }
                        o[q] = rv[q] ? (oq & pxmask) : 0u;
                    }
                }
#ifdef HT_RS_PHASES
#pragma unroll
                for (int q = 0; q < NP; q++) asm volatile("" ::"v"(o[q]));
#endif
                RSB_T(rsb_t1);
                if (f + 1 < f1) {
                    // every LDS read of this frame's band has returned (o[] depends on all of them): the band may be overwritten.  The wait
                    // below is for the DMA alone — this frame's stores are issued behind it, the previous frame's are a pixel phase old.
                    // s_waitcnt through the builtin: the compiler's own counter tracking sees it (behind an inline-asm wait it adds a second
                    // vmcnt(0) in front of the next frame's first LDS read)
                    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
                    RSB_DMA(fr, (uint32_t)arena_stride);
                    RSB_T(rsb_t2);
                    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
                    RSB_T(rsb_t3);
                }
#pragma unroll
                for (int q = 0; q < NP; q++)
                    if (st[q]) __builtin_amdgcn_raw_buffer_store_b32(o[q], fr, doff, (uint32_t)(4 * q * dst_stride), 0);
#ifdef HT_RS_PHASES
                RSB_T(rsb_t4);
                if (f + 1 < f1) {  // one sample per wavefront and frame iteration (the last one of a group has no DMA phase)
                    rsb_acc[0] += 1ull, rsb_acc[1] += rsb_t1 - rsb_t0, rsb_acc[2] += rsb_t2 - rsb_t1, rsb_acc[3] += rsb_t3 - rsb_t2, rsb_acc[4] += rsb_t4 - rsb_t3;
                }
#endif
            }
#ifdef HT_RS_PHASES
            if (l == 0) {
#pragma unroll
                for (int i = 0; i < 7; i++)
                    if (rsb_acc[i]) atomicAdd(&g_rsb_ph[blockIdx.x & 255u][i], rsb_acc[i]);
            }
#endif
        };
        using std::integral_constant;
        if (mode & 1u) {
            switch (np) {
                case 1: frames(integral_constant<int, 1>{}, integral_constant<bool, true>{}); break;
                case 2: frames(integral_constant<int, 2>{}, integral_constant<bool, true>{}); break;
                case 3: frames(integral_constant<int, 3>{}, integral_constant<bool, true>{}); break;
                default: frames(integral_constant<int, RPT>{}, integral_constant<bool, true>{}); break;
            }
        } else {
            switch (np) {
                case 1: frames(integral_constant<int, 1>{}, integral_constant<bool, false>{}); break;
                case 2: frames(integral_constant<int, 2>{}, integral_constant<bool, false>{}); break;
                case 3: frames(integral_constant<int, 3>{}, integral_constant<bool, false>{}); break;
                default: frames(integral_constant<int, RPT>{}, integral_constant<bool, false>{}); break;
            }
        }
#undef RSB_DMA
#undef RS_FRAME_RSRC
        return;
    }
    // nothing drawn in this tile (transparent black), or bands that do not fit: taps straight from HBM, k_resample's thread -> pixel map
    const HtResampleJob &Jm = tiles[blk];
    const int x0 = X0 + (tid & 15) * 4, yt = Y0 + (tid >> 4);
    const int npx = min(4, Jm.dw - x0);
    const bool drawn = ncols > 0 && nrows > 0;
    RsTap cx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cx[k] = rs_tap(min(x0 + k, X0 + max(ncols, 1) - 1), Jm.rx, Jm.sw, Jm.sx);
    for (uint32_t f = f0; f < f1; f++, frame += arena_stride) {
        const uint8_t *src = frame + Jm.src_off;
        int ytl = yt;
        asm volatile("" : "+v"(ytl));
#pragma unroll
        for (int q = 0; q < RPT; q++) {
            const int y = ytl + 16 * q;
            uint32_t o = 0;
            if (drawn && q < np && y < Jm.dh && npx > 0) {
                const RsTap ry = rs_tap(y, Jm.ry, Jm.sh, Jm.sy);
                o = rs_pixels4<const uint8_t *>(src + (size_t)ry.a * Jm.src_stride, src + (size_t)ry.b * Jm.src_stride, cx, ry, 0, npx);
            }
            if (q < np && y < Jm.ch && x0 < Jm.dst_stride) *reinterpret_cast<uint32_t *>(frame + Jm.dst_off + (size_t)y * Jm.dst_stride + x0) = o;
        }
    }
}

// The last generations of the pyramid are a few thousand pixels per frame: as k_resample launches they are 3-4 nearly
// empty grids whose cost is launch + latency chain (C2: 48 us for 6 % of the pixels).  Here ONE workgroup per frame walks
// those generations in order: a thread produces 4 destination pixels straight from HBM/L2 (own tap evaluation, no LDS
// staging), generations are separated by a workgroup barrier.  Same arithmetic as k_resample's HBM-tap path.
#ifndef HT_TAIL_NT
#define HT_TAIL_NT 1024
#endif
#ifndef HT_TAIL_U
#define HT_TAIL_U 1  // groups per thread in flight: 1 = 68 VGPRs; 2-6 (96-240 VGPRs) measured no faster
#endif
constexpr int TAIL_NT = HT_TAIL_NT;
#ifndef HT_TAIL_SMALL_WPS
#define HT_TAIL_SMALL_WPS 8  // waves per SIMD the small-footprint tail leaves room for (8: <= 64 VGPRs)
#endif
// Taps come from tables the host built once per geometry (compact {a, (float)t} for the binary32 estimate, full binary64 form for the
// fallback) and the pixels take the same binary32-estimate / integer-box-mean / binary64-fallback route as k_resample: the tail
// used to spend ~135 binary64 instructions per group of 4 pixels on re-deriving taps and on the lerps.
// LDS_TAPS = false (round 5, option rs_tailtable=2): the compact taps stay in global memory (a few KB per geometry, L1 / L2 resident) and
// the register allocator leaves room for WPS waves per SIMD — the small-footprint form for batches that cover the chip, where the
// 35 KB / 68-VGPR form keeps the other batches' kernels off every CU (see the tail plan in ht_context.hip).
template <bool LDS_TAPS, int WPS>
__global__ __launch_bounds__(TAIL_NT, WPS) void k_resample_tail(const HtResampleJob *__restrict__ jobs, const uint32_t *__restrict__ prefix,
                                                          const HtTailTapRef *__restrict__ tapref, const HtTapFast *__restrict__ tfast,
                                                          const HtTap *__restrict__ tfull, const HtTailGens G, uint8_t *__restrict__ arena,
                                                          uint64_t arena_stride, uint32_t nframes) {
    __shared__ HtResampleJob s_jobs[HT_TAIL_MAX_JOBS];
    __shared__ HtTailTapRef s_ref[HT_TAIL_MAX_JOBS];
    __shared__ uint32_t s_pref[HT_TAIL_MAX_JOBS + 1];
    __shared__ HtTapFast s_taps[LDS_TAPS ? HT_TAIL_LDS_TAPS : 1];  // the generation's compact taps: one global round trip less per group
    uint32_t fidx, item;
    if (!xcd_item(1u, nframes, &fidx, &item)) return;  // same frame -> XCD placement as the generations before
    uint8_t *frame = arena + (uint64_t)fidx * arena_stride;
    const int tid = (int)threadIdx.x;
    for (int g = 0; g < G.ngen; g++) {
        const int jb = G.job_begin[g], nj = G.job_begin[g + 1] - jb;
        const uint32_t total = G.groups[g];
        {
            const uint32_t *src32 = reinterpret_cast<const uint32_t *>(jobs + jb);
            uint32_t *dst32 = reinterpret_cast<uint32_t *>(s_jobs);
            for (int i = tid; i < nj * (int)(sizeof(HtResampleJob) / 4); i += TAIL_NT) dst32[i] = src32[i];
            if (tid < nj) s_pref[tid] = prefix[jb + tid], s_ref[tid] = tapref[jb + tid];
            if (tid == 0) s_pref[nj] = total;
        }
        const uint32_t tap0 = G.tap_begin[g], ntap = G.tap_begin[g + 1] - tap0;
        const bool taps_in_lds = LDS_TAPS && ntap <= (uint32_t)HT_TAIL_LDS_TAPS;  // uniform
        if (taps_in_lds)
            for (uint32_t i = (uint32_t)tid; i < ntap; i += TAIL_NT) s_taps[i] = tfast[tap0 + i];
        __syncthreads();
        // TAIL_U groups per thread in flight: a group is a chain job lookup (LDS) -> taps (L2) -> source bytes (L2) -> store, and a
        // thread that walks its groups one at a time pays that chain once per group (the late generations are pure latency);
        // the phases below are separate loops over the TAIL_U groups so that every load of a phase is issued before the first use
        constexpr int TAIL_U = HT_TAIL_U;
        for (uint32_t i0 = (uint32_t)tid; i0 < total; i0 += TAIL_U * TAIL_NT) {
            bool live[TAIL_U], draw[TAIL_U];
            uint32_t doff[TAIL_U], mode[TAIL_U], tcol[TAIL_U], trow[TAIL_U];
            int npx[TAIL_U], sstride[TAIL_U], dwm1[TAIL_U];
            const uint8_t *sbase[TAIL_U];
#pragma unroll
            for (int u = 0; u < TAIL_U; u++) {  // phase 1: which job / row / columns
                const uint32_t i = i0 + u * TAIL_NT;
                live[u] = i < total;
                const uint32_t ii = live[u] ? i : 0u;
                int lo = 0, hi = nj;  // job of this group: last prefix <= i
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_pref[mid] <= ii) lo = mid;
                    else hi = mid;
                }
                const HtResampleJob &J = s_jobs[lo];
                const HtTailTapRef ref = s_ref[lo];
                const uint32_t q = ii - s_pref[lo], qpr = (uint32_t)(J.cw + 3) >> 2;
                const uint32_t y = q / qpr, x0 = (q - y * qpr) * 4u;
                npx[u] = min(4, J.dw - (int)x0);
                draw[u] = live[u] && (int)y < J.dh && npx[u] > 0;
                doff[u] = J.dst_off + y * (uint32_t)J.dst_stride + x0;
                mode[u] = ref.mode;
                tcol[u] = ref.col + (draw[u] ? x0 : 0u);
                trow[u] = ref.row + (draw[u] ? y : 0u);
                sstride[u] = J.src_stride;
                sbase[u] = frame + J.src_off;
                dwm1[u] = max(J.dw - 1 - (int)x0, 0);
            }
            HtTapFast ctap[TAIL_U][4], rtap[TAIL_U];
#pragma unroll
            for (int u = 0; u < TAIL_U; u++) {  // phase 2: taps (the column table is padded: 4 entries are always loadable)
                if (taps_in_lds) {
                    const HtTapFast *cf = s_taps + (tcol[u] - tap0);
                    ctap[u][0] = cf[0], ctap[u][1] = cf[1], ctap[u][2] = cf[2], ctap[u][3] = cf[3];
                    rtap[u] = s_taps[trow[u] - tap0];
                } else {
                    const HtTapFast *cf = tfast + tcol[u];
                    ctap[u][0] = cf[0], ctap[u][1] = cf[1], ctap[u][2] = cf[2], ctap[u][3] = cf[3];
                    rtap[u] = tfast[trow[u]];
                }
            }
            uint32_t p00[TAIL_U][4], p01[TAIL_U][4], p10[TAIL_U][4], p11[TAIL_U][4];
#pragma unroll
            for (int u = 0; u < TAIL_U; u++) {  // phase 3: source bytes.  b == a + 1 unless the coordinate was clamped to the rect's last
                // sample, and then its weight is exactly 0: the second tap is then read at a itself, so no load ever leaves the source
                // rect (the no-stale-L1 argument below needs every read to hit an already finished plane)
                const uint8_t *r0 = sbase[u] + (size_t)rtap[u].a * sstride[u], *r1 = r0 + (rtap[u].tf != 0.0f ? sstride[u] : 0);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int ca = ctap[u][k].a, cb = ca + (ctap[u][k].tf != 0.0f ? 1 : 0);
                    p00[u][k] = r0[ca], p01[u][k] = r0[cb], p10[u][k] = r1[ca], p11[u][k] = r1[cb];
                }
            }
#pragma unroll
            for (int u = 0; u < TAIL_U; u++) {  // phase 4: pixels + store
                uint32_t o = 0;
                if (draw[u]) {
                    auto exact64 = [&](int k) {  // the declared binary64 sequence with the full taps (b explicit: clamped at the rect's edge)
                        const HtTap cx = tfull[tcol[u] + min(k, dwm1[u])], ry = tfull[trow[u]];
                        const uint8_t *s0 = sbase[u] + (size_t)ry.a * sstride[u], *s1 = sbase[u] + (size_t)ry.b * sstride[u];
                        const double top = __dadd_rn(__dmul_rn((double)s0[cx.a], cx.u), __dmul_rn((double)s0[cx.b], cx.t));
                        const double bot = __dadd_rn(__dmul_rn((double)s1[cx.a], cx.u), __dmul_rn((double)s1[cx.b], cx.t));
                        const double vv = __dadd_rn(__dmul_rn(top, ry.u), __dmul_rn(bot, ry.t));
                        return (uint32_t)(int)__builtin_rint(vv);
                    };
                    uint32_t need = 0, obox = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t sum = p00[u][k] + p01[u][k] + p10[u][k] + p11[u][k], qq = sum >> 2, r4 = sum & 3u;
                        obox |= (qq + ((r4 + (qq & 1u)) > 2u ? 1u : 0u)) << (8 * k);  // exact 2:1: RNE(sum / 4)
                        const float a00 = (float)p00[u][k], a10 = (float)p10[u][k];
                        const float top = __builtin_fmaf(ctap[u][k].tf, (float)p01[u][k] - a00, a00);
                        const float bot = __builtin_fmaf(ctap[u][k].tf, (float)p11[u][k] - a10, a10);
                        const float v = __builtin_fmaf(rtap[u].tf, bot - top, top);
                        const float r = __builtin_rintf(v);
                        if (__builtin_fabsf(v - r) >= 0.5f - RS_EPS) need |= 1u << k;
                        o |= (uint32_t)(int)r << (8 * k);
                    }
                    const uint32_t keep = npx[u] >= 4 ? 0xffffffffu : ((1u << (8 * npx[u])) - 1u);
                    if (mode[u] & 1u) o = obox, need = 0;
                    if (mode[u] & 2u) need = 0xfu;  // option rs_nofast: binary64 everywhere
                    need &= (1u << npx[u]) - 1u;
                    o &= keep;
                    if (need) {
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (need & (1u << k)) o = (o & ~(0xffu << (8 * k))) | (exact64(k) << (8 * k));
                    }
                }
                if (live[u]) *reinterpret_cast<uint32_t *>(frame + doff[u]) = o;  // incl. the transparent border
            }
        }
        // the next generation reads what this workgroup just wrote: a workgroup-scope barrier is enough — the stores are
        // complete in L2 (write-through L1) before the barrier releases, and this CU cannot hold a stale L1 copy of a
        // destination line (nothing reads a plane before the generation that writes it; planes are 256-byte aligned).
        // Agent-scope fences here cost 4x the whole kernel: buffer_wbl2 writes the XCD's entire dirty L2 back.
        __syncthreads();
    }
}

// the round-1 tail kernel: taps re-derived per group in registers, binary64 lerps (kept for A/B: option rs_tailtable=0)
constexpr int TAILF_NT = 1024;
__global__ __launch_bounds__(TAILF_NT) void k_resample_tail_f64(const HtResampleJob *__restrict__ jobs, const uint32_t *__restrict__ prefix,
                                                          const HtTailGens G, uint8_t *__restrict__ arena, uint64_t arena_stride,
                                                          uint32_t nframes) {
    __shared__ HtResampleJob s_jobs[HT_TAIL_MAX_JOBS];
    __shared__ uint32_t s_pref[HT_TAIL_MAX_JOBS + 1];
    uint32_t fidx, item;
    if (!xcd_item(1u, nframes, &fidx, &item)) return;  // same frame -> XCD placement as the generations before
    uint8_t *frame = arena + (uint64_t)fidx * arena_stride;
    const int tid = (int)threadIdx.x;
    for (int g = 0; g < G.ngen; g++) {
        const int jb = G.job_begin[g], nj = G.job_begin[g + 1] - jb;
        const uint32_t total = G.groups[g];
        {
            const uint32_t *src32 = reinterpret_cast<const uint32_t *>(jobs + jb);
            uint32_t *dst32 = reinterpret_cast<uint32_t *>(s_jobs);
            for (int i = tid; i < nj * (int)(sizeof(HtResampleJob) / 4); i += TAILF_NT) dst32[i] = src32[i];
            if (tid < nj) s_pref[tid] = prefix[jb + tid];
            if (tid == 0) s_pref[nj] = total;
        }
        __syncthreads();
        for (uint32_t i = (uint32_t)tid; i < total; i += TAILF_NT) {
            int lo = 0, hi = nj;  // job of this group: last prefix <= i
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_pref[mid] <= i) lo = mid;
                else hi = mid;
            }
            const HtResampleJob &J = s_jobs[lo];
            const uint32_t q = i - s_pref[lo], qpr = (uint32_t)(J.cw + 3) >> 2;
            const uint32_t y = q / qpr, x0 = (q - y * qpr) * 4u;
            uint32_t o = 0;
            const int npx = min(4, J.dw - (int)x0);
            if ((int)y < J.dh && npx > 0) {
                const uint8_t *src = frame + J.src_off;
                RsTap cx[4];
#pragma unroll
                for (int k = 0; k < 4; k++) cx[k] = rs_tap(min((int)x0 + k, J.dw - 1), J.rx, J.sw, J.sx);
                const RsTap ry = rs_tap((int)y, J.ry, J.sh, J.sy);
                o = rs_pixels4<const uint8_t *>(src + (size_t)ry.a * J.src_stride, src + (size_t)ry.b * J.src_stride, cx, ry, 0, npx);
            }
            *reinterpret_cast<uint32_t *>(frame + J.dst_off + (size_t)y * J.dst_stride + x0) = o;  // incl. the transparent border
        }
        // the next generation reads what this workgroup just wrote: a workgroup-scope barrier is enough — the stores are
        // complete in L2 (write-through L1) before the barrier releases, and this CU cannot hold a stale L1 copy of a
        // destination line (nothing reads a plane before the generation that writes it; planes are 256-byte aligned).
        // Agent-scope fences here cost 4x the whole kernel: buffer_wbl2 writes the XCD's entire dirty L2 back.
        __syncthreads();
    }
}

// per-frame channel sums for getWhitebalance; out[f*4 + c] (u64), zeroed by the host
__global__ __launch_bounds__(256) void k_channel_sums(const uint8_t *__restrict__ frames, size_t frame_stride, uint32_t npix,
                                                      unsigned long long *__restrict__ out) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(frames + (size_t)blockIdx.y * frame_stride);
    uint32_t r = 0, g = 0, b = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const uint32_t v = p[i];
        r += v & 0xff;
        g += (v >> 8) & 0xff;
        b += (v >> 16) & 0xff;
    }
    unsigned long long rr = r, gg = g, bb = b;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        rr += __shfl_xor(rr, s, 64);
        gg += __shfl_xor(gg, s, 64);
        bb += __shfl_xor(bb, s, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[blockIdx.y * 4 + 0], rr);
        atomicAdd(&out[blockIdx.y * 4 + 1], gg);
        atomicAdd(&out[blockIdx.y * 4 + 2], bb);
    }
}

}  // namespace

ht_status ht_launch_pyramid(ht_ctx *c, uint32_t flags) {
    const bool gray_in_r = (flags & HT_INPUT_GRAY_IN_R) != 0;
    const HtDevLevel &L0 = c->h_levels[0];
    {
        HtProfScope ps(c, "gray");
        if ((c->W & 3) == 0) {
            const uint32_t ngroups = (uint32_t)((size_t)c->W * c->H / 4);
            const uint32_t bpf = std::min<uint32_t>((ngroups + 255) / 256, 2048);
            dim3 grid((bpf * (uint32_t)c->nframes + 7u) & ~7u);
            unsigned long long *wb = c->wb_fused ? reinterpret_cast<unsigned long long *>(c->d_scratch) : nullptr;
#define HT_GRAY_LAUNCH(G, W_)                                                                                                          \
    hipLaunchKernelGGL((k_gray_linear<G, W_>), grid, dim3(256), 0, c->stream, c->d_frames, c->frame_stride, c->d_arena, c->arena_stride, \
                       L0.off[0], ngroups, bpf, (uint32_t)c->nframes, wb)
            if (gray_in_r) {
                if (wb) HT_GRAY_LAUNCH(true, true);
                else HT_GRAY_LAUNCH(true, false);
            } else {
                if (wb) HT_GRAY_LAUNCH(false, true);
                else HT_GRAY_LAUNCH(false, false);
            }
#undef HT_GRAY_LAUNCH
        } else {
            dim3 grid((L0.stride / 4 + 63) / 64, (c->H + 3) / 4, c->nframes);
            if (gray_in_r)
                hipLaunchKernelGGL(k_gray_rows<true>, grid, dim3(64, 4), 0, c->stream, c->d_frames, c->frame_stride, c->d_arena,
                                   c->arena_stride, L0.off[0], c->W, c->H, L0.stride);
            else
                hipLaunchKernelGGL(k_gray_rows<false>, grid, dim3(64, 4), 0, c->stream, c->d_frames, c->frame_stride, c->d_arena,
                                   c->arena_stride, L0.off[0], c->W, c->H, L0.stride);
        }
        HT_HIP(c, hipGetLastError());
    }
    const size_t regular_end = c->tail_first_gen > 0 ? (size_t)c->tail_first_gen : c->h_gens.size();
    const int dbg_maxgen = c->rs_maxgen;  // 1 << 30 unless a -DHT_DEBUG_KNOBS build was told otherwise (results stale: what do the later generations cost the wall clock?)
    for (size_t g = 1; g < regular_end; g++) {
        if (c->gen_blocks[g] == 0 || (int)g > dbg_maxgen) continue;
        char gname[24];
        std::snprintf(gname, sizeof(gname), "resample_g%d", (int)g);
        HtProfScope ps(c, c->rs_gennames ? gname : "resample");  // option rs_gennames: device time per pyramid generation
        // frames per workgroup: as many as keep >= ~4 workgroups per CU slot in the launch, at most rs_group; groups never
        // straddle the 8 XCD shares of the batch when the batch is a multiple of 8 * K
        uint32_t K = 1;
        const uint32_t kmax = (uint32_t)((!c->rs_group_forced && (int64_t)c->W * c->H >= 400000) ? std::min(c->rs_group, 4) : c->rs_group);
        while (K * 2 <= kmax && (uint64_t)c->gen_blocks[g] * ((uint32_t)c->nframes / (K * 2)) >= (uint64_t)c->rs_min_wgs &&
               (uint32_t)c->nframes % (K * 2 * 8) == 0)
            K *= 2;
        if (c->dbg_rs_k > 0) K = (uint32_t)std::min(c->dbg_rs_k, std::max(1, c->nframes));  // option rs_k: measurement knob
        const uint32_t ngroups = ((uint32_t)c->nframes + K - 1) / K;
        const dim3 rgrid((c->gen_blocks[g] * ngroups + 7u) & ~7u);
        if (c->rs_bands)
            hipLaunchKernelGGL(k_resample_bands<HT_RS_MAX_PASSES>, rgrid, dim3(256), 0, c->stream, c->d_gen_blocks[g], c->d_arena, c->arena_stride,
                               c->gen_blocks[g], ngroups, (uint32_t)c->nframes, K);
        else
            hipLaunchKernelGGL(k_resample<HT_RS_MAX_PASSES>, rgrid, dim3(256), 0, c->stream, c->d_gen_blocks[g], c->d_arena, c->arena_stride,
                               c->gen_blocks[g], ngroups, (uint32_t)c->nframes, K);
        HT_HIP(c, hipGetLastError());
        if ((int)g == c->early_gen && c->early_gen > 0) {  // the early scales' planes are complete: their scan starts on the second stream
            ht_status st = ht_launch_scan_early(c, flags);
            if (st != HT_OK) return st;
        }
    }
    if (c->tail_first_gen > 0 && c->tail_first_gen <= dbg_maxgen) {
        HtProfScope ps(c, c->rs_gennames ? "resample_tail" : "resample");
        if (c->tail_table == 2)
            hipLaunchKernelGGL((k_resample_tail<false, HT_TAIL_SMALL_WPS>), dim3(((uint32_t)c->nframes + 7u) & ~7u), dim3(TAIL_NT), 0, c->stream, c->d_tail_jobs,
                               c->d_tail_prefix, c->d_tail_tapref, c->d_tail_taps_fast, c->d_tail_taps, c->h_tail, c->d_arena, c->arena_stride, (uint32_t)c->nframes);
        else if (c->tail_table)
            hipLaunchKernelGGL((k_resample_tail<true, 4>), dim3(((uint32_t)c->nframes + 7u) & ~7u), dim3(TAIL_NT), 0, c->stream, c->d_tail_jobs, c->d_tail_prefix,
                               c->d_tail_tapref, c->d_tail_taps_fast, c->d_tail_taps, c->h_tail, c->d_arena, c->arena_stride, (uint32_t)c->nframes);
        else
            hipLaunchKernelGGL(k_resample_tail_f64, dim3(((uint32_t)c->nframes + 7u) & ~7u), dim3(TAILF_NT), 0, c->stream, c->d_tail_jobs, c->d_tail_prefix,
                               c->h_tail, c->d_arena, c->arena_stride, (uint32_t)c->nframes);
        HT_HIP(c, hipGetLastError());
    }
    return HT_OK;
}

ht_status ht_launch_gray_inplace(ht_ctx *c, uint8_t *d_rgba, int n, size_t stride) {
    HtProfScope ps(c, "gray_inplace");
    const uint32_t npix = (uint32_t)((size_t)c->W * c->H);
    hipLaunchKernelGGL(k_gray_inplace, dim3(std::min<uint32_t>((npix + 255) / 256, 2048), n), dim3(256), 0, c->stream, d_rgba, stride, npix);
    HT_HIP(c, hipGetLastError());
    return HT_OK;
}

ht_status ht_launch_whitebalance(ht_ctx *c, double *d_out, bool zero) {
    HtProfScope ps(c, "whitebalance");
    unsigned long long *out = reinterpret_cast<unsigned long long *>(d_out);
    if (zero) HT_HIP(c, hipMemsetAsync(out, 0, sizeof(unsigned long long) * 4 * (size_t)c->nframes, c->stream));
    const uint32_t npix = (uint32_t)((size_t)c->W * c->H);
    hipLaunchKernelGGL(k_channel_sums, dim3(std::min<uint32_t>((npix + 1023) / 1024, 256), c->nframes), dim3(256), 0, c->stream,
                       c->d_frames, c->frame_stride, npix, out);
    HT_HIP(c, hipGetLastError());
    return HT_OK;
}

#ifdef HT_RS_PHASES
// k_resample_bands: out8 = g_rsb_ph (see there)
extern "C" int ht_debug_rsb_phases(unsigned long long *out8, int reset) {
    static unsigned long long h[256][8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rsb_ph), sizeof(h)) != hipSuccess) return 1;
    for (int i = 0; i < 8; i++) out8[i] = 0;
    for (int sh = 0; sh < 256; sh++)
        for (int i = 0; i < 8; i++) out8[i] += h[sh][i];
    if (reset) {
        std::memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_rsb_ph), h, sizeof(h)) != hipSuccess) return 1;
    }
    return 0;
}
// sums per phase over all recorded frame iterations: out16[i] = cycles between stamp i-1 and stamp i (i = 3..6; 2 = from the previous
// iteration's stamp 6), out16[8 + i] = samples
extern "C" int ht_debug_rs_phases(unsigned long long *out16, int reset) {
    static unsigned long long h[16384][8][8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rs_tl), sizeof(h)) != hipSuccess) return 1;
    for (int j = 0; j < 16; j++) out16[j] = 0;
    for (int sl = 0; sl < 16384; sl++)
        for (int it = 0; it < 8; it++) {
            const unsigned long long *t = h[sl][it];
            // empty or torn slot: different launches that hash to the same slot overwrite each other's stamps, and a mixture of two workgroups'
            // stamps used to enter the means as a phase of 10^5 .. 10^7 cycles (round 3: "stores + barrier 17 534 cycles" was such a mean)
            if (!t[2] || !t[6] || t[6] < t[2] || t[6] - t[2] > (1ull << 18) || t[3] < t[2] || t[4] < t[3] || t[5] < t[4] || t[6] < t[5]) continue;
            for (int i = 3; i <= 6; i++) out16[i] += t[i] - t[i - 1], out16[8 + i]++;
            // setup of the workgroup: only if stamps 0, 1 and the first loop top are in order and close together (different launches
            // that hashed to the same slot overwrite each other's stamps)
            if (it == 0 && h[sl][0][0] && h[sl][0][0] <= h[sl][0][1] && h[sl][0][1] <= t[2] && t[2] - h[sl][0][0] < (1ull << 16)) {
                out16[1] += h[sl][0][1] - h[sl][0][0], out16[8 + 1]++;
                out16[2] += t[2] - h[sl][0][1], out16[8 + 2]++;
            }
        }
    if (const char *e = std::getenv("HT_RS_SUBSTAMPS")) {  // setup sub-phases: entry -> record/extent inputs -> extents -> addresses -> loads issued -> taps -> barrier
        (void)e;
        unsigned long long sum[6] = {}, n = 0;
        for (int sl = 0; sl < 16384; sl++) {
            const unsigned long long *u = h[sl][7], t0 = h[sl][0][0], t1 = h[sl][0][1];
            if (!t0 || !t1 || !u[0] || !u[4] || t1 < t0 || t1 - t0 > (1ull << 24) || u[0] < t0 || u[4] > t1) continue;
            sum[0] += u[0] - t0, sum[1] += u[1] - u[0], sum[2] += u[2] - u[1], sum[3] += u[3] - u[2], sum[4] += u[4] - u[3], sum[5] += t1 - u[4], n++;
        }
        if (n) std::printf("  setup sub-phases (%llu samples): record %.0f, extents %.0f, addresses %.0f, load issue %.0f, tap tables %.0f, barrier %.0f cycles\n", n,
                           (double)sum[0] / n, (double)sum[1] / n, (double)sum[2] / n, (double)sum[3] / n, (double)sum[4] / n, (double)sum[5] / n);
    }
    if (std::getenv("HT_RS_LAUNCHES")) {  // per launch: how long does a workgroup live?
        static unsigned long long L[1024][4];
        if (hipMemcpyFromSymbol(L, HIP_SYMBOL(g_rs_launch), sizeof(L)) != hipSuccess) return 1;
        for (int k = 0; k < 1024; k++)
            if (L[k][0]) std::printf("  launches with %4d tiles per frame: %6llu workgroups (LDS path), mean life %7.0f cycles\n", k, L[k][0], (double)L[k][1] / (double)L[k][0]);
    }
    if (std::getenv("HT_RS_WAVESTAMPS")) {  // per-wave view of one frame iteration: do the four wavefronts of a workgroup reach the barrier together?
        static unsigned long long w[4096][4][6];
        if (hipMemcpyFromSymbol(w, HIP_SYMBOL(g_rs_tw), sizeof(w)) != hipSuccess) return 1;
        double pix[4] = {}, wait1[4] = {}, mid[4] = {}, spread = 0, pixmax = 0, pixmin = 0;
        unsigned long long n = 0, simd_hist[4][4] = {};
        for (int sl = 0; sl < 4096; sl++) {
            bool ok = true;
            unsigned long long bmin = ~0ull, bmax = 0, pmn = ~0ull, pmx = 0;
            for (int k = 0; k < 4; k++) {
                const unsigned long long *t = w[sl][k];
                if (!t[0] || t[1] < t[0] || t[2] < t[1] || t[3] < t[2] || t[3] - t[0] > (1ull << 22)) ok = false;
                bmin = std::min(bmin, t[1]), bmax = std::max(bmax, t[1]);
                pmn = std::min(pmn, t[1] - t[0]), pmx = std::max(pmx, t[1] - t[0]);
            }
            if (!ok) continue;
            n++;
            spread += (double)(bmax - bmin), pixmax += (double)pmx, pixmin += (double)pmn;
            for (int k = 0; k < 4; k++) {
                const unsigned long long *t = w[sl][k];
                pix[k] += (double)(t[1] - t[0]), wait1[k] += (double)(t[2] - t[1]), mid[k] += (double)(t[3] - t[2]);
                simd_hist[k][(t[4] >> 4) & 3]++;
            }
        }
        if (n) {
            std::printf("  per-wave view (%llu workgroups, third frame iteration): pixel phase / wait at barrier 1 / tile write + stores + barrier 2, cycles\n", n);
            for (int k = 0; k < 4; k++)
                std::printf("    wave %d: %7.0f %7.0f %7.0f   SIMD histogram %llu %llu %llu %llu\n", k, pix[k] / n, wait1[k] / n, mid[k] / n, simd_hist[k][0], simd_hist[k][1],
                            simd_hist[k][2], simd_hist[k][3]);
            std::printf("    pixel phase of the fastest / slowest wave of a workgroup: %.0f / %.0f; spread of their arrival at barrier 1: %.0f cycles\n", pixmin / n, pixmax / n, spread / n);
        }
    }
    if (reset) {
        std::memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_rs_tl), h, sizeof(h)) != hipSuccess) return 1;
        static unsigned long long wz[4096][4][6];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_rs_tw), wz, sizeof(wz)) != hipSuccess) return 1;
        static unsigned long long lz[1024][4];
        for (int k = 0; k < 1024; k++) lz[k][0] = lz[k][1] = lz[k][3] = 0, lz[k][2] = ~0ull;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_rs_launch), lz, sizeof(lz)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
