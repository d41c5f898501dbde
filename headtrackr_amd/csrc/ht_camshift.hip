// ht_camshift.hip — camshift.Tracker on the device (reference: /root/reference/src/camshift.js).
//
//   Histogram            camshift.js:49-72     bins[256*(R>>4) + 16*(G>>4) + (B>>4)]++ over RGBA pixels
//   initTracker          camshift.js:198-211   model histogram of the tracked rect (outside the canvas = transparent
//                                              black = bin 0), search window = rect
//   track -> camShift    camshift.js:213-259   size / angle from second moments, new window = 1.1 x object size
//   meanShift            camshift.js:261-312   full-frame histogram, weights, <= 10 window iterations
//   getWeights           camshift.js:314-330   w = ch ? min(mh/ch, 1) : 0
//   Moments              camshift.js:79-120
//
// Device mapping: one tracker ("stream") per frame of the bound batch; state (model histogram, search window, track
// object) stays in HBM between calls.  track = (1) k_cs_hist: LDS-privatised 4096-bin histogram per frame chunk,
// written as partial histograms (<= 8 per stream, no global atomics); (2) k_cs_meanshift: ONE workgroup per stream keeps the 4096-entry weight LUT in LDS
// (binary64, 32 KB) and runs the whole <=10-iteration mean-shift loop: each iteration is a window moment reduction
// straight from the RGBA pixels through the LUT — the back-projection image of the reference (camshift.js:332-353) is
// never materialised, it is only observable through debug getters.  Moment sums are binary64 with a fixed
// thread-to-pixel assignment and a fixed reduction tree (deterministic); the summation ORDER differs from the
// reference's column-major scalar loop, so results are equal up to rounding in the last bits of xc/yc — the
// acceptance bound of BASELINE.json (+-1 px, +-0.5 deg) covers the rare truncation flips this can cause.
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: librccl.so is opened lazily (dlopen) the first time ht_allgather_records runs

#include "ht_internal.h"

namespace {

#ifdef HT_CS_TIMELINE  // measurement build (tools/gpu_cs_timeline.py): shader-clock stamps of a workgroup's phases
#define CS_STAMP(arr, i)                                                      \
    do {                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                    \
        if ((arr) && threadIdx.x == 0 && (i) < 30) (arr)[(i)] = __builtin_readcyclecounter(); \
        __builtin_amdgcn_sched_barrier(0);                                    \
    } while (0)
#else
#define CS_STAMP(arr, i)
#endif

constexpr int CS_NT = 512;          // threads of the mean-shift workgroup
#ifndef HT_HIST_NT
#define HT_HIST_NT 1024
#endif
constexpr int HIST_NT = HT_HIST_NT;
// Partial histograms per stream: enough chunks to put ~256 workgroups of 1024 threads on the chip (a single 1080p stream gets 127,
// eight of them 32 each, a batch of >= 32 streams 8 each), each chunk >= 16384 pixels and a multiple of 4 * HIST_NT.  Round 5, same box,
// the C5 track step of 8 / 1 feeds (tools/gpu_cs_step.py): 256 threads x ~1024 workgroups 39.1 / 27.8 us, 512 x 512: 36.5 / 24.6,
// 1024 x 512: 36.4 / 24.2, 1024 x 256: 35.8 / 24.0 (a quarter of the chunk histograms to write and to sum), 8 loads in flight per
// thread instead of 4: 39.4 / 26.8.
#ifndef HT_HIST_MAXCHUNKS
#define HT_HIST_MAXCHUNKS 128
#endif
#ifndef HT_HIST_UNROLL
#define HT_HIST_UNROLL 4
#endif
#ifndef HT_HIST_TARGET_WGS
#define HT_HIST_TARGET_WGS 256  // workgroups of a k_cs_hist launch, all streams together
#endif
inline uint32_t hist_max_chunks(int nstreams) { return (uint32_t)std::min(HT_HIST_MAXCHUNKS, std::max(8, HT_HIST_TARGET_WGS / std::max(nstreams, 1))); }
inline void hist_chunks(uint32_t npix, uint32_t max_chunks, uint32_t *chunk_px, uint32_t *nchunks) {
    uint32_t n = std::min<uint32_t>((npix + 16383u) / 16384u, max_chunks);
    n = std::max<uint32_t>(n, 1u);
    const uint32_t q = 4u * HIST_NT;
    *chunk_px = std::max<uint32_t>(((npix + n - 1) / n + q - 1) / q * q, q);
    *nchunks = std::max<uint32_t>((npix + *chunk_px - 1) / *chunk_px, 1u);
}

// camshift.js:63-66 (px = R | G<<8 | B<<16 | A<<24): bin = (R>>4)<<8 | (G>>4)<<4 | (B>>4).  Four instructions instead of the eight of
// the field-by-field form (the histogram pass of k_cs_track_fused spent 32 of its ~50 vector instructions per 16-byte load on its four
// bins): with t = px & 0xf0f0f0 = r<<4 | g<<12 | b<<20 (r, g, b the 4-bit fields), t + (t << 12) puts g at bit 24 next to b at bit 20
// (all fields of the sum are disjoint: no carries; b << 32 leaves the register), t << 24 puts r at bit 28, and the bin is bits 20-31.
__device__ __forceinline__ uint32_t cs_bin(uint32_t px) {
#ifdef HT_CS_BIN_FIELDS  // A/B (tools/build_alt.py): the field-by-field form
    return ((px & 0xf0u) << 4) | ((px >> 8) & 0xf0u) | ((px >> 20) & 0xfu);
#else
    const uint32_t t = px & 0x00f0f0f0u;
    return ((t << 24) | (t + (t << 12))) >> 20;  // v_and, v_mul_u32_u24 0x1001, v_lshl_or_b32, v_lshrrev
#endif
}

// A batch of PREDICATED loads (`v = ok ? p[i] : 0`) followed by cs_bin: the optimiser folds the bin's first instruction (`& 0xf0f0f0`, which maps
// the 0 of the not-taken side to 0) into the load's own block, where it has to wait for the load on the spot — every load of the batch
// then costs its own round trip (k_cs_hist 16.4 -> 20 us at 8 x 1080p, seen in the code object: `s_waitcnt vmcnt(0)` behind every load).
// Laundering the loaded registers AFTER the whole batch keeps the consumers behind all of its loads.
#define CS_BATCH_LOADED(v_) asm volatile("" : "+v"(v_))

__device__ __forceinline__ int32_t toint32(double v) {  // ECMAScript ToInt32 (>>0, <<2)
    if (!(fabs(v) < 1.0e300)) return 0;                  // NaN, +-Infinity
    const double t = trunc(v);
    if (fabs(t) < 2147483648.0) return (int32_t)t;
    double m = fmod(t, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int32_t)(uint32_t)m;
}

// forward declaration: wave-merged LDS histogram update (defined with k_cs_hist below)
__device__ __forceinline__ void hist_add_wave(uint32_t *h, uint32_t bin, uint32_t count, bool active);

// initTracker: one 1024-thread workgroup per stream; rows of the rect by wavefront, columns by lane (no per-pixel division),
// 8 independent loads per lane in flight (a 360 x 360 rect of a 1080p feed took 174 us with the one-pixel-at-a-time loop)
constexpr int INIT_NT = 1024;
__global__ __launch_bounds__(INIT_NT) void k_cs_init(const uint8_t *__restrict__ frames, size_t frame_stride, int W, int H,
                                                     const ht_cs_rect *__restrict__ rects, HtCsState *__restrict__ states, int first) {
    __shared__ uint32_t h[4096];
    const int s = blockIdx.x;
    for (int i = threadIdx.x; i < 4096; i += INIT_NT) h[i] = 0;
    __syncthreads();
    const ht_cs_rect r = rects[s];
    const uint32_t *img = reinterpret_cast<const uint32_t *>(frames + (size_t)s * frame_stride);
    const int rw = max(r.width, 0), rh = max(r.height, 0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int NWV = INIT_NT / 64;
    for (int j0 = wave; j0 - wave < rh; j0 += 8 * NWV) {      // same trip count for every wavefront's lanes (ballots inside)
        for (int cb = 0; cb < rw; cb += 64) {
            const int c = cb + lane;
            uint32_t px[8];
            bool in[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int y = r.y + j0 + u * NWV, x = r.x + c;
                in[u] = c < rw && j0 + u * NWV < rh;                                   // inside the rect
                const bool img_ok = in[u] && x >= 0 && x < W && y >= 0 && y < H;      // inside the canvas
                px[u] = img_ok ? img[(size_t)y * W + x] : 0u;  // getImageData outside the canvas: transparent black -> bin 0 (camshift.js:206)
            }
#pragma unroll
            for (int u = 0; u < 8; u++) CS_BATCH_LOADED(px[u]);
#pragma unroll
            for (int u = 0; u < 8; u++) hist_add_wave(h, cs_bin(px[u]), 1u, in[u]);
        }
    }
    __syncthreads();
    HtCsState &st = states[first + s];
    for (int i = threadIdx.x; i < 4096; i += INIT_NT) st.model[i] = h[i];
    if (threadIdx.x == 0) {
        st.sw[0] = r.x, st.sw[1] = r.y, st.sw[2] = r.width, st.sw[3] = r.height;  // camshift.js:209
        st.x = st.y = st.width = st.height = st.angle = 0.0;                         // camshift.js:210
        st.win_px = st.calls = 0;
    }
}

// initTracker for a FEW streams with large rects (a live 1080p feed: 360 x 360 = 0.5 MB took the single workgroup above 54 us):
// grid (G, streams), workgroup g takes rows g*4 + wavefront, + 4 G, ...; LDS histogram per workgroup, non-zero bins added to the
// model (zeroed by the host) with global atomics — integer counts, any order gives the same model.
__global__ __launch_bounds__(256) void k_cs_init_rows(const uint8_t *__restrict__ frames, size_t frame_stride, int W, int H,
                                                      const ht_cs_rect *__restrict__ rects, HtCsState *__restrict__ states, int first) {
    __shared__ uint32_t h[4096];
    const int s = blockIdx.y, g = blockIdx.x, G = gridDim.x;
    for (int i = threadIdx.x; i < 4096; i += 256) h[i] = 0;
    __syncthreads();
    const ht_cs_rect r = rects[s];
    const uint32_t *img = reinterpret_cast<const uint32_t *>(frames + (size_t)s * frame_stride);
    const int rw = max(r.width, 0), rh = max(r.height, 0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = g * 4 + wave; j - wave < rh + 3; j += 4 * G) {  // same trip count for the four wavefronts of a workgroup
        const int y = r.y + j;
        for (int cb = 0; cb < rw; cb += 256) {
            uint32_t px[4];
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int c = cb + 64 * u + lane, x = r.x + c;
                in[u] = c < rw && j < rh;
                px[u] = (in[u] && x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0u;  // outside the canvas: transparent black (camshift.js:206)
            }
#pragma unroll
            for (int u = 0; u < 4; u++) CS_BATCH_LOADED(px[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) hist_add_wave(h, cs_bin(px[u]), 1u, in[u]);
        }
    }
    __syncthreads();
    HtCsState &st = states[first + s];
    for (int i = threadIdx.x; i < 4096; i += 256)
        if (h[i]) atomicAdd(&st.model[i], h[i]);
    if (g == 0 && threadIdx.x == 0) {
        st.sw[0] = r.x, st.sw[1] = r.y, st.sw[2] = r.width, st.sw[3] = r.height;  // camshift.js:209
        st.x = st.y = st.width = st.height = st.angle = 0.0;                         // camshift.js:210
        st.win_px = st.calls = 0;
    }
}

// full-frame histogram (camshift.js:268): grid (chunks, streams) -> hist[stream][chunk][4096] partial histograms.
// 4 pixels per 16-byte load.  LDS atomics on one address serialise lane by lane, and flat image regions put whole
// wavefronts into one bin (a flat 320x240 background cost 64 cycles per wave instruction: the kernel ran at a quarter of
// HBM speed), so counts are merged before they reach LDS: the 4 pixels of a thread when they share a bin, and all lanes
// that share the first active lane's bin through one ballot — one atomic for the whole wavefront on flat regions,
// a few extra scalar instructions elsewhere.  Counts are integers: any order gives the same histogram.
__device__ __forceinline__ void hist_add_wave(uint32_t *h, uint32_t bin, uint32_t count, bool active) {
    const unsigned long long act = __ballot(active);
    if (!act) return;
    const uint32_t lead_lane = (uint32_t)__builtin_ctzll(act);
    const uint32_t lead_bin = (uint32_t)__builtin_amdgcn_readlane((int)bin, (int)lead_lane);
    const uint32_t lead_cnt = (uint32_t)__builtin_amdgcn_readlane((int)count, (int)lead_lane);
    const bool same = active && bin == lead_bin && count == lead_cnt;
    const unsigned long long m = __ballot(same);
    if ((threadIdx.x & 63u) == lead_lane) atomicAdd(&h[lead_bin], lead_cnt * (uint32_t)__popcll(m));
    if (active && !same) atomicAdd(&h[bin], count);
}

__global__ __launch_bounds__(HIST_NT) void k_cs_hist(const uint8_t *__restrict__ frames, size_t frame_stride, uint32_t npix,
                                                     uint32_t chunk_px, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[4096];
    for (int i = threadIdx.x; i < 4096; i += HIST_NT) h[i] = 0;
    __syncthreads();
    const uint8_t *frame = frames + (size_t)blockIdx.y * frame_stride;
    const uint32_t beg = blockIdx.x * chunk_px, end = min(beg + chunk_px, npix);  // chunk_px is a multiple of 4 * HIST_NT
    const uint32_t nquad = (end - beg) / 4;
    const uint4 *img4 = reinterpret_cast<const uint4 *>(frame + (size_t)beg * 4);
    const uint32_t iters = chunk_px / (4 * HIST_NT);
    // HT_HIST_UNROLL loads of a thread in flight before the first bin is counted, written out: the wave-level merge below is convergent code,
    // which keeps the optimiser from unrolling the loop itself (`#pragma unroll` was refused), and with ONE 16-byte load in flight per
    // thread the pass was a chain of chunk_px / 1024 memory round trips (16 x ~1.3 us at 1080p = the kernel's whole duration)
    for (uint32_t it0 = 0; it0 < iters; it0 += HT_HIST_UNROLL) {
        uint4 pv[HT_HIST_UNROLL];
        bool onv[HT_HIST_UNROLL];
#pragma unroll
        for (int u = 0; u < HT_HIST_UNROLL; u++) {
            const uint32_t i = (it0 + (uint32_t)u) * HIST_NT + threadIdx.x;
            onv[u] = it0 + (uint32_t)u < iters && i < nquad;
            pv[u] = make_uint4(0u, 0u, 0u, 0u);
            if (onv[u]) pv[u] = img4[i];
        }
#pragma unroll
        for (int u = 0; u < HT_HIST_UNROLL; u++) {
            CS_BATCH_LOADED(pv[u].x);
            CS_BATCH_LOADED(pv[u].y);
            CS_BATCH_LOADED(pv[u].z);
            CS_BATCH_LOADED(pv[u].w);
        }
#pragma unroll
        for (int u = 0; u < HT_HIST_UNROLL; u++) {
            if (it0 + (uint32_t)u >= iters) break;  // workgroup-uniform
            const uint4 p = pv[u];
            const bool on = onv[u];
            const uint32_t b0 = cs_bin(p.x), b1 = cs_bin(p.y), b2 = cs_bin(p.z), b3 = cs_bin(p.w);
            const bool flat = (b0 == b1) && (b2 == b3) && (b0 == b2);
            hist_add_wave(h, b0, flat ? 4u : 1u, on);
            if (on && !flat) {
                atomicAdd(&h[b1], 1u);
                atomicAdd(&h[b2], 1u);
                atomicAdd(&h[b3], 1u);
            }
        }
    }
    const uint32_t *img = reinterpret_cast<const uint32_t *>(frame);
    for (uint32_t i = beg + nquad * 4 + threadIdx.x; i < end; i += HIST_NT) atomicAdd(&h[cs_bin(img[i])], 1u);  // < 4 pixels
    __syncthreads();
    // this chunk's partial histogram, written whole (no zeroing pass, no global atomics); k_cs_meanshift adds the chunks
    uint32_t *out = hist + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4096;
    for (int i = threadIdx.x; i < 4096; i += HIST_NT) out[i] = h[i];
}

struct Mom {
    double m00, m10, m01, m11, m20, m02;
};

// binary64 wave sum with DPP row shifts / row broadcasts: a fixed tree (deterministic), VALU only.  The __shfl_xor tree it
// replaces is 6 dependent ds_bpermute round trips per value and half — 36 to 72 LDS round trips per moment pass, which was most
// of a pass's latency on the small windows of C3.  Lanes a shift does not reach read 0 (bound_ctrl) and add +0.0.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    double s = v + dpp_f64<0x111, 0xf, 0xf>(v);  // row_shr:1
    s += dpp_f64<0x112, 0xf, 0xf>(v);            // row_shr:2
    s += dpp_f64<0x113, 0xf, 0xf>(v);            // row_shr:3
    s += dpp_f64<0x114, 0xf, 0xe>(s);            // row_shr:4, banks 1-3
    s += dpp_f64<0x118, 0xf, 0xc>(s);            // row_shr:8, banks 2-3
    s += dpp_f64<0x142, 0xa, 0xf>(s);            // row_bcast:15 into rows 1, 3
    s += dpp_f64<0x143, 0xc, 0xf>(s);            // row_bcast:31 into rows 2, 3
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 63), __builtin_amdgcn_readlane(__double2loint(s), 63));
}

// camshift.Moments (camshift.js:79-120) over columns [x, w) x rows [y, h) — w, h are the right / bottom EDGES.
// Mapping: lanes along a row (coalesced RGBA reads), wavefront wv takes rows wv, wv + NW, ...; a row's y factor leaves the pixel
// loop:  rs = sum val, ts = sum vx*val (us = sum vx^2*val) per row, then m00 += rs, m10 += ts, m01 += vy*rs (m11 += vy*ts,
// m20 += us, m02 += vy^2*rs) — 2 adds + 1 multiply per pixel for the first moments and no integer division.  Same real-number
// sums as the reference's column-major loop, different rounding order (see header).
//
// The pixels of a pass come either from global memory or — REG — from a copy of the neighbourhood of the search window that the
// workgroup made in LDS (12-bit histogram bin per pixel, CsRegion): measured on C3, a pass over a 90 x 90 window cost 7-10 us
// when its loads went out to memory (every CU streams its frame for the histogram at the same time, 4 frame versions do not
// fit the 256 MB Infinity Cache), although it is two rounds of independent loads; from LDS it is a fraction of a microsecond.
struct CsRegion {
    const uint16_t *bins;  // [rh][rw] bins, LDS
    int x0, y0, rw, rh;    // rw == 0: no region cached
};

// NW = wavefronts the rows are dealt out to, PW = wavefronts the workgroup really has (NW % PW == 0): with PW < NW a wavefront plays
// NW / PW of them in turn (rows, partial sums, wave sums and their slots of red[] are those of the NW-wavefront workgroup), so a
// 512-thread workgroup adds exactly what a 1024-thread one adds, in the same order — k_cs_track_fused<.., 512> below.
template <bool SECOND, int NW, bool REG, int PW = NW>
__device__ __forceinline__ Mom window_moments(const uint32_t *__restrict__ img, int W, const double *lut, const CsRegion &R, int x, int y, int w, int h,
                                              double (*red)[NW], unsigned long long *fine = nullptr) {
    Mom m = {0, 0, 0, 0, 0, 0};
    CS_STAMP(fine, 0);
    const int ww = w - x, hh = h - y;
    const int lane = threadIdx.x & 63;
    constexpr int VPP = NW / PW;
    static_assert(NW % PW == 0, "virtual wavefronts per physical one");
    double vs[VPP > 1 ? VPP - 1 : 1][6];  // wave sums of the wavefronts already played (wave-uniform)
#pragma unroll
    for (int vi = 0; vi < VPP; vi++) {
    const int wave = (int)(threadIdx.x >> 6) + vi * PW;
    m = Mom{0, 0, 0, 0, 0, 0};
    // Shader-clock stamps (HT_CS_TIMELINE) of a pass over an 83 x 83 window: pixel loop 9.3 k cycles, wave sums 0.8 k, final sums
    // 1.4 k, scalar mean-shift logic 0.45 k — the pixel loop was a chain of dependent round trips (bin -> LUT -> add), made
    // sequential by per-row early exits that kept the compiler from batching the loads.  So: every load of a batch of 8 rows is
    // unconditional (clamped address, value masked) and issued before the first use, and only as many wavefronts take part as
    // there are 8-row batches (at least 4 = one per SIMD); the others wait at the barriers with zero partial sums.
    const int nwa = min(NW, max(4, (hh + 7) >> 3));
    if (ww > 0 && hh > 0 && wave < nwa) {
        for (int j0 = wave; j0 < hh; j0 += 8 * nwa) {
            // per row: rs = sum val, ts = sum vx val (both are needed again times the row's y); the x^2 sum has no y factor and
            // goes straight into m20 (one accumulator instead of eight: the kernel sits at the 128-VGPR cap of a 1024-thread workgroup)
            double rs[8], ts[8];
#pragma unroll
            for (int r = 0; r < 8; r++) rs[r] = ts[r] = 0.0;
            for (int cb = 0; cb < ww; cb += 64) {  // cb: wave-uniform chunk base
                const int c = cb + lane, cc = min(c, ww - 1);
                uint32_t px[8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int j = min(j0 + r * nwa, hh - 1);  // clamped address, value masked below
                    if (REG) px[r] = R.bins[(y + j - R.y0) * R.rw + (x + cc - R.x0)];
                    else px[r] = img[(size_t)(y + j) * W + (x + cc)];
                }
                if (!REG) {
                    // the eight loads are issued before the first bin is computed.  Left to the scheduler, one build of the 512-thread
                    // kernel (128-VGPR cap) fetched them one at a time into ONE register — eight dependent round trips per 64 columns on
                    // the path that streams with windows beyond the LDS region take: +9 % on the whole launch
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 8; r++) px[r] = cs_bin(px[r]);
                }
                double val[8];
#pragma unroll
                for (int r = 0; r < 8; r++) val[r] = lut[px[r]];
                const double vx = (double)c;
                const bool colok = c < ww;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const double vv = (colok && j0 + r * nwa < hh) ? val[r] : 0.0;
                    rs[r] += vv;
                    ts[r] += vx * vv;
                    if (SECOND) m.m20 += vx * vx * vv;
                }
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const double vy = (double)(j0 + r * nwa);  // rows past the window have rs = ts = us = 0
                m.m00 += rs[r];
                m.m10 += ts[r];
                m.m01 += vy * rs[r];
                if (SECOND) {
                    m.m11 += vy * ts[r];
                    m.m02 += vy * vy * rs[r];
                }
            }
        }
    }
    if (vi + 1 < VPP) {  // not the last wavefront this one plays: its wave sums wait (scalar registers) for the barrier below
        const double vq[6] = {m.m00, m.m10, m.m01, m.m11, m.m20, m.m02};
#pragma unroll
        for (int k = 0; k < (SECOND ? 6 : 3); k++) vs[vi][k] = wave < nwa ? wave_sum_f64(vq[k]) : 0.0;
    }
    }  // vi
    const int wave = (int)(threadIdx.x >> 6) + (VPP - 1) * PW;
    const int nwa = min(NW, max(4, (hh + 7) >> 3));
    double v[6] = {m.m00, m.m10, m.m01, m.m11, m.m20, m.m02};
    constexpr int nv = SECOND ? 6 : 3;
    CS_STAMP(fine, 1);
    __syncthreads();  // red[] may still be read from the previous call
    CS_STAMP(fine, 2);
#pragma unroll
    for (int k = 0; k < nv; k++) {
        const double s = wave < nwa ? wave_sum_f64(v[k]) : 0.0;  // wave-uniform branch; idle waves contribute an exact 0
        if (lane == 0) red[k][wave] = s;
#pragma unroll
        for (int vi = 0; vi + 1 < VPP; vi++)
            if (lane == 0) red[k][(int)(threadIdx.x >> 6) + vi * PW] = vs[vi][k];
    }
    CS_STAMP(fine, 3);
    __syncthreads();
    CS_STAMP(fine, 4);
    // the workgroup's sums: lane k of EVERY wavefront adds the NW wave sums of moment k in the fixed order q = 0 .. NW-1 (the same
    // bits in every wavefront) and the results are broadcast as wave-uniform scalars.  (Every lane used to add all 6 x NW values
    // itself: 96 binary64 adds per thread — a quarter of a pass's VALU time, and 192 VGPRs of loads in flight that spilled.)
    {
        double sacc = 0.0;
        if (lane < nv) {
#pragma unroll
            for (int q = 0; q < NW; q++) sacc += red[lane][q];
        }
#pragma unroll
        for (int k = 0; k < nv; k++)
            v[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sacc), k), __builtin_amdgcn_readlane(__double2loint(sacc), k));
    }
    CS_STAMP(fine, 5);
    m.m00 = v[0], m.m10 = v[1], m.m01 = v[2], m.m11 = v[3], m.m20 = v[4], m.m02 = v[5];
    return m;
}

template <bool SECOND, int NW, int PW = NW>
__device__ __forceinline__ Mom window_moments_any(const uint32_t *__restrict__ img, int W, const double *lut, const CsRegion &R, int x, int y, int w, int h,
                                                  double (*red)[NW], unsigned long long *fine = nullptr) {
    // workgroup-uniform: the whole window lies inside the cached region (it practically always does: the region is the search
    // window plus a margin, and a mean-shift step moves the window by a few pixels)
    if (R.rw > 0 && x >= R.x0 && y >= R.y0 && w <= R.x0 + R.rw && h <= R.y0 + R.rh) return window_moments<SECOND, NW, true, PW>(img, W, lut, R, x, y, w, h, red, fine);
    return window_moments<SECOND, NW, false, PW>(img, W, lut, R, x, y, w, h, red, fine);
}

// Copies the neighbourhood of the search window (the window clamped to the frame, grown by as large a margin as `cap` pixels
// allow, at most 16) into LDS as histogram bins.  All loads are independent: one round of memory latency for the whole region.
// workgroup-uniform integers (search window, region rectangle, loop bounds) are pinned to scalar registers: every thread computes the
// same values, but the compiler cannot know that of something read from LDS and would keep them — and the whole integer side of the
// mean-shift loop — in VGPRs, of which this 1024-thread kernel has exactly 128
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ CsRegion cs_region_rect(int W, int H, const int *s_sw, uint16_t *bins, int cap) {
    CsRegion R = {bins, 0, 0, 0, 0};
    const int sw0 = uni(s_sw[0]), sw1 = uni(s_sw[1]), sw2 = uni(s_sw[2]), sw3 = uni(s_sw[3]);
    const int x0 = max(sw0, 0), y0 = max(sw1, 0), x1 = min(x0 + sw2, W), y1 = min(y0 + sw3, H);
    const int w0 = x1 - x0, h0 = y1 - y0;
    if (w0 <= 0 || h0 <= 0 || (long long)w0 * h0 > cap) return R;
    int mg = 0;
    for (int t = 16; t > 0; t >>= 1)  // largest margin <= 16 px that still fits: a mean-shift step moves the window by a few pixels
        if ((long long)(min(x1 + mg + t, W) - max(x0 - mg - t, 0)) * (min(y1 + mg + t, H) - max(y0 - mg - t, 0)) <= cap && mg + t <= 16) mg += t;
    R.x0 = max(x0 - mg, 0), R.y0 = max(y0 - mg, 0);
    R.rw = min(x1 + mg, W) - R.x0, R.rh = min(y1 + mg, H) - R.y0;
    return R;
}

template <int NT_>
__device__ __forceinline__ CsRegion cs_cache_region(const uint32_t *__restrict__ img, int W, int H, const int *s_sw, uint16_t *bins, int cap) {
    const CsRegion R = cs_region_rect(W, H, s_sw, bins, cap);
    if (R.rw == 0) return R;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int NWV = NT_ / 64;
    for (int j0 = wave; j0 < R.rh; j0 += 8 * NWV) {  // rows by wavefront, columns by lane: no division, 8 independent loads per batch
        for (int cb = 0; cb < R.rw; cb += 64) {
            const int c = min(cb + lane, R.rw - 1);
            uint32_t px[8];
#pragma unroll
            for (int r = 0; r < 8; r++) px[r] = img[(size_t)(R.y0 + min(j0 + r * NWV, R.rh - 1)) * W + (R.x0 + c)];
#pragma unroll
            for (int r = 0; r < 8; r++)
                if (cb + lane < R.rw && j0 + r * NWV < R.rh) bins[(j0 + r * NWV) * R.rw + cb + lane] = (uint16_t)cs_bin(px[r]);
        }
    }
    return R;  // the caller synchronises the workgroup before the first pass
}

// meanShift + camShift (camshift.js:222-312) once the weight LUT is in LDS; every thread runs the identical scalar logic,
// thread 0 writes the state.  NW = wavefronts of the workgroup.
// `moments(x, y, w, h)` computes camshift.Moments (all six sums) over the window for the whole workgroup (every thread gets the same Mom).
template <typename MOMENTS>
__device__ __forceinline__ void meanshift_body(int W, int H, const int *s_sw, HtCsState &st, int calc_angles, int max_it, ht_cs_trackobj *__restrict__ out_s,
                                               unsigned long long *stamps, bool writer, MOMENTS moments, uint32_t *done_flag = nullptr, uint32_t done_seq = 0u) {
    int swx = uni(s_sw[0]), swy = uni(s_sw[1]);
    int n_stamp = 4;
    (void)n_stamp;
    const int sww = uni(s_sw[2]), swh = uni(s_sw[3]);
    int prevx = swx, prevy = swy;  // camshift.js:280-281
    Mom m = {0, 0, 0, 0, 0, 0};
    int wadx = 0, wady = 0, wadw = 0, wadh = 0;
    unsigned long long visited = 0;  // window pixels read by the moment passes (B_track = 4*W*H + 4*sum(window))
    for (int it = 0; it < max_it; it++) {  // camshift.js:284-306; max_it = 10 (option cs_iters: measurement knob, wrong results)
        wadx = max(swx, 0);
        wady = max(swy, 0);
        wadw = min(wadx + sww, W);
        wadh = min(wady + swh, H);
        visited += (unsigned long long)max(wadw - wadx, 0) * (unsigned long long)max(wadh - wady, 0);
        // Every pass computes all six sums.  The reference computes the second moments only in its 10th iteration or, once the
        // window stopped moving, in ONE MORE pass over the same window (camshift.js:299-301) — whose first-moment sums are, operation
        // for operation, the ones this pass already has: the extra pass (a quarter of a typical call's passes) is not run.
        m = moments(wadx, wady, wadw, wadh);
        CS_STAMP(stamps, n_stamp);
        n_stamp++;
        if (it == 0) CS_STAMP(stamps, 22);
        const double inv = 1.0 / m.m00, xc = m.m10 * inv, yc = m.m01 * inv;  // camshift.js:109-111
        swx += uni(toint32(xc - (double)sww / 2));                               // camshift.js:295 (every thread holds the same sums)
        swy += uni(toint32(yc - (double)swh / 2));                               // camshift.js:296
        if (it == 0) CS_STAMP(stamps, 23);
        if (swx == prevx && swy == prevy) {                                      // camshift.js:299-301
            // `visited` keeps counting the reference's passes (B_track = 4 W H + 4 sum of the window passes)
            if (it != 9) visited += (unsigned long long)max(wadw - wadx, 0) * (unsigned long long)max(wadh - wady, 0);
            break;
        }
        prevx = swx;
        prevy = swy;
    }
    CS_STAMP(stamps, n_stamp);
    if (threadIdx.x != 0 || !writer) return;
    swx = max(0, min(swx, W));  // camshift.js:308-309
    swy = max(0, min(swy, H));
    const double invM00 = 1.0 / m.m00, xc = m.m10 * invM00, yc = m.m01 * invM00;
    const double mu20 = m.m20 - m.m10 * xc, mu02 = m.m02 - m.m01 * yc, mu11 = m.m11 - m.m01 * xc;  // camshift.js:116-118
    const double a = mu20 * invM00, c = mu02 * invM00;  // camshift.js:230-231
    double width, height, angle;
    if (calc_angles) {  // camshift.js:233-245
        const double b = mu11 * invM00, d = a + c;
        const double e = sqrt((4 * b * b) + ((a - c) * (a - c)));
        width = (double)(int32_t)((uint32_t)toint32(sqrt((d - e) * 0.5)) << 2);
        height = (double)(int32_t)((uint32_t)toint32(sqrt((d + e) * 0.5)) << 2);
        angle = atan2(2 * b, a - c + e);
        if (angle < 0) angle = angle + 3.141592653589793;
    } else {  // camshift.js:247-249
        width = (double)(int32_t)((uint32_t)toint32(sqrt(a)) << 2);
        height = (double)(int32_t)((uint32_t)toint32(sqrt(c)) << 2);
        angle = 3.141592653589793 / 2;
    }
    double cx = (double)swx + (double)sww / 2, cy = (double)swy + (double)swh / 2;  // camshift.js:253-254 (old window size)
    cx = cx < (double)W ? cx : (double)W;
    cy = cy < (double)H ? cy : (double)H;
    const double tx = floor(cx > 0 ? cx : 0.0), ty = floor(cy > 0 ? cy : 0.0);
    const int nsww = (int)floor(1.1 * width), nswh = (int)floor(1.1 * height);  // camshift.js:257-258
    st.sw[0] = swx, st.sw[1] = swy, st.sw[2] = nsww, st.sw[3] = nswh;
    st.x = tx, st.y = ty, st.width = width, st.height = height, st.angle = angle;
    st.win_px += visited;
    st.calls += 1;
    if (out_s) {
        ht_cs_trackobj o;
        o.x = tx, o.y = ty, o.width = width, o.height = height, o.angle = angle;
        o.sw_x = swx, o.sw_y = swy, o.sw_width = nsww, o.sw_height = nswh;
        *out_s = o;
    }
    if (done_flag) {  // enqueue-only call: the host polls this word of the pinned slot instead of waiting for an event
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the track object above is visible before the flag
        __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(CS_NT) void k_cs_meanshift(const uint8_t *__restrict__ frames, size_t frame_stride, int W, int H,
                                                        const uint32_t *__restrict__ hist, int nchunks, HtCsState *__restrict__ states,
                                                        int first, int calc_angles, int max_it, int region_cap, ht_cs_trackobj *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t cs_dyn[];  // [region_cap] u16 bins of the cached search region
    __shared__ double lut[4096];
    __shared__ double red[6][CS_NT / 64];
    __shared__ int s_sw[4];
    const int s = blockIdx.x;
    HtCsState &st = states[first + s];
    const uint32_t *cur = hist + (size_t)s * nchunks * 4096;
    const uint32_t *img = reinterpret_cast<const uint32_t *>(frames + (size_t)s * frame_stride);
    {  // getWeights, camshift.js:314-330; the frame's histogram = sum of its chunk histograms (4 bins per 16-byte load)
        const uint4 *cur4 = reinterpret_cast<const uint4 *>(cur);
        const uint4 *model4 = reinterpret_cast<const uint4 *>(st.model);
        for (int i4 = threadIdx.x; i4 < 1024; i4 += CS_NT) {
            uint4 acc = make_uint4(0u, 0u, 0u, 0u);
            for (int k = 0; k < nchunks; k++) {
                const uint4 v = cur4[(size_t)k * 1024 + i4];
                acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
            }
            const uint4 m = model4[i4];
            const uint32_t chv[4] = {acc.x, acc.y, acc.z, acc.w}, mv[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                double p = 0.0;
                if (chv[q] != 0) {
                    p = (double)mv[q] / (double)chv[q];
                    p = p < 1.0 ? p : 1.0;
                }
                lut[i4 * 4 + q] = p;
            }
        }
    }
    if (threadIdx.x < 4) s_sw[threadIdx.x] = st.sw[threadIdx.x];
    __syncthreads();
    const CsRegion R = cs_cache_region<CS_NT>(img, W, H, s_sw, reinterpret_cast<uint16_t *>(cs_dyn), region_cap);
    __syncthreads();
    meanshift_body(W, H, s_sw, st, calc_angles, max_it, out ? out + s : nullptr, nullptr, true,
                   [&](int x, int y, int w, int h) { return window_moments_any<true, CS_NT / 64>(img, W, lut, R, x, y, w, h, red); });
}

// One launch per track() call when there are enough streams to fill the chip by themselves: ONE 1024-thread workgroup per stream
// does the full-frame histogram in LDS, turns it into the weight LUT in place (no partial histograms through HBM, no second
// launch) and runs the mean-shift loop.  Per stream and call the only memory traffic left is the frame itself (4*W*H, then the
// window passes from L2), the model histogram (16 KB) and the state.
constexpr int FUSED_NT = 1024;
constexpr int CS_REGION_CAP = 40960;  // pixels of the cached search region: 80 KB of LDS next to the 32 KB LUT (which the 16 KB histogram overlays)
// The 1024-thread form owns its CU: 112 KB of LDS and 16 wavefronts x 122 VGPRs leave no room for anything else, so the track launches
// of several contexts (and the detect kernels of their next batches) run strictly one after the other although a call is a bandwidth
// phase (the frame streams through the histogram) followed by a latency phase (<= 10 dependent moment passes from LDS).  The
// 512-thread form (round 5) is half of it — 8 wavefronts, <= 128 VGPRs, a 44 KB region: 77 KB of LDS — so TWO workgroups share a CU,
// normally at different phases of their calls, or a workgroup shares it with another batch's detect kernels.  Its wavefronts play the
// 16 of the large form in window_moments, so both forms return the same bits.  Chosen per launch (fused_threads below): more streams
// than CUs, or more than one context of the device on this path.
constexpr int FUSED_NT_SMALL = 512;
constexpr int CS_REGION_CAP_SMALL = 22528;  // 44 KB: a 118 x 118 search window + 16 px of margin, or 150 x 150 without

// Up to CS_SEQ_MAX successive track() calls of every stream in ONE launch (ht_camshift_track_sequence): the frame batches of the calls
// travel as kernel arguments, a workgroup walks its stream's calls in order.  The calls of a stream depend on each other through its
// search window, the streams do not: with one launch per call every call waited for its slowest stream (and paid a launch); now a
// workgroup that converges quickly on one frame is already working on the next one.
constexpr int CS_SEQ_MAX = 64;
struct CsFrameList {
    const uint8_t *p[CS_SEQ_MAX];
};
// Everything the kernel needs travels in ONE struct = the whole kernarg segment.  The kernel reads it through the kernarg pointer,
// made opaque at the top of every call of the sequence: the ~20 scalars are then re-loaded (scalar loads from the constant cache) per
// call instead of being kept alive across the whole call body — as loop invariants they cost the 128-VGPR kernel 52 spilled VGPRs and
// 24 spilled SGPRs (212 B of scratch per lane).
struct CsFusedArgs {
    CsFrameList flist;
    int ncalls;
    int W, H;
    uint32_t npix;
    size_t frame_stride;
    HtCsState *states;
    int first, calc_angles, max_it, region_cap;
    ht_cs_trackobj *out;
    uint32_t *dbg_hist;
    uint32_t out_call_stride;
};
typedef const CsFusedArgs __attribute__((address_space(4))) *CsFusedArgsPtr;

template <bool SEQ, int NT>
__global__ __launch_bounds__(NT, 4) void k_cs_track_fused(const CsFusedArgs args_by_value) {
    constexpr int NWV = FUSED_NT / 64;  // wavefronts the moment passes are laid out for (window_moments), whatever NT is
    extern __shared__ __attribute__((aligned(16))) uint8_t cs_dyn[];  // [region_cap] u16 bins of the cached search region
    __shared__ double lut[4096];
    // the frame's histogram overlays the upper half of the LUT it is turned into (every bin is read into registers, then a barrier,
    // then the weights are written): 16 KB of LDS that the cached search region of the 512-thread form gets instead
    uint32_t *const h = reinterpret_cast<uint32_t *>(lut + 2048);
    __shared__ double red[6][NWV];
    __shared__ int s_sw[4];
    (void)args_by_value;
    CsFusedArgsPtr ka = (CsFusedArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    const int s = blockIdx.x;
    const int ncalls = SEQ ? ka->ncalls : 1;
    for (int call = 0; call < ncalls; call++) {
    if (SEQ) asm volatile("" : "+s"(ka));  // opaque: nothing loaded through ka before this point stays live across the call
    if (call) __syncthreads();  // thread 0 stored the new search window; every wavefront is done with h / lut / the cached region
    const int W = ka->W, H = ka->H, region_cap = ka->region_cap;
    const uint32_t npix = ka->npix;
    HtCsState &st = ka->states[ka->first + s];
    const uint8_t *frame = ka->flist.p[call] + (size_t)s * ka->frame_stride;
    uint32_t *const dbg_hist = ka->dbg_hist;
#ifdef HT_CS_TIMELINE
    __shared__ unsigned long long s_stamps[32];
    if (threadIdx.x < 32) s_stamps[threadIdx.x] = 0;
    unsigned long long *stamps = s_stamps;
    __syncthreads();
#else
    unsigned long long *stamps = nullptr;
#endif
    CS_STAMP(stamps, 0);
    for (int i = threadIdx.x; i < 4096; i += NT) h[i] = 0;
    if (threadIdx.x < 4) s_sw[threadIdx.x] = st.sw[threadIdx.x];
    __syncthreads();
    // the search region (search window + margin) is cached in LDS as histogram bins for the moment passes; when rows are whole
    // 16-byte groups the histogram pass below stashes it on the way (the frame is read exactly once), otherwise a separate copy
    // pass after the LUT does (cs_cache_region)
    uint16_t *const rbins = reinterpret_cast<uint16_t *>(cs_dyn);
    const int qpr = W >> 2, rpi = qpr > 0 ? NT / qpr : 0;  // 16-byte groups per row, rows per sweep of the workgroup
    const bool rows2d = (W & 3) == 0 && rpi >= 1;
    CsRegion R = {rbins, 0, 0, 0, 0};
    if (rows2d) R = cs_region_rect(W, H, s_sw, rbins, region_cap);
    {  // camshift.Histogram of the whole frame (camshift.js:268), counts merged per thread and per wavefront like k_cs_hist
        const uint4 *img4 = reinterpret_cast<const uint4 *>(frame);
        if (rows2d) {
            // thread = (row within the sweep, 16-byte group of the row): its column never changes, so the region test is two row
            // compares per load; 8 independent loads per thread in flight
            const int trow = (int)threadIdx.x / qpr, tq = (int)threadIdx.x - trow * qpr;
            const bool tact = trow < rpi;
            const int colx = 4 * tq - R.x0;  // column of the group's first pixel inside the region
            const bool colhit = R.rw > 0 && colx + 3 >= 0 && colx < R.rw;
            for (int row0 = trow; row0 - trow < H; row0 += 8 * rpi) {  // same trip count for every thread (ballots inside)
                uint4 p[8];
#pragma unroll
                for (int u = 0; u < 8; u++) p[u] = img4[(size_t)min(row0 + u * rpi, H - 1) * qpr + tq];  // clamped address, masked below
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int row = row0 + u * rpi;
                    const bool on = tact && row < H;
                    const uint32_t b0 = cs_bin(p[u].x), b1 = cs_bin(p[u].y), b2 = cs_bin(p[u].z), b3 = cs_bin(p[u].w);
                    const bool flat = (b0 == b1) && (b2 == b3) && (b0 == b2);
                    hist_add_wave(h, b0, flat ? 4u : 1u, on);
                    if (on && !flat) {
                        atomicAdd(&h[b1], 1u);
                        atomicAdd(&h[b2], 1u);
                        atomicAdd(&h[b3], 1u);
                    }
                    const int ry = row - R.y0;
                    if (on && colhit && ry >= 0 && ry < R.rh) {
                        uint16_t *d = rbins + ry * R.rw + colx;
                        if (colx >= 0) d[0] = (uint16_t)b0;
                        if (colx + 1 >= 0 && colx + 1 < R.rw) d[1] = (uint16_t)b1;
                        if (colx + 2 >= 0 && colx + 2 < R.rw) d[2] = (uint16_t)b2;
                        if (colx + 3 < R.rw) d[3] = (uint16_t)b3;
                    }
                }
            }
        } else {
            const uint32_t nquad = npix / 4;
            // 8 independent 16-byte loads per thread in flight (128 KB per workgroup): one workgroup has a whole frame to pull
            for (uint32_t i0 = threadIdx.x; i0 < nquad; i0 += 8 * NT) {
                uint4 p[8];
#pragma unroll
                for (int u = 0; u < 8; u++) p[u] = img4[min(i0 + u * NT, nquad - 1)];  // clamped address, masked below
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const bool on = i0 + u * NT < nquad;
                    const uint32_t b0 = cs_bin(p[u].x), b1 = cs_bin(p[u].y), b2 = cs_bin(p[u].z), b3 = cs_bin(p[u].w);
                    const bool flat = (b0 == b1) && (b2 == b3) && (b0 == b2);
                    hist_add_wave(h, b0, flat ? 4u : 1u, on);
                    if (on && !flat) {
                        atomicAdd(&h[b1], 1u);
                        atomicAdd(&h[b2], 1u);
                        atomicAdd(&h[b3], 1u);
                    }
                }
            }
            const uint32_t *img1 = reinterpret_cast<const uint32_t *>(frame);
            for (uint32_t i = nquad * 4 + threadIdx.x; i < npix; i += NT) atomicAdd(&h[cs_bin(img1[i])], 1u);  // < 4 pixels
        }
    }
    __syncthreads();
    CS_STAMP(stamps, 1);
    {  // getWeights, camshift.js:314-330
        const uint4 *model4 = reinterpret_cast<const uint4 *>(st.model);
        constexpr int RND = 1024 / NT;  // 4 bins per thread and round
        uint4 mq[RND], cq[RND];
#pragma unroll
        for (int rd = 0; rd < RND; rd++) {
            const int i4 = (int)threadIdx.x + rd * NT;
            mq[rd] = model4[i4];
            cq[rd] = reinterpret_cast<const uint4 *>(h)[i4];
            if (dbg_hist) reinterpret_cast<uint4 *>(dbg_hist + (size_t)s * 4096)[i4] = cq[rd];
        }
        __syncthreads();  // every bin of h is in a register: the weights may overwrite it
#pragma unroll
        for (int rd = 0; rd < RND; rd++) {
            const int i4 = (int)threadIdx.x + rd * NT;
            const uint32_t chv[4] = {cq[rd].x, cq[rd].y, cq[rd].z, cq[rd].w}, mv[4] = {mq[rd].x, mq[rd].y, mq[rd].z, mq[rd].w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                double p = 0.0;
                if (chv[q] != 0) {
                    p = (double)mv[q] / (double)chv[q];
                    p = p < 1.0 ? p : 1.0;
                }
                lut[i4 * 4 + q] = p;
            }
        }
    }
    CS_STAMP(stamps, 2);
    // the search region: the frame went through this CU a moment ago, but not into any cache that would still hold it
    if (!rows2d) R = cs_cache_region<NT>(reinterpret_cast<const uint32_t *>(frame), W, H, s_sw, rbins, region_cap);
    __syncthreads();  // LUT and region complete
    CS_STAMP(stamps, 3);
    const uint32_t *img = reinterpret_cast<const uint32_t *>(frame);
    ht_cs_trackobj *const out = ka->out;
    meanshift_body(W, H, s_sw, st, ka->calc_angles, ka->max_it, out ? out + (size_t)call * ka->out_call_stride + s : nullptr, stamps, true,
                   [&](int x, int y, int w, int h) { return window_moments_any<true, NWV, NT / 64>(img, W, lut, R, x, y, w, h, red); });
#ifdef HT_CS_TIMELINE
    if (threadIdx.x == 0 && dbg_hist)
        for (int i = 0; i < 30; i++) reinterpret_cast<unsigned long long *>(dbg_hist + (size_t)s * 4096 + 4032)[i] = s_stamps[i];
#endif
    }  // next call of this stream
}

// ---- few large streams: a CLUSTER of workgroups per stream ---------------------------------------------------------------
// One workgroup streams a window at one CU's memory rate: a 360 x 360 search window of a 1080p feed (130 k pixels, 0.5 MB per
// pass) took ~30 us per pass, 146 us per track() call.  Here G workgroups share every pass (rows interleaved over all their
// wavefronts) and publish their six partial sums with agent-scope stores into the pass's exchange slot, whose entries k_cs_lut
// marked "not written yet" (a NaN no sum can be): every workgroup polls the G x 6 entries, one thread per entry, until the mark is
// gone — the value that ends the wait is the partial sum itself (round 5; until then: store, drain, arrive on a counter, poll the
// counter, read the partials = three dependent L2 round trips, ~3 us per pass).  No agent-scope fences: every payload word is an
// sc1 store on one side and an sc1 load on the other.  Every workgroup sums the G partials in the same order, so all of them take
// identical mean-shift decisions; workgroup 0 writes the state.  The grid (streams x G <= 256 workgroups) is always co-resident: the spin cannot starve a workgroup that has not started.
constexpr int CL_NT = 512, CL_MAXG = 32, CL_SLOTS = 12;  // <= 11 moment passes per call (camshift.js:284-306)
// "not written yet" mark of an exchange entry: a NaN no moment sum can be (the sums are finite and >= 0)
constexpr unsigned long long CL_UNWRITTEN = 0xFFF8C0DEC0DE0001ull;

// weight LUT of every stream from its chunk histograms (getWeights, camshift.js:314-330): grid (64, streams) x 512 threads;
// a block owns 64 bins, its 8 wavefronts each sum every 8th chunk (a single 1080p stream has 127 chunk histograms = 2 MB)
__global__ __launch_bounds__(512) void k_cs_lut(const uint32_t *__restrict__ hist, int nchunks, const HtCsState *__restrict__ states, int first,
                                                double *__restrict__ lut, unsigned long long *__restrict__ cluster_parts) {
    __shared__ uint32_t part[8][64];
    const int s = blockIdx.y, lane = threadIdx.x & 63, grp = threadIdx.x >> 6, bin = blockIdx.x * 64 + lane;
    {   // the stream's exchange slots of the cluster launch that follows: every entry "not written yet" (was a memset of its own)
        const uint32_t i = blockIdx.x * 512u + threadIdx.x;
        if (i < (uint32_t)(CL_SLOTS * CL_MAXG * 6)) cluster_parts[(size_t)s * CL_SLOTS * CL_MAXG * 6 + i] = CL_UNWRITTEN;
    }
    const uint32_t *cur = hist + (size_t)s * nchunks * 4096 + bin;
    uint32_t ch = 0;
#pragma unroll 4
    for (int k = grp; k < nchunks; k += 8) ch += cur[(size_t)k * 4096];
    part[grp][lane] = ch;
    __syncthreads();
    if (grp == 0) {
        ch = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) ch += part[q][lane];
        double p = 0.0;
        if (ch != 0) {
            p = (double)states[first + s].model[bin] / (double)ch;
            p = p < 1.0 ? p : 1.0;
        }
        lut[(size_t)s * 4096 + bin] = p;
    }
}

// The wait is a spin on agent-scope loads, so it is BOUNDED: a thread that has waited `budget` shader-clock cycles (a quarter of a
// second by default, against ~1-2 us for a healthy exchange) raises the context's error word and its workgroup stops waiting — at this
// and every later exchange of the call (s_timeout is sticky).  The host reports HT_ERR_STATE with the next result read-back; the stream's
// state is then garbage for this call, but nothing hangs.  Co-residency (the premise of the spin) is arranged by the host: one cluster
// launch in flight per device and process, grid <= one workgroup per CU (launch_track).
// Ordering: an entry is ONE 8-byte word, written by an sc1 (agent-scope) store and read by sc1 loads — single-copy atomic, nothing else
// depends on it.  That is the gfx9 memory model; ht_create refuses any other arch.
struct ClusterSync {
    uint32_t *err;
    uint32_t *err_host;  // pinned host word (plain system-scope store of 1): an enqueue-only call's host side reads it without a copy
    long long budget;
    int *s_timeout;  // LDS flag of the workgroup
};
template <bool SECOND>
__device__ __forceinline__ Mom cluster_moments(const uint32_t *__restrict__ img, int W, const double *lut, int x, int y, int w, int h, double (*red)[CL_NT / 64],
                                               double *s_part, int g, int G, double *__restrict__ parts, const ClusterSync &sync, int slot) {
    constexpr int NW = CL_NT / 64, nv = SECOND ? 6 : 3;
    Mom m = {0, 0, 0, 0, 0, 0};
    const int ww = w - x, hh = h - y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (ww > 0 && hh > 0) {
        for (int j = g * NW + wave; j < hh; j += G * NW) {  // a row per (workgroup, wavefront); 8 column chunks of the row in flight
            double rs = 0.0, ts = 0.0, us = 0.0;
            const uint32_t *rowp = img + (size_t)(y + j) * W + x;
            for (int cb = 0; cb < ww; cb += 512) {
                uint32_t px[8];
#pragma unroll
                for (int u = 0; u < 8; u++) px[u] = rowp[min(cb + 64 * u + lane, ww - 1)];  // clamped address, value masked below
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int c = cb + 64 * u + lane;
                    const double val = c < ww ? lut[cs_bin(px[u])] : 0.0;
                    const double vx = (double)c;
                    rs += val;
                    ts += vx * val;
                    if (SECOND) us += vx * vx * val;
                }
            }
            const double vy = (double)j;
            m.m00 += rs;
            m.m10 += ts;
            m.m01 += vy * rs;
            if (SECOND) {
                m.m11 += vy * ts;
                m.m20 += us;
                m.m02 += vy * vy * rs;
            }
        }
    }
    double v[6] = {m.m00, m.m10, m.m01, m.m11, m.m20, m.m02};
    __syncthreads();  // red[] / s_part[] may still be read from the previous pass
#pragma unroll
    for (int k = 0; k < nv; k++) {
        const double sum = wave_sum_f64(v[k]);
        if (lane == 0) red[k][wave] = sum;
    }
    __syncthreads();
    // this workgroup's partial sums -> its entries of the pass's exchange slot (agent-scope stores, fire and forget); then every entry
    // of the slot is polled by a thread of its own until it no longer holds the "not written yet" mark k_cs_lut left there: the value
    // that ends the wait IS the partial sum — no arrival counter, no drain of the stores, no second read (a pass used to be
    // store -> s_waitcnt -> atomic add -> poll the counter -> read the G partials: three dependent round trips through L2).
    unsigned long long *slot_parts = reinterpret_cast<unsigned long long *>(parts) + (size_t)slot * CL_MAXG * 6;
    if (threadIdx.x < nv) {
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < NW; q++) sum += red[threadIdx.x][q];
        __hip_atomic_store(&slot_parts[g * 6 + threadIdx.x], (unsigned long long)__double_as_longlong(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((int)threadIdx.x < G * 6) {
        unsigned long long bits = 0ull;  // +0.0 for the entries a first-moment pass does not use
        if ((int)(threadIdx.x % 6u) < nv) {
            bits = __hip_atomic_load(&slot_parts[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bits == CL_UNWRITTEN && !*sync.s_timeout) {
                const long long t0 = (long long)__builtin_readcyclecounter();
                for (;;) {
                    __builtin_amdgcn_s_sleep(1);
                    bits = __hip_atomic_load(&slot_parts[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (bits != CL_UNWRITTEN) break;
                    if ((long long)__builtin_readcyclecounter() - t0 > sync.budget) {  // bounded spin: give up, flag it, never wait again
                        *sync.s_timeout = 1;
                        atomicOr(sync.err, 1u);
                        if (sync.err_host) __hip_atomic_store(sync.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
            if (bits == CL_UNWRITTEN) bits = 0ull;  // timed out: the call's result is undefined (reported), but it stays a number
        }
        s_part[threadIdx.x] = __longlong_as_double((long long)bits);
    }
    __syncthreads();
    {  // lane k of every wavefront adds moment k's G partials in the fixed order q = 0 .. G-1: every workgroup of the cluster gets the same bits
        double sacc = 0.0;
        if (lane < nv)
            for (int q = 0; q < G; q++) sacc += s_part[q * 6 + lane];
#pragma unroll
        for (int k = 0; k < nv; k++)
            v[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sacc), k), __builtin_amdgcn_readlane(__double2loint(sacc), k));
    }
    m.m00 = v[0], m.m10 = v[1], m.m01 = v[2], m.m11 = v[3], m.m20 = v[4], m.m02 = v[5];
    return m;
}

__global__ __launch_bounds__(CL_NT) void k_cs_meanshift_cluster(const uint8_t *__restrict__ frames, size_t frame_stride, int W, int H, const double *__restrict__ lut_g,
                                                                HtCsState *__restrict__ states, int first, int calc_angles, int max_it, int G,
                                                                double *__restrict__ parts,
                                                                uint32_t *__restrict__ err, uint32_t *__restrict__ err_host, long long budget,
                                                                ht_cs_trackobj *__restrict__ out, uint32_t *__restrict__ done_flags, uint32_t done_seq) {
    __shared__ double lut[4096];
    __shared__ double red[6][CL_NT / 64];
    __shared__ double s_part[CL_MAXG * 6];
    __shared__ int s_sw[4];
    __shared__ int s_timeout;
    if (threadIdx.x == 0) s_timeout = 0;
    const int s = blockIdx.x / G, g = blockIdx.x - s * G;
    HtCsState &st = states[first + s];
    const uint32_t *img = reinterpret_cast<const uint32_t *>(frames + (size_t)s * frame_stride);
    {
        const double2 *src = reinterpret_cast<const double2 *>(lut_g + (size_t)s * 4096);
        for (int i = threadIdx.x; i < 2048; i += CL_NT) reinterpret_cast<double2 *>(lut)[i] = src[i];
    }
    if (threadIdx.x < 4) s_sw[threadIdx.x] = st.sw[threadIdx.x];
    __syncthreads();
    double *my_parts = parts + (size_t)s * CL_SLOTS * CL_MAXG * 6;
    const ClusterSync sync = {err, err_host, budget, &s_timeout};
    int slot = 0;
    meanshift_body(W, H, s_sw, st, calc_angles, max_it, out ? out + s : nullptr, nullptr, g == 0, [&](int x, int y, int w, int h) {
        const int sl = slot++;
        return cluster_moments<true>(img, W, lut, x, y, w, h, red, s_part, g, G, my_parts, sync, sl);
    }, done_flags ? done_flags + s : nullptr, done_seq);
}

}  // namespace

// One cluster launch (k_cs_meanshift_cluster) in flight per device and process: its workgroups spin on each other, so two such grids
// from different contexts must not share the chip's workgroup slots (each could hold the slots the other's missing workgroups need).
// The serialisation is device-side — the launching stream waits for the event recorded after the previous cluster launch on that
// device — so no host thread blocks.
namespace {
struct ClusterGate {
    std::mutex mu;
    struct Dev {
        hipEvent_t last = nullptr;            // recorded after the most recent cluster launch (only while `multi`)
        std::vector<const ht_ctx *> users;    // contexts that have launched a cluster grid on this device
        std::vector<const ht_ctx *> fused;    // contexts that have launched k_cs_track_fused on this device (fused_threads)
        bool multi = false;                   // more than one user: every launch waits for `last` and records it
    };
    std::map<int, Dev> dev;
};
ClusterGate &cluster_gate() {
    static ClusterGate g;
    return g;
}

}  // namespace
// ht_destroy: the context stops counting as a user of its device's cluster gate (its stream has been synchronised)
void ht_cluster_gate_forget(const ht_ctx *c) {
    ClusterGate &gate = cluster_gate();
    std::lock_guard<std::mutex> lk(gate.mu);
    auto it = gate.dev.find(c->device);
    if (it == gate.dev.end()) return;
    auto &u = it->second.users;
    u.erase(std::remove(u.begin(), u.end(), c), u.end());
    auto &fu = it->second.fused;
    fu.erase(std::remove(fu.begin(), fu.end(), c), fu.end());
    if (u.size() <= 1 && it->second.multi) {
        // back to one user: its recorded grids are ordered by its own stream from here on
        it->second.multi = false;
    }
}
// ht_detect_enqueue around its graph capture.  Setting the flag takes the gate's lock, and fused_threads() holds that lock across its
// queries: a query of this context's stream either completed before the capture began or sees the flag and is skipped.
void ht_capture_mark(ht_ctx *c, bool on) {
    if (on) {
        std::lock_guard<std::mutex> lk(cluster_gate().mu);
        c->capturing.store(true);
    } else {
        c->capturing.store(false);
    }
}
namespace {
// fetched with every result read-back: a cluster barrier that ran out of its cycle budget surfaces as a status code
ht_status cs_check_err(ht_ctx *c, const char *where) {
    const bool direct = c->h_cs_err_direct && __atomic_load_n(c->h_cs_err_direct, __ATOMIC_ACQUIRE) != 0;
    if (!direct && (!c->h_cs_err || *c->h_cs_err == 0)) return HT_OK;
    if (c->h_cs_err) *c->h_cs_err = 0;
    if (c->h_cs_err_direct) __atomic_store_n(c->h_cs_err_direct, 0u, __ATOMIC_RELEASE);
    (void)hipMemsetAsync(c->d_cs_err, 0, sizeof(uint32_t), c->stream);
    return ht_fail(c, HT_ERR_STATE, std::string(where) + ": a camshift cluster barrier timed out (workgroups of one stream were not co-resident); the affected streams' state is undefined — re-initialise them");
}
}  // namespace

extern "C" ht_status ht_camshift_reserve(ht_ctx *c, int32_t nstreams) {
    if (!c || nstreams <= 0) return HT_ERR_INVALID;
    HT_HIP(c, hipSetDevice(c->device));
    if (c->cs_streams >= nstreams) return HT_OK;
    HT_HIP(c, hipStreamSynchronize(c->stream));
    // every new buffer first; the context only changes once all of them exist (a failed reservation keeps the old trackers usable)
    const int ncl = std::min(nstreams, 64);  // the cluster path is only taken for <= 64 streams per call
    HtCsState *ns = nullptr;
    uint32_t *nhist = nullptr, *nerr = c->d_cs_err, *herr = c->h_cs_err;
    ht_cs_trackobj *nout = nullptr;
    double *nlut = nullptr, *nparts = nullptr;
    bool ok = hipMalloc(&ns, sizeof(HtCsState) * (size_t)nstreams) == hipSuccess &&
              hipMalloc(&nhist, sizeof(uint32_t) * 4096 * hist_max_chunks(nstreams) * (size_t)nstreams) == hipSuccess &&
              hipMalloc(&nout, sizeof(ht_cs_trackobj) * (size_t)nstreams) == hipSuccess &&
              // cluster mean-shift (few large streams): per stream a LUT and CL_SLOTS x CL_MAXG partial-sum slots
              hipMalloc(&nlut, sizeof(double) * 4096 * (size_t)ncl) == hipSuccess &&
              hipMalloc(&nparts, sizeof(double) * CL_SLOTS * CL_MAXG * 6 * (size_t)ncl) == hipSuccess;
    if (ok && !nerr) ok = hipMalloc(&nerr, sizeof(uint32_t)) == hipSuccess && hipMemset(nerr, 0, sizeof(uint32_t)) == hipSuccess;
    if (ok && !herr) {
        ok = hipHostMalloc(reinterpret_cast<void **>(&herr), sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
        if (ok) *herr = 0;
    }
    if (ok) ok = hipMemset(ns, 0, sizeof(HtCsState) * (size_t)nstreams) == hipSuccess;
    if (ok && c->d_cs)  // keep existing trackers
        ok = hipMemcpy(ns, c->d_cs, sizeof(HtCsState) * (size_t)c->cs_streams, hipMemcpyDeviceToDevice) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        if (ns) (void)hipFree(ns);
        if (nhist) (void)hipFree(nhist);
        if (nout) (void)hipFree(nout);
        if (nlut) (void)hipFree(nlut);
        if (nparts) (void)hipFree(nparts);
        if (nerr && nerr != c->d_cs_err) (void)hipFree(nerr);
        if (herr && herr != c->h_cs_err) (void)hipHostFree(herr);
        return ht_fail(c, HT_ERR_NOMEM, "ht_camshift_reserve: allocation failed (the previous reservation is unchanged)");
    }
    if (c->d_cs) (void)hipFree(c->d_cs);
    if (c->d_cs_hist) (void)hipFree(c->d_cs_hist);
    if (c->d_cs_out) (void)hipFree(c->d_cs_out);
    if (c->d_cs_lut) (void)hipFree(c->d_cs_lut);
    if (c->d_cs_parts) (void)hipFree(c->d_cs_parts);
    c->d_cs = ns, c->d_cs_hist = nhist, c->d_cs_out = nout, c->d_cs_lut = nlut, c->d_cs_parts = nparts;
    c->d_cs_err = nerr, c->h_cs_err = herr;
    c->cs_last_n = c->cs_last_chunks = 0;  // the debug histogram buffer is new
    c->cs_last_hist = nullptr;
    c->cs_streams = nstreams;
    // result ring of the enqueue-only track calls (the stream was synchronised above: no slot is in use; uncollected results are dropped)
    c->cs_ring_head = c->cs_ring_count = 0;
    if (!c->h_cs_err_direct) {
        if (hipHostMalloc(reinterpret_cast<void **>(&c->h_cs_err_direct), sizeof(uint32_t), hipHostMallocDefault) != hipSuccess)
            return ht_fail(c, HT_ERR_NOMEM, "ht_camshift_reserve: hipHostMalloc failed");
        *c->h_cs_err_direct = 0;
    }
    for (auto &sl : c->cs_ring) {
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        sl.h_out = nullptr;
        if (hipHostMalloc(reinterpret_cast<void **>(&sl.h_out), sizeof(ht_cs_trackobj) * (size_t)nstreams, hipHostMallocDefault) != hipSuccess) {
            c->cs_ring_streams = 0;
            return ht_fail(c, HT_ERR_NOMEM, "ht_camshift_reserve: hipHostMalloc failed");
        }
        if (!sl.ev && hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) return ht_fail(c, HT_ERR_HIP, "ht_camshift_reserve: hipEventCreate failed");
        if (sl.h_flag) (void)hipHostFree(sl.h_flag);
        sl.h_flag = nullptr, sl.seq = 0;
        if (hipHostMalloc(reinterpret_cast<void **>(&sl.h_flag), sizeof(uint32_t) * (size_t)nstreams, hipHostMallocDefault) != hipSuccess) {
            c->cs_ring_streams = 0;
            return ht_fail(c, HT_ERR_NOMEM, "ht_camshift_reserve: hipHostMalloc failed");
        }
        std::memset(sl.h_flag, 0, sizeof(uint32_t) * (size_t)nstreams);
    }
    c->cs_ring_streams = nstreams;
    return HT_OK;
}

extern "C" ht_status ht_camshift_init_batch(ht_ctx *c, int32_t first, int32_t n, const ht_cs_rect *rects) {
    HtRange range("ht_camshift_init_batch");
    if (!c || !rects) return HT_ERR_INVALID;
    if (!c->d_frames || n <= 0 || n > c->nframes) return ht_fail(c, HT_ERR_STATE, "ht_camshift_init_batch: bind n frames first");
    if (first < 0 || first + n > c->cs_streams) return ht_fail(c, HT_ERR_INVALID, "ht_camshift_init_batch: stream range not reserved");
    HT_HIP(c, hipSetDevice(c->device));
    ht_cs_rect *d_rects = reinterpret_cast<ht_cs_rect *>(c->d_cs_out);  // scratch: sizeof(ht_cs_trackobj) >= sizeof(ht_cs_rect); enqueue-only track calls keep their results in the pinned ring
    // rects[] is the caller's (pageable) memory: staged in a pinned buffer of the context, so the copy is asynchronous and the call
    // does not wait for the stream — a streaming host enqueues the next track step right behind initTracker (the stream synchronisation
    // that used to end this call was 15-20 us of idle GPU per detect step of the C5 loop)
    if (c->h_cs_rects_cap < n) {
        if (c->h_cs_rects) {
            HT_HIP(c, hipStreamSynchronize(c->stream));
            (void)hipHostFree(c->h_cs_rects);
        }
        c->h_cs_rects = nullptr, c->h_cs_rects_cap = 0;
        if (hipHostMalloc(reinterpret_cast<void **>(&c->h_cs_rects), sizeof(ht_cs_rect) * (size_t)c->cs_streams, hipHostMallocDefault) != hipSuccess)
            return ht_fail(c, HT_ERR_NOMEM, "ht_camshift_init_batch: hipHostMalloc failed");
        c->h_cs_rects_cap = c->cs_streams;
        if (!c->ev_cs_rects) HT_HIP(c, hipEventCreateWithFlags(&c->ev_cs_rects, hipEventDisableTiming));
    } else {
        HT_HIP(c, hipEventSynchronize(c->ev_cs_rects));  // the previous call's copy has left the staging buffer (long ago, normally)
    }
    std::memcpy(c->h_cs_rects, rects, sizeof(ht_cs_rect) * (size_t)n);
    HT_HIP(c, hipMemcpyAsync(d_rects, c->h_cs_rects, sizeof(ht_cs_rect) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HT_HIP(c, hipEventRecord(c->ev_cs_rects, c->stream));
    {
        HtProfScope ps(c, "cs_init");
        // few streams with tall rects: rows spread over G workgroups per stream (one workgroup per stream would leave the chip idle)
        int max_rh = 0;
        for (int i = 0; i < n; i++) max_rh = std::max(max_rh, rects[i].height);
        const int G = std::min(std::min(32, std::max(1, c->num_cus * 2 / std::max(n, 1))), (max_rh + 15) / 16);
        if (n < 64 && G >= 2) {
            HT_HIP(c, hipMemset2DAsync(c->d_cs[first].model, sizeof(HtCsState), 0, sizeof(uint32_t) * 4096, (size_t)n, c->stream));
            hipLaunchKernelGGL(k_cs_init_rows, dim3(G, n), dim3(256), 0, c->stream, c->d_frames, c->frame_stride, c->W, c->H, d_rects, c->d_cs, first);
        } else {
            hipLaunchKernelGGL(k_cs_init, dim3(n), dim3(INIT_NT), 0, c->stream, c->d_frames, c->frame_stride, c->W, c->H, d_rects, c->d_cs, first);
        }
        HT_HIP(c, hipGetLastError());
    }
    return HT_OK;
}

// Threads per workgroup of k_cs_track_fused for a launch of n streams: option cs_fused_nt, else the small form when the launch has
// more workgroups than the device has CUs (all of them resident at once, two per CU) or when ANOTHER live context of the device that
// uses this path has work in flight right now (hipStreamQuery on its stream: no packet, ~1 us) — its track launch, or the detect
// kernels of its next batch, then share the CUs with this launch instead of queueing behind it —, else the large form (one stream per
// CU with the whole CU to itself: the lowest latency, and the right choice whenever nothing else wants the chip: measured on C3 with
// TWO steps in flight, where a context's track launch never meets the other's, the small form costs 18 %).  Both forms return the
// same bits.
static int fused_threads(ht_ctx *c, int n) {
    if (c->cs_fused_nt == FUSED_NT || c->cs_fused_nt == FUSED_NT_SMALL) return c->cs_fused_nt;
    bool other_busy = false;
    {
        ClusterGate &gate = cluster_gate();
        std::lock_guard<std::mutex> lk(gate.mu);  // ht_destroy forgets a context under this lock before it destroys its stream
        auto &fu = gate.dev[c->device].fused;
        if (std::find(fu.begin(), fu.end(), c) == fu.end()) fu.push_back(c);
        for (const ht_ctx *o : fu) {
            if (o == c || !o->stream) continue;
            // a context that is capturing its detect sequence is about to have work in flight; its stream must not be queried
            // (hipErrorStreamCaptureUnsupported, and the capture may be invalidated: ADVICE round 5)
            if (o->capturing.load() || hipStreamQuery(o->stream) == hipErrorNotReady) {
                other_busy = true;
                break;
            }
        }
        (void)hipGetLastError();  // hipErrorNotReady is an answer, not an error: keep it out of the launch checks that follow
    }
    return (n > c->num_cus || other_busy) ? FUSED_NT_SMALL : FUSED_NT;
}

// one track() call of streams [first, first + n) on frames[0..n): histogram pass + mean-shift, results to d_out[0..n)
// done_flags (pinned, n words) / done_seq: completion marks for an enqueue-only call — written by the cluster kernel if that path is
// taken (*flags_used = true), otherwise the caller records its event.
static ht_status launch_track(ht_ctx *c, const uint8_t *frames, size_t frame_stride, int32_t first, int32_t n, int32_t calc_angles,
                              ht_cs_trackobj *d_out, uint32_t *done_flags = nullptr, uint32_t done_seq = 0u, bool *flags_used = nullptr) {
    const uint32_t npix = (uint32_t)((size_t)c->W * c->H);
    if (!c->cs_attr_set) {  // the cached search region needs more than the default 64 KB of LDS per workgroup (per context = per device)
        HT_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_cs_track_fused<false, FUSED_NT>), hipFuncAttributeMaxDynamicSharedMemorySize, CS_REGION_CAP * 2));
        HT_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_cs_meanshift), hipFuncAttributeMaxDynamicSharedMemorySize, CS_REGION_CAP * 2));
        c->cs_attr_set = true;
    }
    // enough streams to keep (most of) the 256 CUs busy with one workgroup each: the fused single-launch kernel; fewer streams
    // (a handful of large feeds): chunk histograms from every CU, then one mean-shift workgroup per stream
    if (n >= c->cs_fused_min_streams) {
        const bool small = fused_threads(c, n) == FUSED_NT_SMALL;
        HtProfScope ps(c, small ? "cs_track_512" : "cs_track");  // the timer's name tells the form
        c->cs_fused_launches[small ? 1 : 0]++;
        CsFusedArgs ka;
        std::memset(&ka, 0, sizeof(ka));
        ka.flist.p[0] = frames;
        ka.ncalls = 1, ka.W = c->W, ka.H = c->H, ka.npix = npix, ka.frame_stride = frame_stride, ka.states = c->d_cs, ka.first = first;
        ka.calc_angles = calc_angles, ka.max_it = c->dbg_cs_iters, ka.region_cap = c->cs_region_cap, ka.out = d_out, ka.out_call_stride = 0u;
        ka.dbg_hist = c->cs_keep_hist ? c->d_cs_hist : nullptr;
        if (small) {
            ka.region_cap = std::min(ka.region_cap, CS_REGION_CAP_SMALL);
            hipLaunchKernelGGL((k_cs_track_fused<false, FUSED_NT_SMALL>), dim3(n), dim3(FUSED_NT_SMALL), (size_t)CS_REGION_CAP_SMALL * 2, c->stream, ka);
        } else {
            hipLaunchKernelGGL((k_cs_track_fused<false, FUSED_NT>), dim3(n), dim3(FUSED_NT), (size_t)CS_REGION_CAP * 2, c->stream, ka);
        }
        HT_HIP(c, hipGetLastError());
        c->cs_last_first = first, c->cs_last_n = n, c->cs_last_chunks = c->cs_keep_hist ? 1 : 0;
        c->cs_last_hist = c->d_cs_hist;
        return HT_OK;
    }
    uint32_t chunk_px, nchunks;
    hist_chunks(npix, hist_max_chunks(c->cs_streams), &chunk_px, &nchunks);  // buffer sized for cs_streams x that many chunks
    // a few large frames: G workgroups per stream share every moment pass (k_cs_meanshift_cluster); otherwise one workgroup per stream
    // cluster size: the grid never exceeds one workgroup per CU of THIS device, so it is co-resident whatever else is resident
    // (a CU has room for four of these workgroups); fewer than 4 workgroups per stream are not worth the barriers
    const int G = std::min(CL_MAXG, c->num_cus / std::max(n, 1));
    const bool cluster = c->cs_cluster && n <= 64 && G >= 4 && npix >= c->cs_cluster_min_px && c->dbg_cs_iters > 0;
    uint32_t *hist = c->d_cs_hist;
    double *lut = c->d_cs_lut;
    {
        HtProfScope ps(c, "cs_hist");
        hipLaunchKernelGGL(k_cs_hist, dim3(nchunks, n), dim3(HIST_NT), 0, c->stream, frames, frame_stride, npix, chunk_px, hist);
        HT_HIP(c, hipGetLastError());
    }
    if (cluster) {
        HtProfScope ps(c, "cs_lut");
        hipLaunchKernelGGL(k_cs_lut, dim3(64, n), dim3(512), 0, c->stream, hist, (int)nchunks, c->d_cs, first, lut,
                           reinterpret_cast<unsigned long long *>(c->d_cs_parts));
        HT_HIP(c, hipGetLastError());
    }
    if (cluster) {
        HtProfScope ps(c, "cs_meanshift");
        if (flags_used) *flags_used = done_flags != nullptr;
        ClusterGate &gate = cluster_gate();
        std::lock_guard<std::mutex> lk(gate.mu);
        ClusterGate::Dev &gd = gate.dev[c->device];
        if (std::find(gd.users.begin(), gd.users.end(), c) == gd.users.end()) {
            // A single context's cluster grids are ordered by its own stream: no event traffic (a record is a barrier packet of its own
            // between this step's last and the next step's first kernel).  When a second context of the device starts using the
            // cluster path, the unrecorded grids of the first are drained once, and from then on every launch waits and records.
            if (!gd.users.empty() && !gd.multi) {
                HT_HIP(c, hipDeviceSynchronize());
                gd.multi = true;
            }
            gd.users.push_back(c);
        }
        if (gd.multi) {
            if (!gd.last) HT_HIP(c, hipEventCreateWithFlags(&gd.last, hipEventDisableTiming));
            else HT_HIP(c, hipStreamWaitEvent(c->stream, gd.last, 0));  // the previous cluster grid on this device (any context) has drained
        }
        hipLaunchKernelGGL(k_cs_meanshift_cluster, dim3(n * G), dim3(CL_NT), 0, c->stream, frames, frame_stride, c->W, c->H, lut, c->d_cs, first,
                           calc_angles, c->dbg_cs_iters, G, c->d_cs_parts, c->d_cs_err, c->h_cs_err_direct, (long long)c->cs_barrier_budget, d_out,
                           done_flags, done_seq);
        HT_HIP(c, hipGetLastError());
        if (gd.multi) HT_HIP(c, hipEventRecord(gd.last, c->stream));
    } else {
        HtProfScope ps(c, "cs_meanshift");
        hipLaunchKernelGGL(k_cs_meanshift, dim3(n), dim3(CS_NT), (size_t)CS_REGION_CAP * 2, c->stream, frames, frame_stride, c->W, c->H, hist, (int)nchunks,
                           c->d_cs, first, calc_angles, c->dbg_cs_iters, c->cs_region_cap, d_out);
        HT_HIP(c, hipGetLastError());
    }
    c->cs_last_hist = hist;
    c->cs_last_first = first, c->cs_last_n = n, c->cs_last_chunks = (int)nchunks;
    return HT_OK;
}

extern "C" ht_status ht_camshift_track_batch(ht_ctx *c, int32_t first, int32_t n, int32_t calc_angles, ht_cs_trackobj *out) {
    HtRange range("ht_camshift_track_batch");
    if (!c) return HT_ERR_INVALID;
    if (!c->d_frames || n <= 0 || n > c->nframes) return ht_fail(c, HT_ERR_STATE, "ht_camshift_track_batch: bind n frames first");
    if (first < 0 || first + n > c->cs_streams) return ht_fail(c, HT_ERR_INVALID, "ht_camshift_track_batch: stream range not reserved");
    if (c->W == 0 || c->H == 0) return HT_OK;  // camshift.js:219
    HT_HIP(c, hipSetDevice(c->device));
    // A synchronous call with nothing outstanding goes the same way as an enqueue-only call that is collected at once: the kernels write
    // the track objects into a pinned slot and mark it, the host polls the marks — no device-to-host copies (two copy kernels: objects and
    // the error word) and no stream synchronisation.  Wall clock of a single-stream track() call, round 6 (tools/gpu_cs_wall.py):
    // 41.4 -> 28.9 us at 320x240 (event-marked slot), 38.7 -> 20.7 us at 640x480 and 41.0 -> 22.6 us at 1920x1080 (kernel-marked slot).
    const bool via_ring = out && c->cs_sync_ring && c->cs_ring_count == 0 && n <= c->cs_ring_streams;
    if (!out || via_ring) {  // enqueue only: results go straight to the next pinned slot, a mark or an event says they are complete
        if (n > c->cs_ring_streams) return ht_fail(c, HT_ERR_STATE, "ht_camshift_track_batch: no result ring for this many streams (ht_camshift_reserve failed to allocate it)");
        if (c->cs_ring_count == ht_ctx::HT_CS_RING)
            return ht_fail(c, HT_ERR_STATE, "ht_camshift_track_batch: too many enqueue-only calls outstanding (collect with ht_camshift_track_collect)");
        ht_ctx::HtCsSlot &sl = c->cs_ring[(c->cs_ring_head + c->cs_ring_count) % ht_ctx::HT_CS_RING];
        bool flagged = false;
        uint32_t seq = 0;
        if (c->cs_flags && sl.h_flag) {
            seq = ++c->cs_flag_seq ? c->cs_flag_seq : ++c->cs_flag_seq;  // never 0
            for (int i = 0; i < n; i++) __atomic_store_n(&sl.h_flag[i], 0u, __ATOMIC_RELAXED);
        }
        ht_status st = launch_track(c, c->d_frames, c->frame_stride, first, n, calc_angles, sl.h_out, seq ? sl.h_flag : nullptr, seq, &flagged);
        if (st != HT_OK) return st;
        sl.seq = flagged ? seq : 0u;
        if (!flagged) HT_HIP(c, hipEventRecord(sl.ev, c->stream));
        sl.n = n;
        c->cs_ring_count++;
        return via_ring ? ht_camshift_track_collect(c, n, out) : HT_OK;
    }
    ht_status st = launch_track(c, c->d_frames, c->frame_stride, first, n, calc_angles, c->d_cs_out);
    if (st != HT_OK) return st;
    if (out) {
        HT_HIP(c, hipMemcpyAsync(out, c->d_cs_out, sizeof(ht_cs_trackobj) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HT_HIP(c, hipMemcpyAsync(c->h_cs_err, c->d_cs_err, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HT_HIP(c, hipStreamSynchronize(c->stream));
        return cs_check_err(c, "ht_camshift_track_batch");
    }
    return HT_OK;
}

extern "C" ht_status ht_camshift_track_collect(ht_ctx *c, int32_t n, ht_cs_trackobj *out) {
    HtRange range("ht_camshift_track_collect");
    if (!c || !out || n <= 0) return HT_ERR_INVALID;
    if (c->cs_ring_count == 0 || c->cs_ring[c->cs_ring_head].n != n)
        return ht_fail(c, HT_ERR_STATE, "ht_camshift_track_collect: no enqueue-only ht_camshift_track_batch of n streams is pending");
    HT_HIP(c, hipSetDevice(c->device));
    ht_ctx::HtCsSlot &sl = c->cs_ring[c->cs_ring_head];
    if (sl.seq) {  // the kernel marks every stream's slot word; poll them (the OLDEST outstanding call; later ones keep running)
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) {
            uint64_t spins = 0;
            while (__atomic_load_n(&sl.h_flag[i], __ATOMIC_ACQUIRE) != sl.seq) {
                if ((++spins & 0xfff) == 0) {  // not there after a while: is the stream still running at all?
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
                        const hipError_t q = hipStreamQuery(c->stream);
                        if (q == hipSuccess) {
                            if (__atomic_load_n(&sl.h_flag[i], __ATOMIC_ACQUIRE) == sl.seq) break;
                            return ht_fail(c, HT_ERR_HIP, "ht_camshift_track_collect: the stream drained without the call's completion mark");
                        }
                        if (q != hipErrorNotReady) return ht_fail(c, HT_ERR_HIP, std::string("ht_camshift_track_collect: ") + hipGetErrorString(q));
                        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
                            return ht_fail(c, HT_ERR_HIP, "ht_camshift_track_collect: timed out waiting for the call's completion mark");
                    }
                }
            }
        }
    } else {
        HT_HIP(c, hipEventSynchronize(sl.ev));  // the OLDEST outstanding call; later ones keep running
    }
    std::memcpy(out, sl.h_out, sizeof(ht_cs_trackobj) * (size_t)n);
    c->cs_ring_head = (c->cs_ring_head + 1) % ht_ctx::HT_CS_RING;
    c->cs_ring_count--;
    return cs_check_err(c, "ht_camshift_track_collect");
}

extern "C" ht_status ht_camshift_track_sequence(ht_ctx *c, int32_t first, int32_t n, int32_t calc_angles, const void *const *dev_frames,
                                                int32_t ncalls, size_t frame_stride, ht_cs_trackobj *out, int32_t out_all) {
    HtRange range("ht_camshift_track_sequence");
    if (!c || !dev_frames) return HT_ERR_INVALID;
    if (c->W == 0) return ht_fail(c, HT_ERR_STATE, "ht_camshift_track_sequence: call ht_set_geometry first");
    if (n <= 0 || ncalls <= 0 || frame_stride < (size_t)c->W * c->H * 4 || (frame_stride & 3))
        return ht_fail(c, HT_ERR_INVALID, "ht_camshift_track_sequence: bad stream count, call count or frame stride");
    if (first < 0 || first + n > c->cs_streams) return ht_fail(c, HT_ERR_INVALID, "ht_camshift_track_sequence: stream range not reserved");
    for (int k = 0; k < ncalls; k++)
        if (!dev_frames[k] || ((uintptr_t)dev_frames[k] & 3)) return ht_fail(c, HT_ERR_INVALID, "ht_camshift_track_sequence: bad frame pointer");
    HT_HIP(c, hipSetDevice(c->device));
    const size_t need = (size_t)n * (size_t)(out_all ? ncalls : 1);
    if (c->cs_seq_cap < need) {
        HT_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_cs_seq_out) (void)hipFree(c->d_cs_seq_out);
        c->d_cs_seq_out = nullptr;
        c->cs_seq_cap = 0;
        if (hipMalloc(&c->d_cs_seq_out, need * sizeof(ht_cs_trackobj)) != hipSuccess) return ht_fail(c, HT_ERR_NOMEM, "ht_camshift_track_sequence: hipMalloc failed");
        c->cs_seq_cap = need;
    }
    // the calls of one stream are sequentially dependent (search window, camshift.js:257-258), the streams are not: 2 launches
    // per call on the context's stream, no host round trip in between
    if (n >= c->cs_fused_min_streams && c->cs_seq_fused) {
        // one launch per CS_SEQ_MAX calls: every workgroup walks its stream's calls on its own (see k_cs_track_fused)
        if (!c->cs_seq_attr_set) {
            HT_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_cs_track_fused<true, FUSED_NT>), hipFuncAttributeMaxDynamicSharedMemorySize, CS_REGION_CAP * 2));
            c->cs_seq_attr_set = true;
        }
        const uint32_t npix = (uint32_t)((size_t)c->W * c->H);
        const bool small = fused_threads(c, n) == FUSED_NT_SMALL;
        for (int k0 = 0; k0 < ncalls; k0 += CS_SEQ_MAX) {
            const int kc = std::min(CS_SEQ_MAX, ncalls - k0);
            CsFusedArgs ka;
            std::memset(&ka, 0, sizeof(ka));
            for (int k = 0; k < kc; k++) ka.flist.p[k] = static_cast<const uint8_t *>(dev_frames[k0 + k]);
            ka.ncalls = kc, ka.W = c->W, ka.H = c->H, ka.npix = npix, ka.frame_stride = frame_stride, ka.states = c->d_cs, ka.first = first;
            ka.calc_angles = calc_angles, ka.max_it = c->dbg_cs_iters, ka.region_cap = c->cs_region_cap;
            ka.out = c->d_cs_seq_out + (out_all ? (size_t)k0 * n : 0), ka.out_call_stride = out_all ? (uint32_t)n : 0u;
            ka.dbg_hist = c->cs_keep_hist ? c->d_cs_hist : nullptr;
            HtProfScope ps(c, small ? "cs_track_512" : "cs_track");
            c->cs_fused_launches[small ? 1 : 0]++;
            if (small) {
                ka.region_cap = std::min(ka.region_cap, CS_REGION_CAP_SMALL);
                hipLaunchKernelGGL((k_cs_track_fused<true, FUSED_NT_SMALL>), dim3(n), dim3(FUSED_NT_SMALL), (size_t)CS_REGION_CAP_SMALL * 2, c->stream, ka);
            } else {
                hipLaunchKernelGGL((k_cs_track_fused<true, FUSED_NT>), dim3(n), dim3(FUSED_NT), (size_t)CS_REGION_CAP * 2, c->stream, ka);
            }
            HT_HIP(c, hipGetLastError());
        }
        c->cs_last_first = first, c->cs_last_n = n, c->cs_last_chunks = c->cs_keep_hist ? 1 : 0;
        c->cs_last_hist = c->d_cs_hist;
    } else {
        for (int k = 0; k < ncalls; k++) {
            ht_cs_trackobj *d_out = c->d_cs_seq_out + (out_all ? (size_t)k * n : 0);
            ht_status st = launch_track(c, static_cast<const uint8_t *>(dev_frames[k]), frame_stride, first, n, calc_angles, d_out);
            if (st != HT_OK) return st;
        }
    }
    c->cs_seq_pending_n = 0;
    if (out) {
        HT_HIP(c, hipMemcpyAsync(out, c->d_cs_seq_out, need * sizeof(ht_cs_trackobj), hipMemcpyDeviceToHost, c->stream));
        HT_HIP(c, hipMemcpyAsync(c->h_cs_err, c->d_cs_err, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HT_HIP(c, hipStreamSynchronize(c->stream));
        return cs_check_err(c, "ht_camshift_track_sequence");
    }
    c->cs_seq_pending_n = n, c->cs_seq_pending_calls = ncalls, c->cs_seq_pending_all = out_all ? 1 : 0;  // what ht_camshift_sequence_collect may fetch
    return HT_OK;
}

extern "C" ht_status ht_camshift_sequence_collect(ht_ctx *c, int32_t n, int32_t ncalls, int32_t out_all, ht_cs_trackobj *out) {
    if (!c || !out || n <= 0 || ncalls <= 0) return HT_ERR_INVALID;
    const size_t need = (size_t)n * (size_t)(out_all ? ncalls : 1);
    // only the sequence that was enqueued with out == NULL, with the layout it was enqueued with: anything else would read stale or
    // differently strided track objects
    if (!c->d_cs_seq_out || c->cs_seq_cap < need || c->cs_seq_pending_n != n || c->cs_seq_pending_calls != ncalls || c->cs_seq_pending_all != (out_all ? 1 : 0))
        return ht_fail(c, HT_ERR_STATE, "ht_camshift_sequence_collect: no enqueue-only sequence with this n / ncalls / out_all is pending");
    HT_HIP(c, hipSetDevice(c->device));
    HT_HIP(c, hipMemcpyAsync(out, c->d_cs_seq_out, need * sizeof(ht_cs_trackobj), hipMemcpyDeviceToHost, c->stream));
    HT_HIP(c, hipMemcpyAsync(c->h_cs_err, c->d_cs_err, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HT_HIP(c, hipStreamSynchronize(c->stream));
    c->cs_seq_pending_n = 0;
    return cs_check_err(c, "ht_camshift_sequence_collect");
}

extern "C" ht_status ht_camshift_stats(ht_ctx *c, int32_t first, int32_t n, uint64_t *window_pixels, uint64_t *calls, int32_t reset) {
    if (!c || n <= 0 || first < 0 || first + n > c->cs_streams) return HT_ERR_INVALID;
    HT_HIP(c, hipSetDevice(c->device));
    HT_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<unsigned long long> v((size_t)n * 2);  // {win_px, calls} of every stream: one strided copy
    HT_HIP(c, hipMemcpy2D(v.data(), 16, &c->d_cs[first].win_px, sizeof(HtCsState), 16, (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        if (window_pixels) window_pixels[i] = v[2 * (size_t)i];
        if (calls) calls[i] = v[2 * (size_t)i + 1];
    }
    if (reset) HT_HIP(c, hipMemset2D(&c->d_cs[first].win_px, sizeof(HtCsState), 0, 16, (size_t)n));
    return HT_OK;
}

extern "C" ht_status ht_camshift_debug_hist(ht_ctx *c, int32_t stream, uint32_t *model, uint32_t *current) {
    if (!c || stream < 0 || stream >= c->cs_streams) return HT_ERR_INVALID;
    HT_HIP(c, hipSetDevice(c->device));
    HT_HIP(c, hipStreamSynchronize(c->stream));
    if (model) HT_HIP(c, hipMemcpy(model, c->d_cs[stream].model, sizeof(uint32_t) * 4096, hipMemcpyDeviceToHost));
    if (current) {
        if (stream < c->cs_last_first || stream >= c->cs_last_first + c->cs_last_n || c->cs_last_chunks <= 0 || !c->cs_last_hist)
            return ht_fail(c, HT_ERR_STATE, "ht_camshift_debug_hist: the stream was not part of the last track call");
        std::vector<uint32_t> part((size_t)c->cs_last_chunks * 4096);
        HT_HIP(c, hipMemcpy(part.data(), c->cs_last_hist + (size_t)(stream - c->cs_last_first) * c->cs_last_chunks * 4096, part.size() * sizeof(uint32_t),
                            hipMemcpyDeviceToHost));
        for (int b = 0; b < 4096; b++) {
            uint32_t v = 0;
            for (int k = 0; k < c->cs_last_chunks; k++) v += part[(size_t)k * 4096 + b];
            current[b] = v;
        }
    }
    return HT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// multi-GPU: in-place all-gather of fixed-size records over RCCL (xGMI), single-process form for the Node host.
// (bench.py / torch.distributed use one process per GPU and call RCCL through torch instead.)

namespace {
struct CommSet {
    std::vector<int> devs;
    std::vector<ncclComm_t> comms;
};
std::map<std::vector<int>, CommSet> g_comms;

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl &rccl() {
    static Rccl r;
    if (r.h) return r;
    r.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!r.h) r.h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!r.h) r.h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!r.h) return r;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.h, "ncclCommInitAll"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.h, "ncclGroupEnd"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.h, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.h, "ncclGetErrorString"));
    r.ok = r.CommInitAll && r.GroupStart && r.GroupEnd && r.AllGather && r.GetErrorString;
    return r;
}
}  // namespace

extern "C" ht_status ht_allgather_records(ht_ctx *const *ctxs, int32_t nranks, void *const *records_dev, size_t bytes_per_rank) {
    if (!ctxs || !records_dev || nranks <= 0 || bytes_per_rank == 0) return HT_ERR_INVALID;
    if (nranks == 1 && ctxs[0] && !ctxs[0]->force_rccl) return HT_OK;  // (option force_rccl runs RCCL with one rank: dlopen + ncclCommInitAll + ncclAllGather on a 1-GPU box)
    std::vector<int> devs(nranks);
    for (int i = 0; i < nranks; i++) {
        if (!ctxs[i] || !records_dev[i]) return HT_ERR_INVALID;
        devs[i] = ctxs[i]->device;
    }
    Rccl &R = rccl();
    if (!R.ok) return ht_fail(ctxs[0], HT_ERR_HIP, "ht_allgather_records: librccl.so could not be loaded");
    auto it = g_comms.find(devs);
    if (it == g_comms.end()) {
        CommSet cs;
        cs.devs = devs;
        cs.comms.resize(nranks);
        if (R.CommInitAll(cs.comms.data(), nranks, devs.data()) != ncclSuccess)
            return ht_fail(ctxs[0], HT_ERR_HIP, "ht_allgather_records: ncclCommInitAll failed");
        it = g_comms.emplace(devs, cs).first;
    }
    ncclResult_t r = R.GroupStart();
    for (int i = 0; i < nranks && r == ncclSuccess; i++) {
        char *buf = static_cast<char *>(records_dev[i]);
        r = R.AllGather(buf + (size_t)i * bytes_per_rank, buf, bytes_per_rank, ncclChar, it->second.comms[i], ctxs[i]->stream);
    }
    if (r == ncclSuccess) r = R.GroupEnd();
    if (r != ncclSuccess) return ht_fail(ctxs[0], HT_ERR_HIP, std::string("ht_allgather_records: ") + R.GetErrorString(r));
    for (int i = 0; i < nranks; i++) {
        HT_HIP(ctxs[i], hipSetDevice(ctxs[i]->device));
        HT_HIP(ctxs[i], hipStreamSynchronize(ctxs[i]->stream));
    }
    return HT_OK;
}

extern "C" int32_t ht_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// Single-process multi-GPU exchange of the per-frame bounding boxes: rank i's ht_best_faces output goes into slot i of a
// device buffer on ITS GPU, one ncclAllGather per rank over xGMI, then every rank's gathered table is read back and compared —
// all ranks must hold the same table — and the table is returned.
extern "C" ht_status ht_allgather_best_faces(ht_ctx *const *ctxs, int32_t nranks, const ht_rect *const *best, int32_t frames_per_rank, ht_rect *gathered) {
    if (!ctxs || !best || !gathered || nranks <= 0 || frames_per_rank <= 0) return HT_ERR_INVALID;
    const size_t per = sizeof(ht_rect) * (size_t)frames_per_rank, total = per * (size_t)nranks;
    std::vector<void *> bufs(nranks, nullptr);
    for (int i = 0; i < nranks; i++) {
        ht_ctx *c = ctxs[i];
        if (!c || !best[i]) return HT_ERR_INVALID;
        HT_HIP(c, hipSetDevice(c->device));
        if (c->d_gather_bytes < total) {
            HT_HIP(c, hipStreamSynchronize(c->stream));
            if (c->d_gather) (void)hipFree(c->d_gather);
            c->d_gather = nullptr;
            c->d_gather_bytes = 0;
            if (hipMalloc(&c->d_gather, total) != hipSuccess) return ht_fail(c, HT_ERR_NOMEM, "ht_allgather_best_faces: hipMalloc failed");
            c->d_gather_bytes = total;
        }
        HT_HIP(c, hipMemsetAsync(c->d_gather, 0, total, c->stream));
        HT_HIP(c, hipMemcpyAsync(static_cast<char *>(c->d_gather) + (size_t)i * per, best[i], per, hipMemcpyHostToDevice, c->stream));
        HT_HIP(c, hipStreamSynchronize(c->stream));  // best[i] is the caller's pageable memory
        bufs[i] = c->d_gather;
    }
    ht_status st = ht_allgather_records(ctxs, nranks, bufs.data(), per);
    if (st != HT_OK) return st;
    std::vector<char> other(total);
    for (int i = 0; i < nranks; i++) {
        ht_ctx *c = ctxs[i];
        HT_HIP(c, hipSetDevice(c->device));
        HT_HIP(c, hipMemcpy(i == 0 ? reinterpret_cast<char *>(gathered) : other.data(), c->d_gather, total, hipMemcpyDeviceToHost));
        if (i > 0 && std::memcmp(other.data(), gathered, total) != 0)
            return ht_fail(ctxs[0], HT_ERR_HIP, "ht_allgather_best_faces: rank " + std::to_string(i) + " holds a different table than rank 0 after the all-gather");
    }
    return HT_OK;
}
