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
// aligned) inside a per-frame arena; frames are arena_stride bytes apart.  All kernels take the frame index from
// blockIdx.y so a whole batch is one launch per dependency generation.
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
template <bool GRAY_IN_R>
__global__ __launch_bounds__(256) void k_gray_linear(const uint8_t *__restrict__ frames, size_t frame_stride,
                                                     uint8_t *__restrict__ arena, uint64_t arena_stride, uint32_t off0,
                                                     uint32_t ngroups, uint32_t blocks_per_frame, uint32_t nframes) {
    uint32_t f, blk;
    if (!xcd_item(blocks_per_frame, nframes, &f, &blk)) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(frames + (size_t)f * frame_stride);
    uint32_t *dst = reinterpret_cast<uint32_t *>(arena + (uint64_t)f * arena_stride + off0);
    for (uint32_t g = blk * blockDim.x + threadIdx.x; g < ngroups; g += blocks_per_frame * blockDim.x) {
        const uint4 p = src[g];
        uint32_t o;
        if (GRAY_IN_R)
            o = (p.x & 0xff) | ((p.y & 0xff) << 8) | ((p.z & 0xff) << 16) | ((p.w & 0xff) << 24);
        else
            o = gray_of(p.x) | (gray_of(p.y) << 8) | (gray_of(p.z) << 16) | (gray_of(p.w) << 24);
        dst[g] = o;
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
constexpr int RS_SP = 160;                  // LDS source tile: bytes per row (64 * 2.04 + 2, dword aligned start)
// A workgroup's latency chain (taps -> barrier -> HBM loads -> barrier -> LDS reads -> store) is fixed, so the tile
// height decides how much of it is amortised: with 16 rows (1 row per thread) the 260k waves of one C2 generation run
// in ~32 occupancy rounds of ~4 us each.

struct RsTap {
    double t, u;  // weights of b and a
    int a, b;     // source coordinates (absolute, including the source rect origin)
};

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

#ifndef HT_RS_WPS
#define HT_RS_WPS 1
#endif
template <int RPT>
__global__ __launch_bounds__(256, HT_RS_WPS) void k_resample(const HtResampleJob *__restrict__ jobs, const HtBlockRef *__restrict__ refs,
                                                  uint8_t *__restrict__ arena, uint64_t arena_stride, uint32_t blocks_per_frame,
                                                  uint32_t nframes) {
    constexpr int TH = 16 * RPT;               // destination rows per tile
    constexpr int SR = 2 * TH + TH / 16 + 6;   // LDS source rows (ratio <= 2.04 plus the tap pair)
    __shared__ __attribute__((aligned(16))) uint8_t s_src[SR * RS_SP];
    __shared__ RsTap s_col[RS_TW], s_row[TH];
    uint32_t fidx, blk;
    if (!xcd_item(blocks_per_frame, nframes, &fidx, &blk)) return;
    const HtBlockRef ref = refs[blk];  // block -> (job, tile x, tile y): one scalar load
    const HtResampleJob &J = jobs[ref.item];
    const int tid = (int)threadIdx.x;
    const int X0 = (int)ref.bx * RS_TW, Y0 = (int)ref.by * TH;
    uint8_t *frame = arena + (uint64_t)fidx * arena_stride;
    const uint8_t *src = frame + J.src_off;
    const int ncols = min(RS_TW, J.dw - X0), nrows = min(TH, J.dh - Y0);  // drawn part of this tile (may be <= 0)
    const int x0 = X0 + (tid & 15) * 4, yt = Y0 + (tid >> 4);              // this thread: rows yt, yt+16, ...
    uint32_t o[RPT];
#pragma unroll
    for (int q = 0; q < RPT; q++) o[q] = 0;
    if (ncols > 0 && nrows > 0) {
        if (tid < ncols) s_col[tid] = rs_tap(X0 + tid, J.rx, J.sw, J.sx);
        if (tid >= 64 && tid - 64 < nrows) s_row[tid - 64] = rs_tap(Y0 + tid - 64, J.ry, J.sh, J.sy);
        __syncthreads();
        const int xa = s_col[0].a & ~3, xb = s_col[ncols - 1].b, ya = s_row[0].a, yb = s_row[nrows - 1].b;
        const int sw4 = (xb - xa) / 4 + 1, sh = yb - ya + 1;  // dwords per row, rows
        const bool in_lds = (sw4 * 4 <= RS_SP) && (sh <= SR);
        const int npx = min(4, J.dw - x0);
        if (in_lds) {
            // source rows as aligned dwords, all loads issued before the first LDS write.  8 threads share a row (5
            // consecutive dwords each = the 160-byte LDS pitch), 32 rows per pass: the index arithmetic is one add per
            // pass instead of a division per dword (the staging loop used to be as long as the pixel arithmetic itself).
            constexpr int KR = (SR + 31) / 32;
            const int r0 = tid >> 3, cg = (tid & 7) * 5;
            const uint8_t *sbase = src + (size_t)ya * J.src_stride + xa + 4 * cg;
            uint32_t v[KR][5];
#pragma unroll
            for (int k = 0; k < KR; k++) {
                const int r = r0 + 32 * k;
                const uint32_t *rowp = reinterpret_cast<const uint32_t *>(sbase + (size_t)r * J.src_stride);
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    v[k][j] = 0;
                    if (r < sh && cg + j < sw4) v[k][j] = rowp[j];
                }
            }
#pragma unroll
            for (int k = 0; k < KR; k++) {
                const int r = r0 + 32 * k;
                if (r < SR) {
                    uint32_t *dstp = reinterpret_cast<uint32_t *>(&s_src[r * RS_SP + 4 * cg]);
#pragma unroll
                    for (int j = 0; j < 5; j++) dstp[j] = v[k][j];
                }
            }
            __syncthreads();
            RsTap cx[4];
#pragma unroll
            for (int k = 0; k < 4; k++) cx[k] = s_col[min(x0 - X0 + k, ncols - 1)];
#pragma unroll
            for (int q = 0; q < RPT; q++) {
                const int y = yt + 16 * q;
                if (y < J.dh && npx > 0) {
                    const RsTap ry = s_row[y - Y0];
                    o[q] = rs_pixels4<const uint8_t *>(s_src + (ry.a - ya) * RS_SP, s_src + (ry.b - ya) * RS_SP, cx, ry, xa, npx);
                }
            }
        } else {
            RsTap cx[4];
#pragma unroll
            for (int k = 0; k < 4; k++) cx[k] = s_col[min(x0 - X0 + k, ncols - 1)];
#pragma unroll
            for (int q = 0; q < RPT; q++) {
                const int y = yt + 16 * q;
                if (y < J.dh && npx > 0) {
                    const RsTap ry = s_row[y - Y0];
                    o[q] = rs_pixels4<const uint8_t *>(src + (size_t)ry.a * J.src_stride, src + (size_t)ry.b * J.src_stride, cx, ry, 0, npx);
                }
            }
        }
    }
    // pixels outside the drawn dw x dh rect stay transparent black (ccv.js:135-145 draws 2 px short on the variants)
#pragma unroll
    for (int q = 0; q < RPT; q++) {
        const int y = yt + 16 * q;
        if (y < J.ch && x0 < J.dst_stride) *reinterpret_cast<uint32_t *>(frame + J.dst_off + (size_t)y * J.dst_stride + x0) = o[q];
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
            if (gray_in_r)
                hipLaunchKernelGGL(k_gray_linear<true>, grid, dim3(256), 0, c->stream, c->d_frames, c->frame_stride, c->d_arena,
                                   c->arena_stride, L0.off[0], ngroups, bpf, (uint32_t)c->nframes);
            else
                hipLaunchKernelGGL(k_gray_linear<false>, grid, dim3(256), 0, c->stream, c->d_frames, c->frame_stride, c->d_arena,
                                   c->arena_stride, L0.off[0], ngroups, bpf, (uint32_t)c->nframes);
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
    for (size_t g = 1; g < c->h_gens.size(); g++) {
        if (c->gen_blocks[g] == 0) continue;
        HtProfScope ps(c, "resample");
        const dim3 rgrid((c->gen_blocks[g] * (uint32_t)c->nframes + 7u) & ~7u);
        if (c->gen_rpt[g] == 4)
            hipLaunchKernelGGL(k_resample<4>, rgrid, dim3(256), 0, c->stream, c->d_gens[g], c->d_gen_blocks[g], c->d_arena,
                               c->arena_stride, c->gen_blocks[g], (uint32_t)c->nframes);
        else if (c->gen_rpt[g] == 2)
            hipLaunchKernelGGL(k_resample<2>, rgrid, dim3(256), 0, c->stream, c->d_gens[g], c->d_gen_blocks[g], c->d_arena,
                               c->arena_stride, c->gen_blocks[g], (uint32_t)c->nframes);
        else
            hipLaunchKernelGGL(k_resample<1>, rgrid, dim3(256), 0, c->stream, c->d_gens[g], c->d_gen_blocks[g], c->d_arena,
                               c->arena_stride, c->gen_blocks[g], (uint32_t)c->nframes);
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

ht_status ht_launch_whitebalance(ht_ctx *c, double *d_out) {
    HtProfScope ps(c, "whitebalance");
    unsigned long long *out = reinterpret_cast<unsigned long long *>(d_out);
    HT_HIP(c, hipMemsetAsync(out, 0, sizeof(unsigned long long) * 4 * (size_t)c->nframes, c->stream));
    const uint32_t npix = (uint32_t)((size_t)c->W * c->H);
    hipLaunchKernelGGL(k_channel_sums, dim3(std::min<uint32_t>((npix + 1023) / 1024, 256), c->nframes), dim3(256), 0, c->stream,
                       c->d_frames, c->frame_stride, npix, out);
    HT_HIP(c, hipGetLastError());
    return HT_OK;
}
