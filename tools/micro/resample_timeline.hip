// tools/micro/resample_timeline.hip — where does a k_resample workgroup spend its life?  Compiles the product kernel with
// shader-clock stamps (HT_RS_TIMELINE) and runs generation 1 of the C2 pyramid (levels 1..6 from 256 gray 320x240 planes).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I headtrackr_amd/csrc tools/micro/resample_timeline.hip -o tools/micro/resample_timeline
#define HT_RS_TIMELINE 1
#include "../../headtrackr_amd/csrc/ht_pyramid.hip"
#include <algorithm>
#include <cstdio>
#include <vector>

// host-side symbols of the library that the included launchers reference (unused here)
HtProfScope::HtProfScope(ht_ctx *c, const char *) : ctx(c) {}
HtProfScope::~HtProfScope() {}
ht_status ht_fail(ht_ctx *, ht_status st, const std::string &) { return st; }

int main(int argc, char **argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 320, H = argc > 2 ? atoi(argv[2]) : 240, N = argc > 3 ? atoi(argv[3]) : 256;
    const int npmax_cap = argc > 4 ? atoi(argv[4]) : HT_RS_MAX_PASSES;
    const uint32_t K = argc > 5 ? atoi(argv[5]) : 1;
    std::vector<HtResampleJob> tiles;
    size_t off = 0;
    auto plane = [&](int w, int h, int *stride) { *stride = (w + 3) & ~3; size_t o = off; off = (off + (size_t)*stride * h + 255) & ~(size_t)255; return o; };
    int s0; const size_t off0 = plane(W, H, &s0);
    size_t px = 0;
    for (int i = 1; i <= 6; i++) {
        const double sc = pow(2.0, i / 6.0);
        const int w = (int)floor(W / sc), h = (int)floor(H / sc);
        int st; const size_t o = plane(w, h, &st);
        HtResampleJob j{};
        j.src_off = (uint32_t)off0, j.dst_off = (uint32_t)o, j.src_stride = s0, j.dst_stride = st;
        j.sx = j.sy = 0, j.sw = W, j.sh = H, j.dw = j.cw = w, j.dh = j.ch = h;
        j.rx = (double)W / w, j.ry = (double)H / h;
        int npmax = 1;
        for (int t = 2; t <= npmax_cap; t++) if ((int)ceil(16.0 * t * j.ry) + 3 <= HT_RS_SRC_ROWS) npmax = t;
        const int passes = (h + 15) / 16, nby = (passes + npmax - 1) / npmax, nbx = (w + 63) / 64;
        int pass0 = 0;
        for (int y = 0; y < nby; y++) {
            const int np = passes / nby + (y < passes % nby ? 1 : 0);
            for (int x = 0; x < nbx; x++) { HtResampleJob t = j; t.bx = x, t.pass0 = pass0, t.np = np; tiles.push_back(t); }
            pass0 += np;
        }
        px += (size_t)w * h;
    }
    const size_t arena_stride = (off + 511) & ~(size_t)255;
    uint8_t *arena; HtResampleJob *d_tiles;
    hipMalloc(&arena, arena_stride * N + 4096); hipMemset(arena, 7, arena_stride * N + 4096);
    hipMalloc(&d_tiles, tiles.size() * sizeof(HtResampleJob));
    hipMemcpy(d_tiles, tiles.data(), tiles.size() * sizeof(HtResampleJob), hipMemcpyHostToDevice);
    const uint32_t bpf = (uint32_t)tiles.size();
    const uint32_t ngroups = (N + K - 1) / K;
    const dim3 grid((bpf * ngroups + 7u) & ~7u);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_resample<HT_RS_MAX_PASSES>, grid, dim3(256), 0, 0, d_tiles, arena, arena_stride, bpf, ngroups, (uint32_t)N, K);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = std::min(best, ms);
    }
    printf("%dx%d x%d: %zu tiles/frame, grid %u, %.1f us, %.1f Gpx/s\n", W, H, N, tiles.size(), grid.x, best * 1e3, px * N / best / 1e6);
    static unsigned long long tl[1 << 16][8];
    hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_rs_timeline), sizeof(tl));
    const int nb = std::min<int>(grid.x, 1 << 16);
    const char *names[6] = {"prologue (record, extent, loads, taps)", "taps -> registers / earlier frames", "LAST FRAME: wait loads + LDS write", "  barrier", "  issue next loads + pixels", "  stores + barrier"};
    unsigned long long tmin = ~0ull, tmax = 0;
    double sum[6] = {0}; int cnt = 0;
    for (int i = 0; i < nb; i++) {
        if (!tl[i][0] || !tl[i][6] || !tl[i][5]) continue;
        tmin = std::min(tmin, tl[i][0]); tmax = std::max(tmax, tl[i][6]);
        for (int k = 0; k < 6; k++) sum[k] += (double)(tl[i][k + 1] - tl[i][k]);
        cnt++;
    }
    printf("stamped workgroups %d, span %.1f us at 100 MHz-equivalent? (raw ticks %llu)\n", cnt, 0.0, tmax - tmin);
    double tot = 0; for (int k = 0; k < 6; k++) tot += sum[k] / cnt;
    for (int k = 0; k < 6; k++) printf("  %-40s %9.0f ticks  %5.1f %%\n", names[k], sum[k] / cnt, 100.0 * sum[k] / cnt / tot);
    printf("  %-40s %9.0f ticks per workgroup; kernel span %llu ticks => tick = %.3f ns\n", "total", tot, tmax - tmin, best * 1e6 / (double)(tmax - tmin));
    return 0;
}
