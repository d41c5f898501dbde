// tools/micro/lds_read_bench.hip — LDS read throughput by access width (gfx950): does ds_read_u8 issue at the ds_read_b32 rate?
// hipcc --offload-arch=gfx950 -O3 tools/micro/lds_read_bench.hip -o /tmp/lds_bench && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>  // 0: u8 stride 2 B, 1: u16 stride 2 B, 2: b32 stride 4 B, 3: u8 stride 4 B, 4: u8 stride 1 B, 5: b32 stride 2B-pairs (lane>>1)
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[32768];
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t base;
    if (MODE == 0 || MODE == 1) base = wv * 4096 + lane * 2;
    else if (MODE == 2 || MODE == 3) base = wv * 4096 + lane * 4;
    else if (MODE == 4) base = wv * 4096 + lane;
    else base = wv * 4096 + (lane >> 1) * 4;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const uint32_t a = base + k * 304 + (it & 7) * 16;
            if (MODE == 0 || MODE == 3 || MODE == 4) acc += lds[a];
            else if (MODE == 1) acc += *reinterpret_cast<const uint16_t *>(&lds[a & ~1u]);
            else acc += *reinterpret_cast<const uint32_t *>(&lds[a & ~3u]);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
void run(const char *name) {
    uint32_t *d;
    hipMalloc(&d, 4096 * 256 * 4);
    const int iters = 2000, blocks = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instrs = (double)blocks * 4 * iters * 32;
    printf("%-34s %8.3f ms  %.2f G wave-reads/s  -> %.2f cycles per wave-read per CU at 2.4 GHz (256 CUs)\n", name, ms, wave_instrs / ms / 1e6,
           ms * 1e-3 * 2.4e9 * 256 / wave_instrs);
    hipFree(d);
}

int main() {
    run<2>("ds_read_b32 stride 4 B");
    run<0>("ds_read_u8  stride 2 B (our case)");
    run<1>("ds_read_u16 stride 2 B");
    run<3>("ds_read_u8  stride 4 B");
    run<4>("ds_read_u8  stride 1 B");
    run<5>("ds_read_b32 lane>>1 (dword holding the stride-2 byte)");
    return 0;
}
