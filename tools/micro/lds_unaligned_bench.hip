// tools/micro/lds_unaligned_bench.hip — does gfx950's LDS serve UNALIGNED ds_read_u16 / ds_read_b32 / ds_read_b64, are the bytes
// right, and at what rate?  (hipcc merges two adjacent ds_read_u8 into one ds_read_u16 at an arbitrary byte address for gfx950,
// so the hardware is expected to; the scan kernel could then fetch the same feature point of TWO neighbouring windows — bytes
// B+c and B+c+2 — with one ds_read_b32 at an odd address.)
// hipcc --offload-arch=gfx950 -O3 tools/micro/lds_unaligned_bench.hip -o /tmp/lds_unal && /tmp/lds_unal
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

// WIDTH: 1 = u8, 2 = u16, 4 = b32, 8 = b64.  Lane address = wave base + lane * STRIDE + MIS (+ k * 304 + rotating 16-byte step)
template <int WIDTH, int STRIDE, int MIS>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t *bad, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[40960];
    for (int i = threadIdx.x; i < 10240; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = (uint32_t)i * 2654435761u + 12345u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t base = wv * 1024 + lane * STRIDE + MIS;
    uint32_t acc = 0, wrong = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int kk = 0; kk < 32; kk++) {
            const uint32_t a = base + kk * 304 + (it & 7) * 16;
            uint32_t v = 0, v2 = 0;
            if (WIDTH == 1) asm volatile("ds_read_u8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            if (WIDTH == 2) asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            if (WIDTH == 4) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            if (WIDTH == 8) {
                uint64_t w;
                asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(a) : "memory");
                v = (uint32_t)w, v2 = (uint32_t)(w >> 32);
            }
            if (it == 0) {  // check the bytes once (not in the timed iterations' critical path: same code, result only compared at it == 0)
                uint32_t e = 0, e2 = 0;
                for (int b = 0; b < (WIDTH < 4 ? WIDTH : 4); b++) e |= (uint32_t)lds[a + b] << (8 * b);
                if (WIDTH == 8)
                    for (int b = 0; b < 4; b++) e2 |= (uint32_t)lds[a + 4 + b] << (8 * b);
                if (v != e || v2 != e2) wrong++;
            }
            acc += v + v2;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (wrong) atomicAdd(bad, wrong);
}

// throughput variant: 16 independent reads in flight per wait (what real code does)
template <int WIDTH, int STRIDE, int MIS>
__global__ __launch_bounds__(256) void kt(uint32_t *out, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[40960];
    for (int i = threadIdx.x; i < 10240; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = (uint32_t)i * 2654435761u + 12345u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t base = wv * 1024 + lane * STRIDE + MIS;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint32_t a = base + (it & 7) * 16;
        uint32_t v[16];
#define RD(i, op) asm volatile(op " %0, %1 offset:%2" : "=v"(v[i]) : "v"(a), "n"(i * 304) : "memory")
#define RD16(op) RD(0, op); RD(1, op); RD(2, op); RD(3, op); RD(4, op); RD(5, op); RD(6, op); RD(7, op); RD(8, op); RD(9, op); RD(10, op); RD(11, op); RD(12, op); RD(13, op); RD(14, op); RD(15, op)
        if (WIDTH == 1) { RD16("ds_read_u8"); }
        if (WIDTH == 2) { RD16("ds_read_u16"); }
        if (WIDTH == 4) { RD16("ds_read_b32"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i++) acc += v[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int WIDTH, int STRIDE, int MIS>
void run(const char *name) {
    uint32_t *d, *bad;
    hipMalloc(&d, 4096 * 256 * 4);
    hipMalloc(&bad, 4);
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL((k<WIDTH, STRIDE, MIS>), dim3(64), dim3(256), 0, 0, d, bad, 2);
    hipDeviceSynchronize();
    uint32_t hb = 0;
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    const int iters = 4000, blocks = 256 * 8;
    hipLaunchKernelGGL((kt<WIDTH, STRIDE, MIS>), dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((kt<WIDTH, STRIDE, MIS>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instrs = (double)blocks * 4 * iters * 16;
    printf("%-46s bytes %s  %8.3f ms  %.2f cycles per wave-read per CU at 2.4 GHz\n", name, hb ? "WRONG" : "ok   ", ms, ms * 1e-3 * 2.4e9 * 256 / wave_instrs);
    hipFree(d);
    hipFree(bad);
}

int main() {
    run<1, 2, 0>("ds_read_u8  stride 2 (today's scan reads)");
    run<1, 4, 1>("ds_read_u8  stride 4 +1");
    run<2, 2, 0>("ds_read_u16 stride 2 aligned");
    run<2, 2, 1>("ds_read_u16 stride 2 odd address");
    run<2, 4, 3>("ds_read_u16 stride 4 +3 (crosses a dword)");
    run<4, 4, 0>("ds_read_b32 stride 4 aligned");
    run<4, 4, 1>("ds_read_b32 stride 4 +1");
    run<4, 4, 2>("ds_read_b32 stride 4 +2");
    run<4, 4, 3>("ds_read_b32 stride 4 +3");
    run<4, 2, 0>("ds_read_b32 stride 2 (overlapping, half aligned)");
    run<4, 2, 1>("ds_read_b32 stride 2 +1 (overlapping, odd)");
    run<4, 8, 1>("ds_read_b32 stride 8 +1");
    run<8, 8, 0>("ds_read_b64 stride 8 aligned (latency form only)");
    run<8, 8, 3>("ds_read_b64 stride 8 +3 (latency form only)");
    return 0;
}
