// tools/micro/dadd_latency_bench.hip — latency of a DEPENDENT chain of v_add_f64 (the sequential binary64 confidence sum of the cascade's
// last stage is exactly that: 564 adds, each waiting for the one before), alone on a SIMD and with other chains beside it.
// hipcc --offload-arch=gfx950 -O3 tools/micro/dadd_latency_bench.hip -o /tmp/dadd && /tmp/dadd
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>  // 0: v_add_f64 chain (VGPR operands), 1: v_add_f32 chain, 2: v_add_f64 chain whose addend comes from v_readlane (constant lane)
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *cyc, int iters) {
    double s = threadIdx.x * 1e-9, a = 1.0 + threadIdx.x * 1e-12;
    float fs = threadIdx.x * 1e-9f, fa = 1.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(s) : "v"(a));
            if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(fs) : "v"(fa));
            if (MODE == 2) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(a), u), hi = __builtin_amdgcn_readlane(__double2hiint(a), u);
                s = __dadd_rn(s, __hiloint2double(hi, lo));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = s + fs;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char *name) {
    double *d;
    unsigned long long *dc;
    (void)hipMalloc(&d, 2048 * 256 * 8);
    (void)hipMalloc(&dc, 2048 * 8);
    const int iters = 2000;
    printf("%-44s", name);
    for (int w = 1; w <= 8; w *= 2) {
        const int blocks = 256 * w;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, 10);
        (void)hipDeviceSynchronize();
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        printf("  w=%d: %6.1f clk@2.4GHz per dependent add", w, ms * 1e-3 * 2.4e9 / ((double)iters * 32));
    }
    printf("\n");
}
int main() {
    printf("latency of a dependent chain, per add (wall clock, w waves per SIMD each running its own chain)\n");
    run<0>("v_add_f64, VGPR operands");
    run<1>("v_add_f32, VGPR operands");
    run<2>("v_add_f64, addend from 2 x v_readlane (imm lane)");
    return 0;
}
