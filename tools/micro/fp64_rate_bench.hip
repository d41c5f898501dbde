// tools/micro/fp64_rate_bench.hip — issue rate of the binary64 instructions the pyramid resampler uses (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/micro/fp64_rate_bench.hip -o /tmp/fp64_bench && /tmp/fp64_bench
// Each kernel runs 8 independent chains of one instruction per lane, 4 waves per SIMD, every SIMD busy: the printed
// figure is SIMD cycles per wave64 instruction (4 = full rate, 16 = quarter rate) assuming 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, uint32_t seed) {
    double d[8];
    uint32_t u[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u[i] = (threadIdx.x * 8 + i + seed) & 255u;
        d[i] = 1.0 + 1e-9 * (double)(threadIdx.x + i);
        f[i] = (float)u[i];
    }
    const double c = 1.0000000001, m = 4503599627370496.0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (MODE == 0) {
#define X(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
                REP8(X)
#undef X
            } else if (MODE == 1) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (MODE == 2) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (MODE == 3) {
#define X(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
                REP8(X)
#undef X
            } else if (MODE == 4) {
#define X(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i]));
                REP8(X)
#undef X
            } else if (MODE == 5) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
                REP8(X)
#undef X
            } else if (MODE == 6) {
#define X(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[i]) : "v"(u[i]));
                REP8(X)
#undef X
            } else if (MODE == 7) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (MODE == 8) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(1.0001f));
                REP8(X)
#undef X
            } else if (MODE == 9) {  // magic-number u32 -> f64: v_mov hi + v_add_f64
#define X(i) { d[i] = __hiloint2double(0x43300000, (int)u[i]) - m; asm volatile("" : "+v"(d[i])); u[i] = (u[i] + 1) & 255u; }
                REP8(X)
#undef X
            } else if (MODE == 10) {
#define X(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
                REP8(X)
#undef X
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += d[i] + (double)u[i] + (double)f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int per_rep) {
    double *d;
    hipMalloc(&d, 4096 * 256 * 8);
    const int iters = 4000, blocks = 256 * 4;  // 4 workgroups per CU = 4 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instrs = (double)blocks * 4 * iters * 4 * 8 * per_rep;  // per launch
    printf("%-42s %8.3f ms  -> %.2f SIMD cycles per wave instruction (2.4 GHz, 1024 SIMDs)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / wave_instrs);
    hipFree(d);
}

int main() {
    run<8>("v_mul_f32 (calibration: expect 4)", 1);
    run<1>("v_mul_f64", 1);
    run<2>("v_add_f64", 1);
    run<7>("v_fma_f64", 1);
    run<0>("v_cvt_f64_u32", 1);
    run<10>("v_cvt_f64_i32", 1);
    run<5>("v_cvt_f64_f32", 1);
    run<6>("v_cvt_f32_ubyte1", 1);
    run<3>("v_rndne_f64", 1);
    run<4>("v_cvt_i32_f64", 1);
    run<9>("magic u32->f64 (hi:lo pair - 2^52) + 2 int ops", 1);
    return 0;
}
