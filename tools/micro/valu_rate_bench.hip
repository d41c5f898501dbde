// tools/micro/valu_rate_bench.hip — issue rate of the VALU instructions the detect kernels are made of (gfx950): cycles per wave64
// instruction per SIMD, from the shader clock (s_memtime) and from the wall clock, at 1 / 2 / 4 / 8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate_bench.hip -o /tmp/valu_bench && /tmp/valu_bench
// Is a plain VALU instruction 4 cycles per wave64 (16 lanes per SIMD and clock) or 2 (32 lanes)?  DESIGN.md's pipe-utilisation figures
// depend on it.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *cyc, int iters) {
    uint32_t a[16];
    double da[8];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 2654435761u + i * 40503u;
#pragma unroll
    for (int i = 0; i < 8; i++) da[i] = 1.0 + 1e-9 * (threadIdx.x + i);
    uint32_t b = threadIdx.x | 0x01010101u, c = 0x07060100u + blockIdx.x % 3;
    float fb = 1.0000001f, fc = 1e-7f * threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#define OP0(i) asm volatile("v_min_u32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define OP1(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP2(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP3(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(fb), "v"(fc));
#define OP4(i) asm volatile("v_cvt_f32_ubyte0_e32 %0, %0" : "+v"(a[i]));
#define OP5(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP6(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP7(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP8(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(fb));
#define OP9(i) asm volatile("v_rndne_f32_e32 %0, %0" : "+v"(a[i]));
#define OP10(i) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 0) { REP16(OP0) REP16(OP0) }
        if (OP == 1) { REP16(OP1) REP16(OP1) }
        if (OP == 2) { REP16(OP2) REP16(OP2) }
        if (OP == 3) { REP16(OP3) REP16(OP3) }
        if (OP == 4) { REP16(OP4) REP16(OP4) }
        if (OP == 5) { REP16(OP5) REP16(OP5) }
        if (OP == 6) { REP16(OP6) REP16(OP6) }
        if (OP == 7) { REP16(OP7) REP16(OP7) }
        if (OP == 8) { REP16(OP8) REP16(OP8) }
        if (OP == 9) { REP16(OP9) REP16(OP9) }
        if (OP == 10) { REP16(OP10) REP16(OP10) }
        if (OP == 11) {  // v_pk_fma_f32 on 8 register pairs, 4 times = 32 instructions
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(da[i]) : "v"(da[(i + 1) & 7]));
        }
        if (OP == 12) {  // v_fma_f64
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(da[i]) : "v"(da[(i + 1) & 7]));
        }
        if (OP == 13) {  // v_mul_f64
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(da[i]) : "v"(da[(i + 1) & 7]));
        }
        if (OP == 14) {  // v_add_f64
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(da[i]) : "v"(da[(i + 1) & 7]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += (uint32_t)__double2loint(da[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name) {
    uint32_t *d;
    unsigned long long *dc;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    hipMalloc(&dc, 256 * 8 * 8);
    const int iters = 4000;
    printf("%-22s", name);
    for (int w = 1; w <= 8; w *= 2) {
        const int blocks = 256 * w;  // 4 waves per block = one per SIMD of a CU; w blocks per CU
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, dc, 10);
        hipDeviceSynchronize();
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, dc, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        unsigned long long hc[2048];
        hipMemcpy(hc, dc, blocks * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < blocks; i++) mean += (double)hc[i];
        mean /= blocks;
        const double per_simd_instrs = (double)iters * 32 * w;  // wave-instructions one SIMD issues
        printf("  w=%d: %6.2f clk (s_memtime) %6.2f clk@2.4GHz (wall)", w, mean / per_simd_instrs, ms * 1e-3 * 2.4e9 / per_simd_instrs);
    }
    printf("\n");
    hipFree(d);
    hipFree(dc);
}

int main() {
    printf("cycles per wave64 instruction per SIMD (w waves per SIMD, 32 independent instructions per loop iteration)\n");
    run<0>("v_min_u32");
    run<5>("v_min3_u32");
    run<1>("v_perm_b32");
    run<2>("v_pk_min_u16");
    run<6>("v_mad_u32_u24");
    run<7>("v_mul_lo_u32");
    run<10>("v_sad_u8");
    run<3>("v_fma_f32");
    run<11>("v_pk_fma_f32");
    run<4>("v_cvt_f32_ubyte0");
    run<8>("v_cvt_pk_u8_f32");
    run<9>("v_rndne_f32");
    run<12>("v_fma_f64");
    run<13>("v_mul_f64");
    run<14>("v_add_f64");
    return 0;
}
