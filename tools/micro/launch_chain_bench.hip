// tools/micro/launch_chain_bench.hip — what does a chain of N DEPENDENT kernels cost on this box, launched the way ht_detect_enqueue launches
// a small batch (one captured hipGraph per call, replayed, then a D2H of a few bytes + stream synchronisation)?
//   empty:  kernels that do nothing                      -> the per-boundary cost of the command processor (barrier bit, cache flush / invalidate)
//   touch:  64 workgroups that load 16 KB, wait, store   -> + one memory round trip per kernel (the floor of any kernel that consumes its predecessor's output)
// Output: wall us per replay + synchronise for N = 1, 2, 4, 8, 12 and the slope (us per additional kernel).
//   hipcc --offload-arch=gfx950 -O2 -o launch_chain_bench tools/micro/launch_chain_bench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_empty() {}
__global__ __launch_bounds__(256) void k_touch(const uint4 *__restrict__ src, uint4 *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint4 v = src[i];
    v.x += 1u;
    dst[i] = v;
}

static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint4 *a = nullptr, *b = nullptr;
    hipMalloc(&a, 64 * 256 * 16);
    hipMalloc(&b, 64 * 256 * 16);
    hipMemset(a, 0, 64 * 256 * 16);
    uint32_t *h = nullptr;
    hipHostMalloc(&h, 64, hipHostMallocDefault);
    const int Ns[] = {1, 2, 4, 8, 12};
    for (int mode = 0; mode < 2; mode++) {
        double t1 = 0, tl = 0;
        for (int N : Ns) {
            hipGraph_t g;
            hipGraphExec_t ge;
            hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            for (int k = 0; k < N; k++) {
                if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s);
                else hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, s, (k & 1) ? b : a, (k & 1) ? a : b);
            }
            hipStreamEndCapture(s, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            std::vector<double> us;
            for (int it = 0; it < 300; it++) {
                const auto t0 = std::chrono::steady_clock::now();
                hipGraphLaunch(ge, s);
                hipMemcpyAsync(h, a, 64, hipMemcpyDeviceToHost, s);
                hipStreamSynchronize(s);
                const auto t1c = std::chrono::steady_clock::now();
                if (it >= 50) us.push_back(std::chrono::duration<double, std::micro>(t1c - t0).count());
            }
            const double m = median(us);
            if (N == 1) t1 = m;
            tl = m;
            std::printf("%s chain of %2d kernels: %.1f us per replay + 64-byte D2H + synchronise\n", mode ? "touch" : "empty", N, m);
            hipGraphExecDestroy(ge);
            hipGraphDestroy(g);
        }
        std::printf("%s: %.2f us per additional dependent kernel (N = 1 -> 12)\n", mode ? "touch" : "empty", (tl - t1) / 11.0);
    }
    return 0;
}
