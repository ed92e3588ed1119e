// Probe: cost of back-to-back DEPENDENT v_mfma_f32_32x32x16_f16 (same accumulator) against round-robin over 4 accumulators,
// with one and two waves per SIMD. Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_dep.hip -o tools/probes/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>  // 0: chain of 3 on one accumulator then next (t-major), 1: round robin over 4, 2: single accumulator
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(threadIdx.x * 0.001f + i), b[i] = (_Float16)(i * 0.5f);
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 12; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
        }
        asm volatile("" : "+v"(a));
    }
    float s = 0;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int threads) {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    const int iters = getenv("MFMA_ITERS") ? atoi(getenv("MFMA_ITERS")) : 20000;  // 20000: a 4-7 ms burst; 5000000: ~1.5 s under the power cap
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(d, 100);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 12 * (threads / 64) * 256;
    printf("%-34s %d waves/SIMD: %.3f ms, %.1f ns per MFMA per SIMD, %.2f PFLOP/s\n", name, threads / 256, ms, ms * 1e6 / ((double)iters * 12 * (threads / 256)),
           mf * 32768 / (ms * 1e-3) / 1e15);
    hipFree(d);
}
int main() {
    for (int th : {256, 512}) {
        run<0>("3 dependent, then next accumulator", th);
        run<1>("round robin over 4 accumulators", th);
        run<2>("one accumulator only", th);
    }
    return 0;
}
