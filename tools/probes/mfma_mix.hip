// Probe: MFMA rate of v_mfma_f32_32x32x16_f16 with V independent VALU instructions and R ds_read_b128 per MFMA in the same
// wave (the instruction mix of the fsplit heads: ~3 VALU and 0.8 LDS reads per MFMA), one and two waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_mix.hip -o tools/probes/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V, int R4>  // V VALU per MFMA; R4 = LDS reads per 4 MFMAs
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[512 * 8 * 4];
    for (int i = threadIdx.x; i < 512 * 8 * 4; i += blockDim.x) lds[i] = (_Float16)(i & 7);
    __syncthreads();
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(threadIdx.x * 0.001f + i), b[i] = (_Float16)(i * 0.5f);
    f32x16 acc[4] = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    const f16x8* lp = (const f16x8*)lds + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            f16x8 w[R4 ? R4 : 1];
#pragma unroll
            for (int q = 0; q < R4; ++q) w[q] = lp[512 * ((g + q) & 3)];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R4 ? w[t % (R4 ? R4 : 1)] : a, b, acc[t], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < V; ++j) v[(t + j) & 7] = fmaxf(v[(t + j) & 7] * 1.0001f, 0.5f);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2 * V, 0);
            }
        }
        {
            auto u = __builtin_bit_cast(uint4, a);
            u.x = u.x * 1664525u + 1013904223u, u.y ^= u.x >> 3, u.z += u.y, u.w ^= u.z;  // (operands that toggle: realistic power)
            u.x &= 0x3bff3bffu, u.y &= 0x3bff3bffu, u.z &= 0x3bff3bffu, u.w &= 0x3bff3bffu;  // finite fp16, |x| < 2
            a = __builtin_bit_cast(f16x8, u);
        }
    }
    float s = 0;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V, int R4>
void run(int threads) {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    const int iters = getenv("MFMA_ITERS") ? atoi(getenv("MFMA_ITERS")) : 10000;  // 2000000: ~1-2 s per line (power sampling)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    k<V, R4><<<256, threads>>>(d, 100);
    (void)hipEventRecord(e0);
    k<V, R4><<<256, threads>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 12 * (threads / 64) * 256;
    printf("VALU (mul+max pairs) per MFMA %d x2, ds_read_b128 per 4 MFMA %d, %d waves/SIMD: %.2f PFLOP/s (%.0f %% of 2.5)\n", V, R4, threads / 256,
           mf * 32768 / (ms * 1e-3) / 1e15, mf * 32768 / (ms * 1e-3) / 2.5e13);
    (void)hipFree(d);
}
int main(int argc, char** argv) {
    if (argc >= 4) {  // one configuration: mfma_mix V R4 threads  (MFMA_ITERS for the length; tools/power_mix.sh samples rocm-smi)
        const int V = atoi(argv[1]), R4 = atoi(argv[2]), th = atoi(argv[3]);
        if (V == 0 && R4 == 0) run<0, 0>(th);
        else if (V == 3 && R4 == 0) run<3, 0>(th);
        else if (V == 0 && R4 == 3) run<0, 3>(th);
        else run<3, 3>(th);
        return 0;
    }
    for (int th : {256, 512}) {
        run<0, 0>(th), run<1, 0>(th), run<2, 0>(th), run<3, 0>(th), run<0, 3>(th), run<1, 3>(th), run<2, 3>(th), run<3, 3>(th);
    }
    return 0;
}
