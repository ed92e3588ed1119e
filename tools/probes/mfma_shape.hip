// Probe (r04): does the MFMA SHAPE matter under the package power cap? The same FLOPs, the same LDS reads and the same VALU
// filler as v_mfma_f32_32x32x16_f16 (1 per step) or as v_mfma_f32_16x16x32_f16 (2 per step), operands that toggle, two waves
// per SIMD, each configuration long enough (~1.5 s) for the clock to settle. MI355X_MICROARCH.md, DVFS give-back (7): the
// 16x16x32 loop delivered 1.12-1.15x the FLOP/s of the 32x32x16 loop on random data.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_shape.hip -o tools/probes/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE, int V, int R>  // SHAPE 32 or 16; V = VALU (mul + max pairs) per 32768 FLOP; R = ds_read_b128 per 4 steps
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned seed) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[512 * 8 * 4];
    unsigned x = seed + threadIdx.x * 2654435761u;
    for (int i = threadIdx.x; i < 512 * 8 * 4; i += blockDim.x) {
        x = x * 1664525u + 1013904223u;
        lds[i] = __builtin_bit_cast(_Float16, (unsigned short)((x >> 16) & 0x3bff));  // finite, |v| < 2, random mantissas
    }
    __syncthreads();
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(threadIdx.x * 0.001f + i), b[i] = (_Float16)(i * 0.5f);
    f32x16 acc[4] = {};
    f32x4 acs[8] = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    const f16x8* lp = (const f16x8*)lds + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            f16x8 w[R ? R : 1];
#pragma unroll
            for (int q = 0; q < R; ++q) w[q] = lp[512 * ((g + q) & 3)];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f16x8 wa = R ? w[t % (R ? R : 1)] : a;
                if (SHAPE == 32) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, b, acc[t], 0, 0, 0);
                } else {
                    acs[2 * t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, b, acs[2 * t], 0, 0, 0);
                    acs[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, a, acs[2 * t + 1], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < V; ++j) v[(t + j) & 7] = fmaxf(v[(t + j) & 7] * 1.0001f, 0.5f);
                __builtin_amdgcn_sched_group_barrier(0x008, SHAPE == 32 ? 1 : 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2 * V, 0);
            }
        }
        {
            auto u = __builtin_bit_cast(uint4, a);
            u.x = u.x * 1664525u + 1013904223u, u.y ^= u.x >> 3, u.z += u.y, u.w ^= u.z;  // (operands that toggle: realistic power)
            u.x &= 0x3bff3bffu, u.y &= 0x3bff3bffu, u.z &= 0x3bff3bffu, u.w &= 0x3bff3bffu;
            a = __builtin_bit_cast(f16x8, u);
            auto q = __builtin_bit_cast(uint4, b);
            q.x ^= u.y, q.y += u.z, q.z ^= u.w, q.w += u.x;
            q.x &= 0x3bff3bffu, q.y &= 0x3bff3bffu, q.z &= 0x3bff3bffu, q.w &= 0x3bff3bffu;
            b = __builtin_bit_cast(f16x8, q);
        }
    }
    float s = 0;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    for (int t = 0; t < 8; ++t)
        for (int i = 0; i < 4; ++i) s += acs[t][i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE, int V, int R>
void run(int threads, double seconds) {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    int iters = 20000;
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {  // pass 0 calibrates the length, pass 1 is the measurement
        (void)hipEventRecord(e0);
        k<SHAPE, V, R><<<256, threads>>>(d, iters, 12345u);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (pass == 0) iters = (int)(iters * seconds * 1e3 / ms);
    }
    const double steps = (double)iters * 12 * (threads / 64) * 256;
    printf("shape %dx%dx%d  VALU pairs per step %d  ds_read_b128 per 4 steps %d  %d waves/SIMD: %.3f PFLOP/s over %.2f s\n", SHAPE, SHAPE, SHAPE == 32 ? 16 : 32, V, R,
           threads / 256, steps * 32768 / (ms * 1e-3) / 1e15, ms * 1e-3);
    (void)hipFree(d);
}
int main() {
    const double T = 1.5;
    for (int rep = 0; rep < 2; ++rep) {
        run<32, 0, 0>(512, T), run<16, 0, 0>(512, T);
        run<32, 3, 3>(512, T), run<16, 3, 3>(512, T);
        run<32, 3, 3>(256, T), run<16, 3, 3>(256, T);
    }
    return 0;
}
