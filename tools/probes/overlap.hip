// Probe (not product): does a second co-resident workgroup per CU hide the fused learn kernel's update stream?
// Each "tile" models one agent of learn_kernel_t<fused>: a compute phase (f32 MFMAs at ~50 % pipe duty, registers only)
// followed by a stream phase of 9 blocks [issue Adam operand loads | 160 MFMAs | update | stores] over the agent's
// 9 x 64 x 128 floats of (w, w_target, m, v).  LDS per workgroup decides how many workgroups share a CU.
//   overlap <lds_kb> <f4:0|1> <tiles> <compute_mfma_per_wave> <duty_sleep>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int BLK = 64 * 128;  // floats per block
constexpr int NBLK = 9;

#ifdef CHEAP_UPDATE
__device__ __forceinline__ float upd(float w, float& t, float& m, float& v, float g) {  // no division / square root
    m = m + g * 0.1f, v = v + g * 0.001f;
    const float wn = w - m * 1e-3f;
    t = wn * 0.001f + t * 0.999f;
    return wn;
}
#else
__device__ __forceinline__ float upd(float w, float& t, float& m, float& v, float g) {
#pragma clang fp contract(off)
    m = m + (g - m) * 0.1f;
    v = v + (g * g - v) * 0.001f;
    const float wn = w - (m * 1e-3f) / (sqrtf(v) + 1e-7f);
    t = wn * 0.001f + t * 0.999f;
    return wn;
}
#endif

template <bool F4, int ROWMAP, int HALF = 0>
__global__ __launch_bounds__(256, F4 ? (HALF ? 3 : 2) : 1) void tile_kernel(const float* w, float* wo, float* __restrict__ t,
                                                    float* __restrict__ m, float* __restrict__ v, int tiles_total,
                                                    int compute_mfma, int duty_sleep, int do_stream, int stagger, int* __restrict__ slots, int* __restrict__ counter,
                                                    float* __restrict__ sinkbuf) {
    extern __shared__ float lds[];
    __shared__ int s_tile;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float av = 1e-3f * lane, bv = 1e-3f * (lane + 1);
    if (stagger > 0) {  // every second workgroup that lands on a CU starts `stagger` x 8128 cycles late (anti-phase)
        __shared__ int s_par;
        if (threadIdx.x == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
            s_par = atomicAdd(&slots[((xcc & 15) << 8) | ((hw >> 8) & 0xFF)], 1) & 1;
        }
        __syncthreads();
        if (s_par)
            for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    for (;;) {
        if (threadIdx.x == 0) s_tile = atomicAdd(counter, 1);
        __syncthreads();
        const int tile = s_tile;
        __syncthreads();
        if (tile >= tiles_total) break;
        // ---- compute phase: bursts of 64 MFMAs, then idle (models VALU / LDS / barrier time of the real kernel)
        for (int i = 0; i < compute_mfma; i += 64) {
            if (duty_sleep >= 0) {
#pragma unroll
                for (int k = 0; k < 64; ++k) acc[k & 7] = MFMA16(av, bv, acc[k & 7]);
                for (int s = 0; s < duty_sleep; ++s) __builtin_amdgcn_s_sleep(16);
            } else {  // duty_sleep < 0: the compute phase idles for the same time (no matrix work at all)
                for (int s = 0; s < 2 - duty_sleep; ++s) __builtin_amdgcn_s_sleep(16);
            }
        }
        // ---- stream phase
        const long abase = (long)tile * NBLK * BLK;
        for (int b = 0; b < (do_stream ? NBLK : 0); ++b) {
            const long bbase = abase + (long)b * BLK;
            const float* w_ = w + bbase; float* wo_ = wo + bbase; float* t_ = t + bbase; float* m_ = m + bbase; float* v_ = v + bbase;
            if constexpr (F4) {
                const int col = 64 * (wave & 1) + 4 * lr, r0 = 32 * (wave >> 1);
                constexpr int NQ = HALF ? 4 : 8;  // rows per lane and sub-block
#pragma unroll 1
                for (int h = 0; h < (HALF ? 2 : 1); ++h) {
                    f32x4 qw[NQ], qt[NQ], qm[NQ], qv[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int rj = h * 4 + q;
                        const int o = (r0 + (ROWMAP ? 4 * rj + lg : 8 * lg + rj)) * 128 + col;
                        qw[q] = *(const f32x4*)(w_ + o), qt[q] = *(const f32x4*)(t_ + o);
                        qm[q] = *(const f32x4*)(m_ + o), qv[q] = *(const f32x4*)(v_ + o);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < (HALF ? 80 : 160); ++k) acc[k & 7] = MFMA16(av, bv, acc[k & 7]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int rj = h * 4 + q;
                        const int o = (r0 + (ROWMAP ? 4 * rj + lg : 8 * lg + rj)) * 128 + col;
                        f32x4 ow;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float tt = qt[q][e], mm = qm[q][e], vv = qv[q][e];
                            ow[e] = upd(qw[q][e], tt, mm, vv, acc[q][e] + 1e-3f);
                            qt[q][e] = tt, qm[q][e] = mm, qv[q][e] = vv;
                        }
                        *(f32x4*)(wo_ + o) = ow, *(f32x4*)(t_ + o) = qt[q], *(f32x4*)(m_ + o) = qm[q], *(f32x4*)(v_ + o) = qv[q];
                    }
                }
            } else {
                const int col = 32 * wave + 2 * lr;
                f32x2 qw[16], qt[16], qm[16], qv[16];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ta = 0; ta < 4; ++ta) {
                        const int o = (4 * (lg * 4 + j) + ta) * 128 + col;
                        qw[j * 4 + ta] = *(const f32x2*)(w_ + o), qt[j * 4 + ta] = *(const f32x2*)(t_ + o);
                        qm[j * 4 + ta] = *(const f32x2*)(m_ + o), qv[j * 4 + ta] = *(const f32x2*)(v_ + o);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 160; ++k) acc[k & 7] = MFMA16(av, bv, acc[k & 7]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ta = 0; ta < 4; ++ta) {
                        const int q = j * 4 + ta;
                        const int o = (4 * (lg * 4 + j) + ta) * 128 + col;
                        f32x2 ow;
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                        {
                            float tt = qt[q][e], mm = qm[q][e], vv = qv[q][e];
                            ow[e] = upd(qw[q][e], tt, mm, vv, acc[q & 7][e] + 1e-3f);
                            qt[q][e] = tt, qm[q][e] = mm, qv[q][e] = vv;
                        }
                        *(f32x2*)(wo_ + o) = ow, *(f32x2*)(t_ + o) = qt[q], *(f32x2*)(m_ + o) = qm[q], *(f32x2*)(v_ + o) = qv[q];
                    }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sinkbuf[threadIdx.x] = s + lds[threadIdx.x];
}

int main(int argc, char** argv) {
    const int lds_kb = argc > 1 ? atoi(argv[1]) : 150;
    const int f4 = argc > 2 ? atoi(argv[2]) : 0;
    const int tiles = argc > 3 ? atoi(argv[3]) : 20480;
    const int cm = argc > 4 ? atoi(argv[4]) : 4800;
    const int duty = argc > 5 ? atoi(argv[5]) : 2;
    const int stream = argc > 6 ? atoi(argv[6]) : 1;
    const int rowmap = argc > 7 ? atoi(argv[7]) : 0;
    const int inplace = argc > 8 ? atoi(argv[8]) : 0;
    const int stagger = argc > 9 ? atoi(argv[9]) : 0;  // start delay of every second workgroup on a CU, in units of 8128 cycles  // 1: updated weights written over the old ones (4 streams, not 5)
    const size_t n = (size_t)(stream ? tiles : 1) * NBLK * BLK;
    float *w, *wo, *t, *m, *v, *sk;
    int* counter;
    hipMalloc(&w, n * 4), hipMalloc(&wo, n * 4), hipMalloc(&t, n * 4), hipMalloc(&m, n * 4), hipMalloc(&v, n * 4);
    hipMalloc(&sk, 4096), hipMalloc(&counter, 4);
    int* slots;
    hipMalloc(&slots, 4096 * 4);
    hipMemset(w, 0, n * 4), hipMemset(wo, 0, n * 4), hipMemset(t, 0, n * 4), hipMemset(m, 0, n * 4), hipMemset(v, 0, n * 4);
    const int wgs_per_cu = lds_kb <= 53 ? 3 : (lds_kb <= 80 ? 2 : 1);
    const int grid = 256 * wgs_per_cu;
    auto k = f4 == 2 ? tile_kernel<true, 0, 1> : f4 ? (rowmap ? tile_kernel<true, 1> : tile_kernel<true, 0>) : tile_kernel<false, 0>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(counter, 0, 4);
        hipMemset(slots, 0, 4096 * 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_kb * 1024, 0, w, inplace ? w : wo, t, m, v, tiles, cm, duty, stream, stagger, slots, counter, sk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = 8.0 * NBLK * BLK * 4 * tiles;
    printf("lds=%3d KB (%d WG/CU) f4=%d tiles=%d compute_mfma/wave=%d duty_sleep=%d stream=%d rowmap=%d inplace=%d stagger=%d: %.3f ms  %.1f us/tile/CU  %.2f TB/s (r+w)\n",
           lds_kb, wgs_per_cu, f4, tiles, cm, duty, stream, rowmap, inplace, stagger, best, best * 1e3 * 256 / tiles, bytes / best * 1e-9);
    return 0;
}
