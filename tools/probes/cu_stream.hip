// Probe (not product): how fast can ONE workgroup of 4 waves (1 wave/SIMD, LDS-limited to 1 WG/CU) stream an
// Adam-like update (read 4 arrays, write 4) as a function of loads in flight per wave and of how many CUs stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int DEPTH, class V>
__global__ __launch_bounds__(256) void stream_kernel(const V* __restrict__ w, V* __restrict__ wo, V* __restrict__ t,
                                                      V* __restrict__ m, V* __restrict__ v, long per_wg) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) lds[0] = 0.f;
    const long base = (long)blockIdx.x * per_wg;
    for (long i = threadIdx.x; i < per_wg; i += 256L * DEPTH) {
        V a[DEPTH], b[DEPTH], c[DEPTH], d[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const long j = base + i + 256L * k;
            a[k] = w[j], b[k] = t[j], c[k] = m[j], d[k] = v[j];
        }
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const long j = base + i + 256L * k;
            V mm = c[k] + (a[k] - c[k]) * 0.1f, vv = d[k] + (a[k] * a[k] - d[k]) * 0.001f;
            V ww = a[k] - mm * 0.01f;
            wo[j] = ww, t[j] = ww * 0.001f + b[k] * 0.999f, m[j] = mm, v[j] = vv;
        }
    }
}

template <int DEPTH, class V>
void run(int wgs, long bytes_per_wg_per_array, float* bufs[5]) {
    const long per_wg = bytes_per_wg_per_array / sizeof(V);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    auto launch = [&] {
        hipLaunchKernelGGL((stream_kernel<DEPTH, V>), dim3(wgs), dim3(256), 150 * 1024, 0, (const V*)bufs[0], (V*)bufs[1],
                           (V*)bufs[2], (V*)bufs[3], (V*)bufs[4], per_wg);
    };
    hipFuncSetAttribute((const void*)stream_kernel<DEPTH, V>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double bytes = 8.0 * bytes_per_wg_per_array * wgs;
    printf("V=%zuB depth=%3d (%5zu B in flight/wave) wgs=%4d: %.3f ms  %.1f GB/s total  %.1f GB/s per CU\n", sizeof(V), DEPTH,
           DEPTH * 4 * 64 * sizeof(V), wgs, ms, bytes / ms * 1e-6, bytes / ms * 1e-6 / wgs);
}

int main() {
    const long per = 64L << 20;  // 64 MiB per array per... total arrays sized for 256 WGs x 1 MiB... use 16 MiB per WG
    const long bytes_per_wg = 8L << 20;
    float* bufs[5];
    for (auto& b : bufs) {
        hipMalloc(&b, bytes_per_wg * 256);
        hipMemset(b, 0, bytes_per_wg * 256);
    }
    (void)per;
    for (int wgs : {16, 64, 256}) {
        run<2, f2>(wgs, bytes_per_wg, bufs);
        run<4, f2>(wgs, bytes_per_wg, bufs);
        run<8, f2>(wgs, bytes_per_wg, bufs);
        run<16, f2>(wgs, bytes_per_wg, bufs);
        run<4, f4>(wgs, bytes_per_wg, bufs);
        run<8, f4>(wgs, bytes_per_wg, bufs);
        run<16, f4>(wgs, bytes_per_wg, bufs);
    }
    return 0;
}
