// Probe: what the global_load_lds (LDS-DMA) path delivers per CU when the source sits in L2 -- the stream every fw:: kernel of
// csrc/wide.hip lives on (32 KB per step and workgroup). One workgroup of 8 waves per CU; every wave requests PER 1 KB pieces per
// step (16 B per lane), keeps DEPTH steps in flight with counted vmcnt waits, and the workgroup meets at a barrier per step (as
// the kernels do). Sources: `shared` = all workgroups walk the SAME 2 MB (one weight block: fwd_gen's pattern), `private` = a 512 KB
// region per workgroup group of 4 (dx_gen's A tiles, shared by the 4 feature blocks) + a shared 512 KB block, `hbm` = every
// workgroup its own large region (no reuse at all). With M MFMAs per wave and step beside the stream (0 = the stream alone).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/lds_dma_bw.hip -o tools/probes/lds_dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void fill_random(unsigned* d, long n) {  // finite bf16 pairs with toggling bits (|x| < 2): what the matrix cores draw on real data
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u;
        x ^= x >> 15, x *= 2246822519u, x ^= x >> 13;
        d[i] = x & 0xbfffbfffu & ~0x40004000u | 0x30003000u;
    }
}

template <int PER, int MF>
__global__ __launch_bounds__(512) void k(const char* src, long wg_stride, int wg_div, long region, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* base = src + (long)(blockIdx.x / wg_div) * wg_stride;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(lane * 0.01f + i), b[i] = (__bf16)(i * 0.25f);
    f32x16 acc[4] = {};
    long off = 0;
    constexpr int STG = 8 * PER * 1024;
    for (int s = 0; s < steps; ++s) {
        unsigned char* l = lds + (s % 3) * STG;
        unsigned vw = lane * 16u;
        asm volatile("" : "+v"(vw));
#pragma unroll
        for (int i = 0; i < PER; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(base + off + (long)(wave * PER + i) * 1024 + vw), (lptr_t)(l + (wave * PER + i) * 1024), 16, 0, 0);
        off += STG;
        if (off >= region) off = 0;
#pragma unroll
        for (int m = 0; m < MF; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        // one step stays in flight
        if (PER == 4) __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
        if (PER == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | 2);
        if (PER == 8) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
        __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    float r = lds[threadIdx.x * 4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int PER, int MF>
void run(const char* name, const char* src, long wg_stride, int wg_div, long region, float* out, int nwg) {
    const int steps = 20000;
    (void)hipFuncSetAttribute((const void*)k<PER, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 8 * PER * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    k<PER, MF><<<nwg, 512, 3 * 8 * PER * 1024>>>(src, wg_stride, wg_div, region, out, 200);
    (void)hipEventRecord(e0);
    k<PER, MF><<<nwg, 512, 3 * 8 * PER * 1024>>>(src, wg_stride, wg_div, region, out, steps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nwg * steps * 8 * PER * 1024;
    printf("%-8s %2d KB/step %2d MFMA/wave/step: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU @2.2GHz  %5.0f cycles/step  MFMA busy %4.1f %%\n", name, 8 * PER, MF, ms,
           bytes / ms * 1e-9, bytes / nwg / (ms * 1e-3 * 2.2e9), ms * 1e-3 * 2.2e9 / steps, 100.0 * MF * 2 * 32 / (ms * 1e-3 * 2.2e9 / steps));
}


// dx_gen's step without ping-pong: every wave requests its 4 pieces of chunk s + 2, reads 12 fragments (4 A + 8 B, 16 B per lane)
// of chunk s from LDS and multiplies 2 x 4 tiles over two k-steps (16 MFMAs); one barrier per step. XCD-aware sources: the
// workgroups of an XCD (id % 8) walk `region` bytes per group of `wg_div` slots.
template <int PP, int ROWS, int VADDR = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void g(const char* src, long grp_stride, int wg_div, long region, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const char* base = src + (long)(xcd * 8 + slot / wg_div) * grp_stride;
    const int rq = wave & 3, fh = wave >> 2, sw = (r >> 2) & 3;
    const int ra0 = ((64 * rq + r) * 32 + (((0 + h) ^ sw) << 3)) * 2, ra1 = ((64 * rq + r) * 32 + (((2 + h) ^ sw) << 3)) * 2;
    const int rb0 = 16384 + ((128 * fh + r) * 32 + (((0 + h) ^ sw) << 3)) * 2, rb1 = 16384 + ((128 * fh + r) * 32 + (((2 + h) ^ sw) << 3)) * 2;
    f32x16 acc[2][4] = {};
    long off = 0;
    constexpr int STG = 32768;
    auto dma = [&](int stg) {
        unsigned char* l = lds + stg * STG;
        // ROWS: a piece = 16 rows x 64 B of a row-major matrix with 2 KB rows (what dx_gen / fwd_gen request: half a cache line per
        // row), the step's chunk 64 B further along the rows; else 1 KB contiguous
        // ROWS == 2: the four 16-byte pieces of a row requested in the XOR-swizzled lane order dx_gen / fwd_gen use
        unsigned vw = ROWS == 2 ? (lane >> 2) * 2048u + (((lane & 3) ^ ((lane >> 4) & 3)) * 16u) : ROWS ? (lane >> 2) * 2048u + (lane & 3) * 16u : lane * 16u;
        asm volatile("" : "+v"(vw));
        unsigned long long vq = vw;  // VADDR: the 64-bit per-lane address form (global_load_lds v[a:a+1], off) instead of scalar base + lane offset
        if (VADDR) asm volatile("" : "+v"(vq));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(base + (ROWS ? (off & 2047) + (off >> 11) * (512L * 2048) + (long)(wave * 4 + i) * 16 * 2048 : off + (long)(wave * 4 + i) * 1024) + (VADDR ? vq : (unsigned long long)vw)),
                                             (lptr_t)(l + (wave * 4 + i) * 1024), 16, 0, 0);
        off += ROWS ? 64 : STG;
        if (off >= (ROWS ? region / 512 : region)) off = 0;
    };
    dma(0), dma(1);
    __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
    __builtin_amdgcn_s_barrier();
    if (PP && fh == 1) {
        __builtin_amdgcn_s_barrier();
    }
    int stg = 0;
#ifdef PROBE_STAMP
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tlast;
    auto now = []() {
        unsigned long long t_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        return t_;
    };
    tlast = now();
#define ST(i) { const unsigned long long t_ = now(); tacc[i] += t_ - tlast; tlast = t_; }
#else
#define ST(i)
#endif
    for (int s = 0; s < steps; ++s) {
        const unsigned char* l = lds + stg * STG;
        bf16x8 A[2][2], B[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) A[0][rt] = *(const bf16x8*)(l + ra0 + rt * 2048), A[1][rt] = *(const bf16x8*)(l + ra1 + rt * 2048);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) B[0][ft] = *(const bf16x8*)(l + rb0 + ft * 2048), B[1][ft] = *(const bf16x8*)(l + rb1 + ft * 2048);
        dma((stg + 2) % 3);
        if (PP) {
#ifdef PROBE_STAMP
            ST(0);
            __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
            ST(1);
#else
            __builtin_amdgcn_s_waitcnt(0x0070 | 4);
#endif
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            ST(2);
            __builtin_amdgcn_s_setprio(3);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
#pragma unroll
#ifdef PROBE_F16  // (the same bytes as fp16 operands: v_mfma_f32_32x32x16_f16, the split learner's forward type)
                for (int rt = 0; rt < 2; ++rt) {
                    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
                    acc[rt][ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[ks][rt]), __builtin_bit_cast(f16x8, B[ks][ft]), acc[rt][ft], 0, 0, 0);
                }
#else
                for (int rt = 0; rt < 2; ++rt) acc[rt][ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ks][rt], B[ks][ft], acc[rt][ft], 0, 0, 0);
#endif
        if (PP) {
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
        ST(3);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ST(4);
        stg = stg + 1 == 3 ? 0 : stg + 1;
    }
#ifdef PROBE_STAMP
    if (blockIdx.x == 16 && lane == 0 && steps > 1000)
        printf("wave %d per step: reads + requests landed %llu, vm wait %llu, barrier %llu, multiply %llu, barrier %llu\n", wave, tacc[0] / steps, tacc[1] / steps, tacc[2] / steps,
               tacc[3] / steps, tacc[4] / steps);
#endif
    if (PP && fh == 0) {
        __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    float rr = 0;
    for (int a = 0; a < 2; ++a)
        for (int t = 0; t < 4; ++t)
            for (int i = 0; i < 16; ++i) rr += acc[a][t][i];
    out[blockIdx.x * 512 + threadIdx.x] = rr;
}
template <int PP, int ROWS, int VADDR = 0>
void rung(const char* name, const char* src, long grp_stride, int wg_div, long region, float* out, int nwg) {
    const int steps = 20000;
    const int lds_bytes = getenv("PROBE_LDS_KB") ? atoi(getenv("PROBE_LDS_KB")) * 1024 : 3 * 32768;
    (void)hipFuncSetAttribute((const void*)g<PP, ROWS, VADDR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    g<PP, ROWS, VADDR><<<nwg, 512, lds_bytes>>>(src, grp_stride, wg_div, region, out, 200);
    (void)hipEventRecord(e0);
    g<PP, ROWS, VADDR><<<nwg, 512, lds_bytes>>>(src, grp_stride, wg_div, region, out, steps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nwg * steps * 32768;
    printf("gemm-like %-22s %s %s: %7.3f ms  %6.2f TB/s  %5.0f cycles/step  MFMA busy %4.1f %%\n", name, ROWS == 2 ? "64 B x 16 rows, pieces swizzled" : ROWS ? (VADDR ? "64 B x 16 rows, 64-bit lane addresses" : "64 B x 16 rows") : "1 KB pieces   ", PP ? "ping-pong  " : "one program", ms, bytes / ms * 1e-9,
           ms * 1e-3 * 2.2e9 / steps, 100.0 * 16 * 2 * 32 / (ms * 1e-3 * 2.2e9 / steps));
}

int main() {
    const int nwg = 256;
    const long big = 256L * 8 * 1024 * 1024;  // 2 GB
    char* src;
    float* out;
    (void)hipMalloc(&src, big);
    (void)hipMemset(src, 1, big);
    if (getenv("PROBE_RANDOM")) fill_random<<<4096, 256>>>((unsigned*)src, big / 4);
    (void)hipDeviceSynchronize();
    (void)hipMalloc(&out, nwg * 512 * 4);
    if (getenv("PROBE_LOOP")) {
        for (int i = 0; i < atoi(getenv("PROBE_LOOP")); ++i) {
            rung<1, 1>("1 MB per 4 slots", src, 1L << 20, 4, 1L << 20, out, nwg);
            if (getenv("PROBE_SWZ")) rung<1, 2>("1 MB per 4 slots", src, 1L << 20, 4, 1L << 20, out, nwg);
        }
        return 0;
    }
    // all workgroups the same 2 MB
    run<4, 0>("shared", src, 0, 1, 2L << 20, out, nwg);
    run<4, 8>("shared", src, 0, 1, 2L << 20, out, nwg);
    run<4, 16>("shared", src, 0, 1, 2L << 20, out, nwg);
    run<2, 0>("shared", src, 0, 1, 2L << 20, out, nwg);
    // groups of 4 workgroups share 512 KB (walked again and again: L2-resident, 32 MB over the chip)
    run<4, 0>("quad", src, 512L << 10, 4, 512L << 10, out, nwg);
    run<4, 16>("quad", src, 512L << 10, 4, 512L << 10, out, nwg);
    // every workgroup its own 8 MB: from memory
    run<4, 0>("hbm", src, 8L << 20, 1, 8L << 20, out, nwg);
    run<4, 16>("hbm", src, 8L << 20, 1, 8L << 20, out, nwg);
    // the gemm-like step: all workgroups of the chip the same 512 KB; per XCD one 512 KB block per 4 slots (dx_gen's A tiles: 8 x 512 KB
    // per L2, walked round and round); the same from memory (8 MB per group)
    rung<0, 0>("same 512 KB", src, 0, 1, 512L << 10, out, nwg);
    rung<1, 0>("same 512 KB", src, 0, 1, 512L << 10, out, nwg);
    rung<0, 0>("512 KB per 4 slots", src, 512L << 10, 4, 512L << 10, out, nwg);
    rung<1, 0>("512 KB per 4 slots", src, 512L << 10, 4, 512L << 10, out, nwg);
    rung<0, 0>("8 MB per 4 slots", src, 8L << 20, 4, 8L << 20, out, nwg);
    rung<1, 0>("8 MB per 4 slots", src, 8L << 20, 4, 8L << 20, out, nwg);
    // the same with row-major sources: 512 rows x 2 KB = 1 MB per group (two tiles of 256 rows: both operands of dx_gen)
    rung<1, 1>("same 1 MB", src, 0, 1, 1L << 20, out, nwg);
    rung<1, 1>("1 MB per 4 slots", src, 1L << 20, 4, 1L << 20, out, nwg);
    rung<0, 1>("1 MB per 4 slots", src, 1L << 20, 4, 1L << 20, out, nwg);
    rung<1, 1, 1>("1 MB per 4 slots", src, 1L << 20, 4, 1L << 20, out, nwg);
    return 0;
}
