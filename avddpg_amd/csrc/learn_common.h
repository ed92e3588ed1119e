// Building blocks shared by the learn kernels (mlp.hip: learn_kernel_t / learn_kernel_g; lean.hip: learn_kernel_l):
// sinks (gradient store vs fused Adam+Polyak), LDS-only barrier, width-1 output layer forward/backward, the generic
// MFMA GEMM routines over LDS-resident activations, LDS carving of the 1-workgroup-per-CU kernels.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace avd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = 64;       // batch rows per workgroup (= batch_size 64, config.py:106)
constexpr int NTHREADS = 256;  // 4 waves, one per SIMD
constexpr float BN_EPS = 1e-3f;

// LDS row stride: a multiple of 4 floats (16-byte rows for b128 reads) with (ld/4) odd so that 16 consecutive rows
// start on 16 different 4-bank slots
__host__ __device__ constexpr int ld_of(int k) { return ((k >> 2) & 1) ? k : k + 4; }

#ifdef AVD_PHASE_TIMING
// Diagnostic build only (tools/phase_profile.py @ tag r06-pre-prune): per-phase shader-cycle sums of workgroup thread 0.
static __device__ unsigned long long g_phase_cycles[32];  // per translation unit (mlp.hip reads its own)
#define PH_INIT() unsigned long long ph_last = clock64()
#define PH(id)                                                                  \
    do {                                                                        \
        if (threadIdx.x == 0) {                                                 \
            const unsigned long long ph_now = clock64();                        \
            atomicAdd(&g_phase_cycles[id], ph_now - ph_last);                   \
            ph_last = ph_now;                                                   \
        }                                                                       \
    } while (0)
#define PHX_T0() unsigned long long phx_t = clock64()
#define PHX(id)                                                                 \
    do {                                                                        \
        if (threadIdx.x == 0) {                                                 \
            const unsigned long long phx_n = clock64();                         \
            atomicAdd(&g_phase_cycles[id], phx_n - phx_t);                      \
            phx_t = phx_n;                                                      \
        }                                                                       \
    } while (0)
#define PH_ARG , unsigned long long& ph_last
#define PH_PASS , ph_last
// In-kernel clock (MI355X_MICROARCH.md, DVFS item 6): slot 30 sums the shader-cycle counter (s_memtime), slot 31 the
// constant 100 MHz counter (s_memrealtime) over each workgroup's lifetime; clock = 100 MHz * [30] / [31].
#define PH_CLK_INIT()                                                   \
    const unsigned long long ph_c0 = __builtin_amdgcn_s_memtime();     \
    const unsigned long long ph_r0 = __builtin_amdgcn_s_memrealtime()
#define PH_CLK_END()                                                                        \
    do {                                                                                    \
        if (threadIdx.x == 0) {                                                             \
            atomicAdd(&g_phase_cycles[30], __builtin_amdgcn_s_memtime() - ph_c0);           \
            atomicAdd(&g_phase_cycles[31], __builtin_amdgcn_s_memrealtime() - ph_r0);       \
        }                                                                                   \
    } while (0)
#else
#define PH_CLK_INIT()
#define PH_CLK_END()
#define PH_INIT()
#define PHX_T0()
#define PHX(id)
#define PH(id)
#define PH_ARG
#define PH_PASS
#endif

// Workgroup barrier that orders LDS traffic only. __syncthreads() also drains vmcnt, i.e. waits for every
// outstanding gradient STORE to be acknowledged by HBM; nothing in learn_kernel re-reads what it stored.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Where a parameter gradient goes. Gradient routines address the agent's slab by pointer; a sink decides what a
// "store" means: StoreSink writes the gradient itself (grads slab); AdamSink treats the pointer as the position of
// the parameter in the OUTPUT weight slab and applies Adam + Polyak right there (fused update, no gradient slab).
struct StoreSink {
    __device__ __forceinline__ void put(float* p, float g) const { *p = g; }
};
struct AdamSink {
    float* wo;        // agent's slab in theta_out (updated weights are written here)
    const float* wi;  // same agent in theta (pre-update weights: every forward/backward of the step reads these)
    float *wt, *m, *v;
    float alpha_a, alpha_c, tau, omt;
    int actor_size;
    __device__ __forceinline__ void put(float* p, float g) const {
#pragma clang fp contract(off)
        const long off = p - wo;
        const float alpha = off < actor_size ? alpha_a : alpha_c;
        float mm = m[off], vv = v[off];
        mm = mm + (g - mm) * (1.0f - 0.9f);          // TF ApplyAdam, identical to adam_polyak_kernel (optim.hip)
        vv = vv + (g * g - vv) * (1.0f - 0.999f);
        const float w = wi[off] - (mm * alpha) / (sqrtf(vv) + 1e-7f);
        m[off] = mm, v[off] = vv;
        *p = w;
        wt[off] = w * tau + wt[off] * omt;  // update_target on the freshly updated weight
    }
    // Two-phase form for bulk gradients (weight-gradient GEMM epilogues): all operand loads of a block are issued
    // first (load2), the arithmetic and the stores follow (update2) -- one memory round trip per block, not per element.
    struct Quad {
        float w[2], t[2], m[2], v[2];
    };
    __device__ __forceinline__ void load2(Quad& q, long off) const {
        const float2 a = *(const float2*)(wi + off), b = *(const float2*)(wt + off);
        const float2 c = *(const float2*)(m + off), d = *(const float2*)(v + off);
        q.w[0] = a.x, q.w[1] = a.y, q.t[0] = b.x, q.t[1] = b.y, q.m[0] = c.x, q.m[1] = c.y, q.v[0] = d.x, q.v[1] = d.y;
    }
    __device__ __forceinline__ void update2(const Quad& q, long off, const float (&g)[2]) const {
#pragma clang fp contract(off)
        const float alpha = off < actor_size ? alpha_a : alpha_c;
        float2 ow, ot, om, ov;
        float* pw = &ow.x;
        float* pt = &ot.x;
        float* pm = &om.x;
        float* pv = &ov.x;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float mm = q.m[e] + (g[e] - q.m[e]) * (1.0f - 0.9f);
            const float vv = q.v[e] + (g[e] * g[e] - q.v[e]) * (1.0f - 0.999f);
            // exact div/sqrt: approximate rcp/sqrt measured no faster (the epilogue is bound by per-CU memory throughput)
            const float w = q.w[e] - (mm * alpha) / (sqrtf(vv) + 1e-7f);
            pm[e] = mm, pv[e] = vv, pw[e] = w, pt[e] = w * tau + q.t[e] * omt;
        }
        *(float2*)(wo + off) = ow;
        *(float2*)(wt + off) = ot;
        *(float2*)(m + off) = om;
        *(float2*)(v + off) = ov;
    }
    // Four-column form (16-byte accesses): the operands of 4 consecutive columns of one row.
    struct Quad4 {
        f32x4 w, t, m, v;
    };
    __device__ __forceinline__ void load4(Quad4& q, long off) const {
        q.w = *(const f32x4*)(wi + off), q.t = *(const f32x4*)(wt + off);
        q.m = *(const f32x4*)(m + off), q.v = *(const f32x4*)(v + off);
    }
    __device__ __forceinline__ void update4(const Quad4& q, long off, const float (&g)[4]) const {
#pragma clang fp contract(off)
        const float alpha = off < actor_size ? alpha_a : alpha_c;
        f32x4 ow, ot, om, ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float mm = q.m[e] + (g[e] - q.m[e]) * (1.0f - 0.9f);
            const float vv = q.v[e] + (g[e] * g[e] - q.v[e]) * (1.0f - 0.999f);
            const float w = q.w[e] - (mm * alpha) / (sqrtf(vv) + 1e-7f);
            om[e] = mm, ov[e] = vv, ow[e] = w, ot[e] = w * tau + q.t[e] * omt;
        }
        *(f32x4*)(wo + off) = ow;
        *(f32x4*)(wt + off) = ot;
        *(f32x4*)(m + off) = om;
        *(f32x4*)(v + off) = ov;
    }
    // Pointer form (lean.hip): the caller passes uniform base + per-lane u32 offset addresses, one per array.
    __device__ __forceinline__ static void load4p(Quad4& q, const float* pw, const float* pt, const float* pm, const float* pv) {
        q.w = *(const f32x4*)pw, q.t = *(const f32x4*)pt, q.m = *(const f32x4*)pm, q.v = *(const f32x4*)pv;
    }
    __device__ __forceinline__ void update4p(const Quad4& q, float alpha, float* pwo, float* pt, float* pm, float* pv,
                                             const float (&g)[4]) const {
#pragma clang fp contract(off)
        f32x4 ow, ot, om, ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float mm = q.m[e] + (g[e] - q.m[e]) * (1.0f - 0.9f);
            const float vv = q.v[e] + (g[e] * g[e] - q.v[e]) * (1.0f - 0.999f);
            const float w = q.w[e] - (mm * alpha) / (sqrtf(vv) + 1e-7f);
            om[e] = mm, ov[e] = vv, ow[e] = w, ot[e] = w * tau + q.t[e] * omt;
        }
        *(f32x4*)pwo = ow;
        *(f32x4*)pt = ot;
        *(f32x4*)pm = om;
        *(f32x4*)pv = ov;
    }
};

struct Net {  // pointers into one weight set
    const float* th;
    const float* st;
};

// Optimisation fences. LLVM's loop-invariant code motion otherwise hoists every `base + lane offset` address of the
// pass loop's body (a hundred 64-bit values) above the loop and spills them to scratch; a value that passes through
// one of these inside the loop body is opaque, so its uses are recomputed (one add) where they are needed.
__device__ __forceinline__ int opaque_zero() {  // a wave-uniform 0 the optimiser cannot see through
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
}

// ------------------------------------------------------------------------------------------
// small building blocks (called by all 256 threads of the workgroup)
// ------------------------------------------------------------------------------------------

// Row loops of the VALU phases work in register blocks of RB rows: all RB reads are issued before the first
// FMA/ds_write (a read -> compute -> write loop pays one LDS round trip per row: LDS returns in order).
constexpr int RB = 16;

// out[r] = sum_k (P[r][k]*inv[k] + sh[k]) * w[k] + b   (output width 1), blockDim/64 lanes per row, 16-byte LDS reads
__device__ __forceinline__ float out_layer_row(const float* P, int ld, const float* inv, const float* sh,
                                               const float* w, float b, int K) {
    const int lpr = blockDim.x >> 6;  // lanes per row: 4 (256 threads) or 8 (512 threads)
    const int r = threadIdx.x / lpr, part = threadIdx.x % lpr;
    float acc = 0.f;
    for (int k = 4 * part; k < K; k += 4 * lpr) {
        const f32x4 p = *(const f32x4*)(P + r * ld + k);
        const f32x4 y = p * *(const f32x4*)(inv + k) + *(const f32x4*)(sh + k);
        const f32x4 wk = *(const f32x4*)(w + k);
        acc = fmaf(y[0], wk[0], acc), acc = fmaf(y[1], wk[1], acc), acc = fmaf(y[2], wk[2], acc),
        acc = fmaf(y[3], wk[3], acc);
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (lpr == 8) acc += __shfl_xor(acc, 4);
    return acc + b;  // valid in all lanes of row r
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// MFMA operand conventions used below (v_mfma_f32_16x16x4_f32, lane = 16*lg + lr):
//   A[i = lr][kk = lg], B[kk = lg][j = lr], D[i = 4*lg + reg][j = lr].
// The reduction index kk and the tile column j may be permuted freely as long as A and B agree, so every
// routine picks the permutation that turns its operand fetches into 8/16-byte accesses:
//   * reduction permuted: the 4 MFMAs of a 16-deep block take kk = 4*lg + jj (jj = 0..3) -> one b128 per block;
//   * columns permuted:   n-tile t holds columns base + NT*lr + t -> one NT-float load feeds NT tiles.

constexpr int FWD_RING = 4;  // B-operand register ring: blocks of 16 k in flight ahead of the MFMAs

// One (16*MT rows) x 32 columns output tile of gemm_fwd_relu: rows (m0 + m) * 16 .., m < MT.
template <int MT>
__device__ __forceinline__ void fwd_relu_tile(const float* X, int ldx, const float* inv, const float* sh, int K,
                                              const float* __restrict__ W, const float* __restrict__ b, int N, float* out,
                                              int ldo, int n0, int m0) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    const int nblk = K >> 4;
    f32x4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m][0] = acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wp = W + (long)(4 * lg) * N + n0 + 2 * lr;  // row k = 16*blk + 4*lg + jj, columns n0+2lr, +1
    f32x2 ring[FWD_RING][4];
#pragma unroll
    for (int d = 0; d < FWD_RING - 1; ++d)
        if (d < nblk)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) ring[d][jj] = *(const f32x2*)(wp + (long)(16 * d + jj) * N);
    for (int kb = 0; kb < nblk; kb += FWD_RING) {
#pragma unroll
        for (int d = 0; d < FWD_RING; ++d) {
            const int blk = kb + d;
            if (blk < nblk) {
                const int pre = blk + FWD_RING - 1;
                if (pre < nblk)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        ring[(d + FWD_RING - 1) % FWD_RING][jj] = *(const f32x2*)(wp + (long)(16 * pre + jj) * N);
                const int k4 = 16 * blk + 4 * lg;
                const f32x4 iv = *(const f32x4*)(inv + k4);
                const f32x4 sf = *(const f32x4*)(sh + k4);
                f32x4 a[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f32x4 xv = *(const f32x4*)(X + ((m0 + m) * 16 + lr) * ldx + k4);
                    a[m] = xv * iv + sf;
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        acc[m][0] = MFMA16(a[m][jj], ring[d][jj][0], acc[m][0]);
                        acc[m][1] = MFMA16(a[m][jj], ring[d][jj][1], acc[m][1]);
                    }
            }
        }
    }
    const f32x2 bc = *(const f32x2*)(b + n0 + 2 * lr);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 o;
            o[0] = fmaxf(acc[m][0][j] + bc[0], 0.f);
            o[1] = fmaxf(acc[m][1][j] + bc[1], 0.f);
            *(f32x2*)(out + ((m0 + m) * 16 + lg * 4 + j) * ldo + n0 + 2 * lr) = o;
        }
}

// Forward hidden layer: out[r][n] = relu(sum_k bn(X[r][k]) * W[k][n] + b[n]); X,out in LDS, W global [K][N].
// K % 16 == 0, N % 32 == 0. Full rounds: wave w owns the 64 x 32 tile of column group 4*round + w. The N/32 % 4 left-over
// column groups are split by ROW tiles as well (16 x 32 pieces dealt over the four waves), so that e.g. N = 160 costs
// 1.25 rounds, not 2 with three waves idle in the second.
__device__ __forceinline__ void gemm_fwd_relu(const float* X, int ldx, const float* inv, const float* sh, int K,
                                              const float* __restrict__ W, const float* __restrict__ b, int N,
                                              float* out, int ldo) {
    const int wave = threadIdx.x >> 6;
    const int groups = N >> 5, full = groups >> 2, rem = groups & 3;
    for (int rd = 0; rd < full; ++rd) fwd_relu_tile<4>(X, ldx, inv, sh, K, W, b, N, out, ldo, (rd * 4 + wave) * 32, 0);
    for (int item = wave; item < rem * 4; item += 4)
        fwd_relu_tile<1>(X, ldx, inv, sh, K, W, b, N, out, ldo, (full * 4 + (item >> 2)) * 32, item & 3);
}

// Weight gradient of a hidden layer fed by a BN output (all operands in LDS):
//   dW[k][n] = inv[k] * sum_r P[r][k]*DZ[r][n] + sh[k]*db[n]   for k < K, n < N  -> global gW[k*N + n]
// Output tile (ta, tb) of a 64x32 block holds rows k0 + 4*i + ta (i = 4*lg + reg) and columns n0 + 2*lr + tb.
// Sink = AdamSink: gW is the tensor's position in the OUTPUT weight slab and every element is updated where it is produced
// (the 16 (row, column-pair) operand groups of a block are requested before its MFMA loop, consumed after it).
template <class Sink = StoreSink>
__device__ __forceinline__ void gemm_dw(const float* P, int ldp, const float* inv, const float* sh, int K,
                                        const float* DZ, int ldz, const float* db, int N, float* __restrict__ gW,
                                        Sink sink = Sink()) {
    constexpr bool kFused = !std::is_same<Sink, StoreSink>::value;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    // (column group of 32, block of 64 feature rows) items dealt round-robin over the four waves
    const int groups = N >> 5, kblocks = (K + 63) >> 6;
    for (int item = wave; item < groups * kblocks; item += 4) {
        const int n0 = (item % groups) * 32;
        const f32x2 dbc = *(const f32x2*)(db + n0 + 2 * lr);
        {
            const int k0 = (item / groups) * 64;
            typename std::conditional<kFused, AdamSink::Quad, int>::type q[16];
            long base = 0;
            if constexpr (kFused) {
                base = (gW - sink.wo) + n0 + 2 * lr;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ta = 0; ta < 4; ++ta)  // clamped: loads for rows past K stay inside the tensor, never used
                        sink.load2(q[j * 4 + ta], base + (long)(min(k0 + 4 * (lg * 4 + j), K - 4) + ta) * N);
            }
            f32x4 acc[4][2];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m][0] = acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* pp = P + lg * ldp + k0 + 4 * lr;  // may read past K in the last block: those rows are not stored
            const float* dp = DZ + lg * ldz + n0 + 2 * lr;
#pragma unroll 4
            for (int r = 0; r < TILE; r += 4) {
                const f32x4 pa = *(const f32x4*)(pp + r * ldp);
                const f32x2 dz = *(const f32x2*)(dp + r * ldz);
#pragma unroll
                for (int ta = 0; ta < 4; ++ta) {
                    acc[ta][0] = MFMA16(pa[ta], dz[0], acc[ta][0]);
                    acc[ta][1] = MFMA16(pa[ta], dz[1], acc[ta][1]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kbase = k0 + 4 * (lg * 4 + j);
                if (kbase < K) {  // K % 4 == 0: the four ta rows are valid together
                    const f32x4 iv = *(const f32x4*)(inv + kbase);
                    const f32x4 sf = *(const f32x4*)(sh + kbase);
#pragma unroll
                    for (int ta = 0; ta < 4; ++ta) {
                        if constexpr (kFused) {
                            const float o[2] = {fmaf(iv[ta], acc[ta][0][j], sf[ta] * dbc[0]),
                                                fmaf(iv[ta], acc[ta][1][j], sf[ta] * dbc[1])};
                            sink.update2(q[j * 4 + ta], base + (long)(kbase + ta) * N, o);
                        } else {
                            f32x2 o;
                            o[0] = fmaf(iv[ta], acc[ta][0][j], sf[ta] * dbc[0]);
                            o[1] = fmaf(iv[ta], acc[ta][1][j], sf[ta] * dbc[1]);
                            *(f32x2*)(gW + (long)(kbase + ta) * N + n0 + 2 * lr) = o;
                        }
                    }
                }
            }
        }
    }
}

constexpr int DX_NB = 16;  // reduction blocks of 16 held in registers per tile (N <= 256)

// Input gradient of a hidden layer + BN/ReLU backward of the layer below, in place:
//   dy[r][c] = sum_n DZ[r][n] * W[c][n]               (c in [c_begin, c_end), W global [K][N])
//   dgamma[c] = sum_r dy*(p - mm[c])*rs[c];  dbeta[c] = sum_r dy;  P[r][c] <- dy * rs*g * (p > 0)
// g/mm/mv/dg/dbe are indexed by (c - c_begin).  dg == nullptr skips the parameter gradients.
// 16-column tiles round-robin over the waves; a tile's whole W slice (16 x N) is fetched with N/16 16-byte
// loads per lane, the NEXT tile's slice being requested before the current tile's MFMAs start.
__device__ __forceinline__ void gemm_dx_bn(const float* DZ, int ldz, int N, const float* __restrict__ W, int c_begin,
                                           int c_end, float* P, int ldp, const float* __restrict__ g,
                                           const float* __restrict__ mm, const float* __restrict__ mv,
                                           float* __restrict__ dg, float* __restrict__ dbe) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    {
        const int nb0 = 0;  // single reduction pass: N <= 16*DX_NB (checked on the host)
        const int nblk = N >> 4;
        f32x4 wc[DX_NB], wn[DX_NB];
        float bnc[3] = {0.f, 0.f, 1.f}, bnn[3] = {0.f, 0.f, 1.f};  // gamma, mean, var of column c0 + lr
        int c0 = c_begin + wave * 16;
        if (c0 < c_end) {
            const float* wrow = W + (long)(c0 + lr) * N + nb0 + 4 * lg;
#pragma unroll
            for (int q = 0; q < DX_NB; ++q)
                if (q < nblk) wc[q] = *(const f32x4*)(wrow + 16 * q);
            bnc[0] = g[c0 + lr - c_begin], bnc[1] = mm[c0 + lr - c_begin], bnc[2] = mv[c0 + lr - c_begin];
        }
        for (; c0 < c_end; c0 += 4 * 16) {
            const int cn = c0 + 4 * 16;
            if (cn < c_end) {
                const float* wrow = W + (long)(cn + lr) * N + nb0 + 4 * lg;
#pragma unroll
                for (int q = 0; q < DX_NB; ++q)
                    if (q < nblk) wn[q] = *(const f32x4*)(wrow + 16 * q);
                bnn[0] = g[cn + lr - c_begin], bnn[1] = mm[cn + lr - c_begin], bnn[2] = mv[cn + lr - c_begin];
            }
            f32x4 acc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < DX_NB; ++q) {
                if (q < nblk) {
                    f32x4 a[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) a[m] = *(const f32x4*)(DZ + (m * 16 + lr) * ldz + nb0 + 16 * q + 4 * lg);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int m = 0; m < 4; ++m) acc[m] = MFMA16(a[m][jj], wc[q][jj], acc[m]);
                }
            }
            const int c = c0 + lr;
            const float rs = 1.0f / sqrtf(bnc[2] + BN_EPS);
            const float gam = bnc[0];
            const float mean = bnc[1];
            float sg = 0.f, sb = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = m * 16 + lg * 4 + j;
                    const float dy = acc[m][j];
                    const float p = P[r * ldp + c];
                    sg = fmaf(dy * (p - mean), rs, sg);
                    sb += dy;
                    P[r * ldp + c] = (p > 0.f) ? dy * (rs * gam) : 0.f;
                }
            sg += __shfl_xor(sg, 16);
            sg += __shfl_xor(sg, 32);
            sb += __shfl_xor(sb, 16);
            sb += __shfl_xor(sb, 32);
            if (dg && lg == 0) {
                dg[c - c_begin] = sg;
                dbe[c - c_begin] = sb;
            }
#pragma unroll
            for (int q = 0; q < DX_NB; ++q) wc[q] = wn[q];
            bnc[0] = bnn[0], bnc[1] = bnn[1], bnc[2] = bnn[2];
        }
    }
}

// Column sums db[n] = sum_r DZ[r][n] -> LDS db[] and global gdb[]
template <class Sink = StoreSink>
__device__ __forceinline__ void col_sums(const float* DZ, int ldz, int N, float* db, float* __restrict__ gdb,
                                         Sink sink = Sink()) {
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < TILE; ++r) s += DZ[r * ldz + n];
        db[n] = s;
        if (gdb) sink.put(gdb + n, s);
    }
}

// First-layer gradients from dz[r][c0..c0+H): dW[j][k] = sum_r X[r*xs+j]*dz[r][k], db[k] = sum_r dz[r][k]
template <int K, class Sink = StoreSink>
__device__ __forceinline__ void dense_in_grads_k(const float* X, int xs, const float* DZ, int ldz, int c0, int H,
                                                 float* __restrict__ gW, float* __restrict__ gb, Sink sink = Sink()) {
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        float acc[K];
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = 0.f;
        float sb = 0.f;
#pragma nounroll  // unrolled, the scheduler hoists all 64 rows' LDS reads to the top and spills them to scratch
        for (int rb = 0; rb < TILE; rb += RB) {  // RB rows of operands in registers before the FMAs
            float dv[RB], xv[RB][K];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                dv[i] = DZ[(rb + i) * ldz + c0 + k];
#pragma unroll
                for (int j = 0; j < K; ++j) xv[i][j] = X[(rb + i) * xs + j];
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                sb += dv[i];
#pragma unroll
                for (int j = 0; j < K; ++j) acc[j] = fmaf(xv[i][j], dv[i], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j) sink.put(gW + j * H + k, acc[j]);
        sink.put(gb + k, sb);
    }
}
// Output layer (width 1) backward through the BN below it, for column k < K (K <= 256):
//   dW3[k] = sum_r bn(p[r][k])*d[r]; dy = d[r]*w3[k]; dgamma, dbeta; DZ[r][k] = dy*inv*(p>0)
// The 64 rows are split over blockDim/K thread groups; partial sums meet in LDS scratch scr[3*blockDim].
// rs/mean come from LDS tables (rsl, mml) filled when the layer's coefficients were built. Ends with a barrier.
template <class Sink = StoreSink>
__device__ __forceinline__ void out_layer_backward(const float* P, int ldp, const float* inv, const float* sh,
                                                   const float* d, const float* w3, const float* rsl,
                                                   const float* mml, int K, float* DZ, int ldz, float* scr,
                                                   float* __restrict__ gW3, float* __restrict__ gg,
                                                   float* __restrict__ gbe, Sink sink = Sink()) {
    const int nth = blockDim.x;
    int parts = 1;
    while (parts * 2 * K <= nth && parts < 8) parts *= 2;
    const int part = threadIdx.x / K, k = threadIdx.x - part * K;
    const int rows = TILE / parts;
    float dw = 0.f, dgm = 0.f, dbt = 0.f;
    if (part < parts) {
        const float wk = w3[k], iv = inv[k], s = sh[k], rs = rsl[k], mean = mml[k];
        for (int rb = part * rows; rb < (part + 1) * rows; rb += 8) {  // rows % 8 == 0; reads before compute/writes
            float pv[8], dv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pv[i] = P[(rb + i) * ldp + k], dv[i] = d[rb + i];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float p = pv[i], dr = dv[i];
                dw = fmaf(fmaf(p, iv, s), dr, dw);
                const float dy = dr * wk;
                dgm = fmaf(dy * (p - mean), rs, dgm);
                dbt += dy;
                DZ[(rb + i) * ldz + k] = (p > 0.f) ? dy * iv : 0.f;
            }
        }
    }
    if (gW3) {
        scr[threadIdx.x] = dw, scr[nth + threadIdx.x] = dgm, scr[2 * nth + threadIdx.x] = dbt;
        lds_barrier();
        if (threadIdx.x < K) {
            float a = 0.f, b2 = 0.f, c = 0.f;
            for (int q = 0; q < parts; ++q)
                a += scr[q * K + k], b2 += scr[nth + q * K + k], c += scr[2 * nth + q * K + k];
            sink.put(gW3 + k, a), sink.put(gg + k, b2), sink.put(gbe + k, c);
        }
    }
    lds_barrier();
}

// ------------------------------------------------------------------------------------------
// fused learn kernel: one agent (64-row batch) per workgroup
// ------------------------------------------------------------------------------------------
struct LearnLds {
    float *bufA, *bufB, *bufC;  // [64][ldA], [64][ldB], [64][ldB]
    float *invA, *shA;          // H1+Ha: BN coefficients of the features held in bufA
    float *invB, *shB;          // H2:    BN coefficients of the features held in bufB
    float *w3B, *rsB, *mmB;     // H2:    output-layer weights, rsqrt(var+eps) and mean of that BN layer
    float *db;                  // H2
    float *scr;                 // 3*blockDim reduction scratch
    float *sS, *sS2;            // [64][S]
    float *sAct, *sR, *sY, *sQ, *sD, *sA1, *sT, *sDa;  // [64] each
    float* red;                                         // [8]
};

__host__ __device__ inline size_t learn_lds_floats(const avd_mlp_layout& L, int nth = NTHREADS) {
    const int ldA = ld_of(L.H1 + L.Ha), ldB = ld_of(L.H2);
    return (size_t)TILE * ldA + 2 * (size_t)TILE * ldB + 2 * (L.H1 + L.Ha) + 6 * L.H2 + 3 * nth +
           2 * TILE * L.S + 8 * TILE + 8;
}

__device__ __forceinline__ LearnLds carve(float* smem, const avd_mlp_layout& L, int nth = NTHREADS) {
    LearnLds l;
    const int ldA = ld_of(L.H1 + L.Ha), ldB = ld_of(L.H2);
    float* p = smem;
    l.bufA = p, p += TILE * ldA;
    l.bufB = p, p += TILE * ldB;
    l.bufC = p, p += TILE * ldB;
    l.invA = p, p += L.H1 + L.Ha;
    l.shA = p, p += L.H1 + L.Ha;
    l.invB = p, p += L.H2;
    l.shB = p, p += L.H2;
    l.w3B = p, p += L.H2;
    l.rsB = p, p += L.H2;
    l.mmB = p, p += L.H2;
    l.db = p, p += L.H2;
    l.scr = p, p += 3 * nth;
    l.sS = p, p += TILE * L.S;
    l.sS2 = p, p += TILE * L.S;
    l.sAct = p, p += TILE;
    l.sR = p, p += TILE;
    l.sY = p, p += TILE;
    l.sQ = p, p += TILE;
    l.sD = p, p += TILE;
    l.sA1 = p, p += TILE;
    l.sT = p, p += TILE;
    l.sDa = p, p += TILE;
    l.red = p;
    return l;
}

// coefficients of the BN layer in front of a width-1 output layer + that layer's weights -> LDS (thread k < H2)
struct L2Col {
    float g, be, mm, mv, w3;
};
__device__ __forceinline__ L2Col l2_load(const float* __restrict__ g, const float* __restrict__ be,
                                         const float* __restrict__ mm, const float* __restrict__ mv,
                                         const float* __restrict__ w3, int H2, int k) {
    L2Col c = {0.f, 0.f, 0.f, 1.f, 0.f};
    if (k < H2) c.g = g[k], c.be = be[k], c.mm = mm[k], c.mv = mv[k], c.w3 = w3[k];
    return c;
}
__device__ __forceinline__ void l2_store(const L2Col& c, LearnLds& l, int H2, int k) {
    if (k < H2) {
        const float rs = 1.0f / sqrtf(c.mv + BN_EPS);
        const float iv = rs * c.g;
        l.invB[k] = iv, l.shB[k] = c.be - c.mm * iv, l.w3B[k] = c.w3, l.rsB[k] = rs, l.mmB[k] = c.mm;
    }
}

__device__ __forceinline__ float block_sum64(const float* v, float* red) {
    // sum of 64 LDS values by wave 0; result broadcast through red[0]
    if (threadIdx.x < 64) {
        float s = v[threadIdx.x];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (threadIdx.x == 0) red[0] = s;
    }
    lds_barrier();
    const float r = red[0];
    lds_barrier();
    return r;
}


// ---- batch-1 forward building blocks (mlp_rows_kernel; the next-action epilogue of learn_kernel_l<fused>) ----
// y[n] = relu(sum_k x[k]*W[k][n] + b[n]) for n < N with x in LDS (already BN'ed); 256 threads split K.
__device__ __forceinline__ void gemv_relu(const float* x, int K, const float* __restrict__ W,
                                          const float* __restrict__ b, int N, float* part, float* y) {
    const int cols = N < NTHREADS ? N : NTHREADS;
    const int ksplit = NTHREADS / cols;
    for (int n0 = 0; n0 < N; n0 += cols) {
        const int n = n0 + (threadIdx.x % cols);
        const int kh = threadIdx.x / cols;
        float acc = 0.f;
        if (kh < ksplit && n < N) {
            const int kb = (K * kh) / ksplit, ke = (K * (kh + 1)) / ksplit;
#pragma unroll 8
            for (int k = kb; k < ke; ++k) acc = fmaf(x[k], W[(long)k * N + n], acc);
        }
        part[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < cols && n < N) {
            float sum = b[n];
            for (int h = 0; h < ksplit; ++h) sum += part[h * cols + threadIdx.x];
            y[n] = fmaxf(sum, 0.f);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void bn_apply(float* y, int n, const float* __restrict__ g, const float* __restrict__ be,
                                         const float* __restrict__ mm, const float* __restrict__ mv) {
    for (int k = threadIdx.x; k < n; k += NTHREADS) {
        const float iv = (1.0f / sqrtf(mv[k] + BN_EPS)) * g[k];
        y[k] = fmaf(y[k], iv, be[k] - mm[k] * iv);
    }
}

__device__ __forceinline__ float block_dot(const float* x, const float* __restrict__ w, int wstride, int n,
                                           float* part) {
    float acc = 0.f;
    for (int k = threadIdx.x; k < n; k += NTHREADS) acc = fmaf(x[k], w[(long)k * wstride], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    const float r = part[0] + part[1] + part[2] + part[3];
    __syncthreads();
    return r;
}


// Extra arguments of the fused learn+update form (FUSED): Adam + Polyak are applied where each gradient is produced.
struct UpdArgs {
    float* theta_out;  // [n_agents][theta_size]: updated weights (theta itself stays pre-update for the whole step)
    float *m, *v;      // Adam moments, in place
    const int32_t* step;  // [n_agents] Adam iteration count AFTER this update
    float actor_lr, critic_lr, tau, omt;
    // optional: act_out[agent][0] = actor(act_x[agent * act_x_stride ..]) with the UPDATED weights (avd_learn_update_act_f32)
    const float* act_x;
    int act_x_stride;
    float* act_out;
};

// cen.hip: learn_kernel_c, the centralized framework's shapes (H1 / H2 / Ha = 320 / 160 / 64, (S, A) = (12, 3) or (20, 5)): gradients
// out (avd_learn_f32's contract) and the whole update (avd_learn_update_f32's: learn chunks and their Adam + Polyak passes on two streams)
bool cen_supports(const avd_mlp_layout* lay);
void cen_update_plan(int n_agents, int* chunk_out, int* groups_out);
int cen_launch(const avd_mlp_layout* lay, int n_agents, int set_mod, const float* theta, const float* stats, const float* theta_t,
               const float* stats_t, const float* s, const float* a, const float* r, const float* s2, float gamma, float high, float* grads,
               float* losses, void* stream);
int cen_launch_update(const avd_mlp_layout* lay, int n_agents, const float* theta, const float* stats, float* theta_out, float* theta_t,
                      float* stats_t, float* m, float* v, const int32_t* step, const float* s, const float* a, const float* r,
                      const float* s2, float gamma, float high, float actor_lr, float critic_lr, double tau, float* grads, float* losses,
                      void* stream);

// lean.hip: learn_kernel_l (two workgroups per CU; layer-1 activations recomputed on the fly). Same contract as
// fast::launch in mlp.hip; the reference widths 256/128/48 with S in {3, 4} only.
int lean_launch(const avd_mlp_layout* lay, bool fused, int n_agents, int set_mod, const float* theta, const float* stats,
                float* theta_t, float* stats_t, const float* s, const float* a, const float* r, const float* s2,
                float gamma, float high, float* grads, float* losses, UpdArgs upd, void* stream);

}  // namespace avd
