// Shared-weight-set ("wide") learner: Trainer.learn for agents that share their networks, as layer-wise bf16 MFMA
// GEMMs over ALL rows of a set at once.
//
// When interfrl averages gradients every step, the P platoons' vehicle-m agents keep identical weights (SURVEY 3.4;
// workers/trainer.py:121-128, 400-431) and the federated mean of their 64-row batch gradients IS the gradient of one
// (P x 64)-row batch: the per-agent LDS-resident kernel (mlp.hip) is then the wrong shape -- and at hidden sizes
// like BASELINE config 5 (1024) it does not fit LDS at all. Here each layer is one GEMM over N = P*64 rows per set:
//     forward   P2  = relu(C  @ (inv (.) W2) + (b2 + sh @ W2))          D[N ][H2] = A[N ][K ] . B[H2][K ]^T
//     backward  dC  = dZ2 @ W2^T, BN/ReLU backward in the epilogue      D[N ][K ] = A[N ][H2] . B[K ][H2]^T
//     weights   dW2 = inv (.) (C^T @ dZ2) + sh (x) db2                  D[K ][H2] = A[K ][N ] . B[H2][N ]^T
// -- all three in the one `A . B^T, both operands reduction-contiguous` form, because the producers of C and dZ2 also
// write their transposes. Activations and GEMM operands are bf16, accumulation / parameters / gradients f32
// (v_mfma_f32_16x16x32_bf16). First layers (K = S or A), the width-A output layer and the BN parameter gradients
// are bandwidth-bound row/column reductions.
//
// Output: the MEAN gradient per weight set in the standard slab layout (so avd_adam_polyak_f32 applies it), i.e.
// exactly what fed_sum/fed_finalize produce from per-agent gradients, up to bf16 rounding of the GEMM operands.
#include <type_traits>

#include "common.h"
#include <vector>
#include <atomic>

namespace avd {
namespace wide {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr float BN_EPS = 1e-3f;  // tf.keras BatchNormalization default epsilon (agent/model.py:28)

__host__ __device__ constexpr long rup(long x, long m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------
// D = A . B^T   A[M][K], B[Nc][K] bf16 with K contiguous (leading dimensions lda, ldb), f32 accumulate.
// 128x128 macro tile, K step 64, 4 waves as 2x2, 64x64 per wave (4x4 MFMA tiles), LDS double buffer filled through
// registers, one barrier per K step. Rows of 128 B in LDS are XOR-swizzled in 16-byte chunks so that the 16 lanes
// of a fragment read (same k chunk, 16 consecutive rows) hit 8 different bank groups.
// Operands are over-allocated to tile multiples by the caller (no bounds checks on loads); the epilogue guards.
// ------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64, GT = 256;

// The relu mask as the exact bf16 pair {1, 0} of a packed pair, two packed VALU each (inline asm: written as vector arithmetic hipcc
// makes two compares, two selects and a v_perm of it):
//   sign_mask2: of SIGNED bf16 pre-activations: x >> 15 (arithmetic, per half) is -1 / 0, times 0x3F80 plus 0x3F80 is 0 / 0x3F80 (a +0
//               counts as positive);
//   relu_mask2: of relu'd bf16 activations (non-negative int16): min(x, 1) is 0 / 1, times 0x3F80.
// The rank-one backward's producers (the actor's forward pass, the critic(s, mu) delta pass) store the mask image with them.
__device__ __forceinline__ unsigned sign_mask2(unsigned x) {
    unsigned t, m;
    asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(t) : "s"(0x000F000F), "v"(x));
    asm("v_pk_mad_u16 %0, %1, %2, %2" : "=v"(m) : "v"(t), "s"(0x3F803F80));
    return m;
}
// a packed pair of relu'd bf16 activations -> the pair c where they are positive, 0 elsewhere: min(x, 1) is 0 / 1, times the bits of c
__device__ __forceinline__ unsigned relu_select2(unsigned x, unsigned c) {
    unsigned t, m;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "s"(0x00010001), "v"(x));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(m) : "v"(t), "v"(c));
    return m;
}
// relu as ONE instruction (fmaxf quiets its operand first: two). A builtin, not inline asm: the operand is an MFMA result, and only for
// instructions it knows does hipcc keep the wait states between the MFMA and the first read of its result (asm here read stale registers)
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }
__device__ __forceinline__ unsigned relu_mask2(unsigned x) {
    unsigned t, m;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "s"(0x00010001), "v"(x));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(m) : "s"(0x3F803F80), "v"(t));
    return m;
}

struct GemmP {
    const bf16* A;
    const bf16* B;
    long lda, ldb;
    long setA, setB;  // element strides between weight sets (blockIdx.z)
    int M, Nc, K;     // K: reduction length of ONE split (multiple of BK)
    int ksplit;       // number of K splits (dW); split s covers reduction [s*K, (s+1)*K)
};

// 16 consecutive rows at one k chunk (a quarter-wave of a ds_read_b128) must cover all 64 banks: row bit 0 selects the
// 128-byte half, bits 1..3 permute the eight 16-byte chunks
__device__ __forceinline__ int swz(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3); }

// 64 KiB of LDS per workgroup = 2 workgroups (2 waves per SIMD) per CU: tell the register allocator so (it otherwise
// aims at 4 waves per SIMD and spills the staging registers to scratch)
template <class Epi>
__global__ __launch_bounds__(GT) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bt_kernel(GemmP p, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* sA = (bf16*)smem_raw;       // [2][BM*BK]
    bf16* sB = sA + 2 * BM * BK;      // [2][BN*BK]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order. Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the tiles that
    // share an operand -- the column tiles of one row tile (forward, dX), all tiles of one reduction chunk (dW) -- are
    // given ids that are congruent mod 8 and consecutive in time: 8 such groups are in flight, one per XCD.
    const int ncol = (p.Nc + BN - 1) / BN, nrow = (p.M + BM - 1) / BM;
    const int gsz = (p.ksplit > 1) ? nrow * ncol : ncol;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int grp = (jj / gsz) * 8 + xcd, tin = jj - (jj / gsz) * gsz;
    const int set = blockIdx.z;
    int ks = 0, rt = grp, ct = tin;
    if (p.ksplit > 1) ks = grp, rt = tin / ncol, ct = tin - rt * ncol;
    if (ks >= p.ksplit || rt >= nrow) return;
    const int row0 = rt * BM, col0 = ct * BN;
    const bf16* A = p.A + (long)set * p.setA + (long)row0 * p.lda + (long)ks * p.K;
    const bf16* B = p.B + (long)set * p.setB + (long)col0 * p.ldb + (long)ks * p.K;

    // Operand tiles go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no LDS store
    // instructions). One wave instruction fills 1024 consecutive LDS bytes = 8 tile rows in order, lane l -> row l / 8,
    // 16-byte slot l % 8; the swizzle is therefore applied on the GLOBAL side: the lane fetches the k chunk that
    // belongs in its slot. Wave w issues instructions i = 0..3 for rows 32w + 8i .. + 8 of each operand.
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int l8 = lane >> 3, slot = lane & 7;
    const int ch0 = slot ^ (lane >> 4), ch1 = slot ^ (4 + (lane >> 4));  // ((row >> 1) & 7) for even / odd i
    const bf16* ga = A + (long)(32 * wave + l8) * p.lda;
    const bf16* gb = B + (long)(32 * wave + l8) * p.ldb;
    const long sa8 = 8 * p.lda, sb8 = 8 * p.ldb;
    bf16* la = sA + (32 * wave) * BK;
    bf16* lb = sB + (32 * wave) * BK;
#define WIDE_GLDS(buf, k0)                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                                  \
        const int ch_ = ((i_ & 1) ? ch1 : ch0) * 8 + (k0);                                                             \
        __builtin_amdgcn_global_load_lds((gptr_t)(ga + i_ * sa8 + ch_), (lptr_t)(la + (buf) * BM * BK + i_ * 8 * BK), 16, 0, 0); \
        __builtin_amdgcn_global_load_lds((gptr_t)(gb + i_ * sb8 + ch_), (lptr_t)(lb + (buf) * BN * BK + i_ * 8 * BK), 16, 0, 0); \
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    WIDE_GLDS(0, 0)
    __syncthreads();  // (drains vmcnt: the tile has landed)
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const int kn = min(kt + 1, nk - 1) * BK;  // the last step re-loads its own tile instead of branching
        // buf ^ 1 was last read in step kt - 1, which every wave has left (barrier below)
        WIDE_GLDS(buf ^ 1, kn)
        const bf16* a_s = sA + buf * BM * BK;
        const bf16* b_s = sB + buf * BN * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ch = lg + 4 * kk;
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(a_s + swz(wm * 64 + 16 * i + lr, ch));
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(b_s + swz(wn * 64 + 16 * j + lr, ch));
            // operands swapped: the tile comes out transposed, i.e. a lane's 4 accumulator registers are 4
            // CONSECUTIVE OUTPUT COLUMNS of one row (8-byte bf16 / 16-byte f32 stores in the epilogues)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();  // vmcnt(0) + barrier: the next tile is in LDS, this one is free
    }
#undef WIDE_GLDS
    // acc[i][j][r]: row = row0 + wm*64 + 16i + lr, col = col0 + wn*64 + 16j + 4lg + r
    epi.template operator()<4>(acc, row0 + wm * 64 + lr, col0 + wn * 64 + 4 * lg, set, ks, p.M, p.Nc);
}


// ------------------------------------------------------------------------------------------
// The same GEMM for large problems: 256x256 macro tile, 8 waves as 2 (M) x 4 (N), 128x64 per wave. The 128^2 kernel
// above moves 1 operand byte per 64 FLOP from L2 into LDS and saturates the L2 -> LDS path at ~0.7 PFLOP/s; this tile
// halves the bytes per FLOP. One workgroup per CU (96 KiB of LDS), so nothing else hides a drained pipeline: K steps
// of 32 through THREE LDS stages, the loads of two steps stay in flight across the barrier (counted vmcnt, raw
// s_barrier -- __syncthreads() would wait for vmcnt(0)).
// LDS rows are 64 B (4 chunks of 16 B); chunk' = chunk ^ 2*bit2(row) is conflict-free for the (non-contiguous) 16-lane
// groups ds_read_b128 is serviced in (MI355X_MICROARCH.md, LDS table; checked by enumeration). One global_load_lds wave
// instruction fills 16 rows.
// ------------------------------------------------------------------------------------------
constexpr int TM = 256, TK = 32, NSTG = 3;
__device__ __forceinline__ int swz32(int row, int chunk) { return row * TK + ((chunk ^ (((row >> 2) & 1) << 1)) << 3); }

// TNV = 256: 8 waves, one workgroup per CU (96 KiB). TNV = 128: 4 waves, 72 KiB, two independent workgroups per CU.
template <class Epi, int TNV>
__global__ __launch_bounds__(TNV * 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bt256_kernel(GemmP p, Epi epi) {
    constexpr int NWV = TNV / 32;                             // waves: 2 (M) x TNV/64 (N)
    constexpr int IA = TM / (16 * NWV), IB = TNV / (16 * NWV);  // global_load_lds instructions per wave, stage and operand
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* sA = (bf16*)smem_raw;         // [NSTG][TM*TK]
    bf16* sB = sA + NSTG * TM * TK;     // [NSTG][TNV*TK]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    const int wm = wave / (TNV / 64), wn = wave % (TNV / 64);
    const int ncol = (p.Nc + TNV - 1) / TNV, nrow = (p.M + TM - 1) / TM;
    const int gsz = (p.ksplit > 1) ? nrow * ncol : ncol;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int grp = (jj / gsz) * 8 + xcd, tin = jj - (jj / gsz) * gsz;
    const int set = blockIdx.z;
    int ks = 0, rt = grp, ct = tin;
    if (p.ksplit > 1) ks = grp, rt = tin / ncol, ct = tin - rt * ncol;
    if (ks >= p.ksplit || rt >= nrow) return;
    const int row0 = rt * TM, col0 = ct * TNV;
    const bf16* A = p.A + (long)set * p.setA + (long)row0 * p.lda + (long)ks * p.K;
    const bf16* B = p.B + (long)set * p.setB + (long)col0 * p.ldb + (long)ks * p.K;

    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // wave w fills rows [16 IA w, 16 IA (w+1)) of A and [16 IB w, ..) of B, 16 rows per instruction; lane -> row l / 4, slot l % 4
    const int l4 = lane >> 2, slot = lane & 3;
    const int chs = (slot ^ (((l4 >> 2) & 1) << 1)) * 8;  // bit 2 of the row is bit 2 of l4 (row offsets are multiples of 16)
    const bf16* ga = A + (long)(16 * IA * wave + l4) * p.lda + chs;
    const bf16* gb = B + (long)(16 * IB * wave + l4) * p.ldb + chs;
    const long sa16 = 16 * p.lda, sb16 = 16 * p.ldb;
    bf16* la = sA + (16 * IA * wave) * TK;
    bf16* lb = sB + (16 * IB * wave) * TK;
#define WIDE_GLDS3(stg, k0)                                                                                                       \
    {                                                                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < IA; ++i_)                                                                         \
            __builtin_amdgcn_global_load_lds((gptr_t)(ga + i_ * sa16 + (k0)), (lptr_t)(la + (stg) * TM * TK + i_ * 16 * TK), 16, 0, 0); \
        _Pragma("unroll") for (int i_ = 0; i_ < IB; ++i_)                                                                         \
            __builtin_amdgcn_global_load_lds((gptr_t)(gb + i_ * sb16 + (k0)), (lptr_t)(lb + (stg) * TNV * TK + i_ * 16 * TK), 16, 0, 0); \
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / TK;
    WIDE_GLDS3(0, 0)
    WIDE_GLDS3(1, min(1, nk - 1) * TK)
    if constexpr (NWV == 8) {
        // PING-PONG (r03): waves 0..3 (one per SIMD) and waves 4..7 alternate between "prepare" (the 12 fragment reads of a K
        // step, the refill DMA of the stage both groups finished with) and "multiply" (its 32 MFMAs at raised priority), a barrier
        // after each phase, the second group one phase behind: one wave per SIMD multiplies while the other prepares. With one
        // barrier per K step and all eight waves in the same program the two waves of a SIMD read together and multiplied together.
        // Every wave waits for all but its youngest step's loads before every barrier: whatever is read next phase has landed.
        const int grp = wave >> 2;
        __builtin_amdgcn_s_waitcnt(0x0F70 | (IA + IB));
        __builtin_amdgcn_s_barrier();
        if (grp == 1) {
            __builtin_amdgcn_s_waitcnt(0x0F70 | (IA + IB));
            __builtin_amdgcn_s_barrier();
        }
        int stg = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const bf16* a_s = sA + stg * TM * TK;
            const bf16* b_s = sB + stg * TNV * TK;
            bf16x8 af[8], bfr[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *(const bf16x8*)(a_s + swz32(wm * 128 + 16 * i + lr, lg));
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(b_s + swz32(wn * 64 + 16 * j + lr, lg));
            const int nstg = (stg + 2 >= NSTG) ? stg + 2 - NSTG : stg + 2;
            WIDE_GLDS3(nstg, min(kt + 2, nk - 1) * TK)  // into the stage of step kt - 1; (tail: harmless re-loads)
            __builtin_amdgcn_s_waitcnt(0x0070 | (IA + IB));  // vmcnt(one step) lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0F70 | (IA + IB));
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stg = (stg + 1 == NSTG) ? 0 : stg + 1;
        }
        if (grp == 0) {
            __builtin_amdgcn_s_waitcnt(0x0F70 | (IA + IB));
            __builtin_amdgcn_s_barrier();
        }
    } else {
    int stg = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // step kt's IA + IB loads are the oldest in flight; those of step kt+1 may stay in flight
        __builtin_amdgcn_s_waitcnt(0x0F70 | (IA + IB));  // vmcnt(IA + IB), expcnt/lgkmcnt untouched
        __builtin_amdgcn_s_barrier();                     // every wave's share of step kt has landed; everyone has left step kt-1
        const int nstg = (stg + 2 >= NSTG) ? stg + 2 - NSTG : stg + 2;
        WIDE_GLDS3(nstg, min(kt + 2, nk - 1) * TK)  // into the stage read in step kt-1; (tail: harmless re-loads)
        const bf16* a_s = sA + stg * TM * TK;
        const bf16* b_s = sB + stg * TNV * TK;
        bf16x8 af[8], bfr[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *(const bf16x8*)(a_s + swz32(wm * 128 + 16 * i + lr, lg));
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(b_s + swz32(wn * 64 + 16 * j + lr, lg));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        stg = (stg + 1 == NSTG) ? 0 : stg + 1;
    }
    }
#undef WIDE_GLDS3
    __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the tail re-loads before the workgroup's LDS is released
    epi.template operator()<8>(acc, row0 + wm * 128 + lr, col0 + wn * 64 + 4 * lg, set, ks, p.M, p.Nc);
}

// ---- epilogues: operator()(acc, row_base, col_base, set, ks, M, Nc); element (i, j, r) -> (row_base + 16i, col_base + 16j + r)
struct EpiStoreF32 {  // plain D (tests)
    float* D;
    long ldd, setD;
    template <int MI>
    __device__ void operator()(f32x4 (&acc)[MI][4], int rb, int cb, int set, int, int M, int Nc) const {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = rb + 16 * i, col = cb + 16 * j;
                if (row < M && col < Nc) *(f32x4*)(D + (long)set * setD + (long)row * ldd + col) = acc[i][j];
            }
    }
};

struct EpiFwd {  // out = relu(acc + bias[col]) as bf16; optionally the width-1 output layer on top: z[row] += out[row][:] . cf[:]
    bf16* out;
    long ldo, setO;
    const float* bias;  // [sets][setBias]
    long setBias;
    const float* cf;    // [sets][setBias] output-layer coefficients (BN folded) or NULL
    float* z;           // [sets][setZ], pre-filled with the constant term; f32 atomics (16 partial sums per row at H2 = 1024)
    long setZ;
    template <int MI>
    __device__ void operator()(f32x4 (&acc)[MI][4], int rb, int cb, int set, int, int M, int Nc) const {
        const float* b = bias + (long)set * setBias;
        float dot[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) dot[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = cb + 16 * j;
            if (col >= Nc) continue;
            const f32x4 bv = *(const f32x4*)(b + col);
            f32x4 cv = {0.f, 0.f, 0.f, 0.f};
            if (cf) cv = *(const f32x4*)(cf + (long)set * setBias + col);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = rb + 16 * i;
                if (row >= M) continue;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = (bf16)fmaxf(acc[i][j][r] + bv[r], 0.f);
                    dot[i] = fmaf((float)o[r], cv[r], dot[i]);  // the stored (rounded) activation, as the backward pass sees it
                }
                *(bf16x4*)(out + (long)set * setO + (long)row * ldo + col) = o;
            }
        }
        if (cf) {  // lanes lr, lr+16, lr+32, lr+48 hold the same rows: one atomic per row per wave
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                float d = dot[i];
                d += __shfl_xor(d, 16);
                d += __shfl_xor(d, 32);
                const int row = rb + 16 * i;
                if ((threadIdx.x & 63) < 16 && row < M) atomicAdd(z + (long)set * setZ + row, d);
            }
        }
    }
};

constexpr int NSLICE = 32;
// BatchNorm(inference form) + ReLU backward of the layer below: dy = acc; dz = dy * inv * (p > 0);
// dgamma[c] += sum_rows dy (p - mean) rs; dbeta[c] += sum_rows dy.   Columns are offset by c_off in all tables.
struct EpiDx {
    const bf16* P;  // activations of the layer below [rows][ldp], same rows/cols as the output
    bf16* dZ;
    long ldp, setP;
    const float *inv, *rs, *mean;  // [sets][setTab] tables over the concatenated features
    float *dgamma, *dbeta;         // [NSLICE][sets][setTab] accumulators (or NULL): row tile t adds into slice t % NSLICE
    long setTab, sliceStride;      // (thousands of row tiles adding into one address would serialise)
    int c_off;
    const float* rsc;              // [sets][setR] or NULL: row factor of the accumulator (rank-one form: A = the relu mask, dy = d[row] acc)
    long setR;
    template <int MI>
    __device__ void operator()(f32x4 (&acc)[MI][4], int rb, int cb, int set, int, int M, int Nc) const {
        const long tb = (long)set * setTab + c_off;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = cb + 16 * j;
            const bool cok = col < Nc;
            f32x4 iv = {0, 0, 0, 0}, rsv = iv, mv = iv;
            if (cok) iv = *(const f32x4*)(inv + tb + col), rsv = *(const f32x4*)(rs + tb + col), mv = *(const f32x4*)(mean + tb + col);
            float sg[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = rb + 16 * i;
                if (!cok || row >= M) continue;
                const long o = (long)set * setP + (long)row * ldp + c_off + col;
                const bf16x4 pv = *(const bf16x4*)(P + o);
                const float rf = rsc ? rsc[(long)set * setR + row] : 1.0f;
                bf16x4 dz;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dy = acc[i][j][r] * rf, pp = (float)pv[r];
                    sg[r] = fmaf(dy * (pp - mv[r]), rsv[r], sg[r]);
                    sb[r] += dy;
                    dz[r] = (bf16)(pp > 0.f ? dy * iv[r] : 0.f);
                }
                *(bf16x4*)(dZ + o) = dz;
            }
            if (dgamma) {  // reduce over the 16 lanes (rows) of each lane group, one atomic per column per wave
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) sg[r] += __shfl_xor(sg[r], o), sb[r] += __shfl_xor(sb[r], o);
                }
                if ((threadIdx.x & 15) == 0 && cok) {
                    const long so = (long)((rb >> 7) & (NSLICE - 1)) * sliceStride + tb + col;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        atomicAdd(dgamma + so + r, sg[r]);
                        atomicAdd(dbeta + so + r, sb[r]);
                    }
                }
            }
        }
    }
};

// dW[k][j] += inv[k] * acc (+ sh[k] * db[j] from split 0): rows = features k, cols = outputs j, f32 atomics
struct EpiDw {
    float* dW;  // [sets][setW] + offset of this matrix
    long ldw, setW;
    const float *inv, *sh;  // [sets][setTab]
    const float* db;        // [sets][setDb]
    long setTab, setDb;
    float scale;
    template <int MI>
    __device__ void operator()(f32x4 (&acc)[MI][4], int rb, int cb, int set, int ks, int M, int Nc) const {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = rb + 16 * i;
            if (row >= M) continue;
            const float iv = inv[(long)set * setTab + row] * scale, sf = (ks == 0) ? sh[(long)set * setTab + row] * scale : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = cb + 16 * j;
                if (col >= Nc) continue;
                const f32x4 dbv = *(const f32x4*)(db + (long)set * setDb + col);
                float* o = dW + (long)set * setW + (long)row * ldw + col;
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(o + r, fmaf(iv, acc[i][j][r], sf * dbv[r]));
            }
        }
    }
};

template <class Epi>
static int launch_gemm(const GemmP& p, const Epi& e, int n_sets, hipStream_t st, const char* who) {
    constexpr size_t lds128 = 2 * (BM + BN) * BK * sizeof(bf16), lds256 = NSTG * (TM + 256) * TK * sizeof(bf16),
                     lds256x128 = NSTG * (TM + 128) * TK * sizeof(bf16);
    // the > 64 KB dynamic-LDS opt-in, once per DEVICE of this process and per epilogue (the attribute belongs to the device's copy of
    // the function); a failure is reported, not swallowed. (Two host threads racing here both set the same attribute: harmless.)
    static std::atomic<unsigned long long> attr_done{0};  // bit = device ordinal
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !((attr_done.load(std::memory_order_acquire) >> dev) & 1ull)) {
        hipError_t err = hipFuncSetAttribute((const void*)gemm_bt_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
        if (err == hipSuccess)
            err = hipFuncSetAttribute((const void*)gemm_bt256_kernel<Epi, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
        if (err == hipSuccess)
            err = hipFuncSetAttribute((const void*)gemm_bt256_kernel<Epi, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256x128);
        if (err != hipSuccess) {
            set_error("%s: hipFuncSetAttribute(dynamic LDS %zu B) on device %d: %s", who, lds256, dev, hipGetErrorString(err));
            return AVD_E_LAUNCH;
        }
        if (dev >= 0 && dev < 64) attr_done.fetch_or(1ull << dev, std::memory_order_release);
    }
    static const char* force = AVD_DIAG_ENV("GEMM_TILE");  // diagnostics: "128", "256x128", "256"
    int tile = (p.M >= 2 * TM && p.Nc >= 2 * 256) ? 256 : 128;  // large problems: 256^2 tiles
    if (force) tile = !strcmp(force, "128") ? 128 : (!strcmp(force, "256x128") ? 192 : 256);
    if (tile != 128 && (p.M < TM || p.K % TK)) tile = 128;
    const int tm = tile == 128 ? BM : TM, tn = tile == 256 ? 256 : 128;
    const long ncol = rup(p.Nc, tn) / tn, nrow = rup(p.M, tm) / tm;
    const long gsz = p.ksplit > 1 ? nrow * ncol : ncol, groups = p.ksplit > 1 ? p.ksplit : nrow;
    dim3 grid((unsigned)(gsz * rup(groups, 8)), 1, (unsigned)n_sets);
    if (tile == 256)
        hipLaunchKernelGGL((gemm_bt256_kernel<Epi, 256>), grid, dim3(512), lds256, st, p, e);
    else if (tile == 192)
        hipLaunchKernelGGL((gemm_bt256_kernel<Epi, 128>), grid, dim3(256), lds256x128, st, p, e);
    else
        hipLaunchKernelGGL((gemm_bt_kernel<Epi>), grid, dim3(GT), lds128, st, p, e);
    return check_launch(who);
}

// ------------------------------------------------------------------------------------------
// bandwidth-bound pieces. Rows of a set: n in [0, Ns); buffers hold Np = rup(Ns, 128) rows, rows >= Ns are ZERO
// wherever a buffer feeds a reduction over rows (the transposed copies).
// ------------------------------------------------------------------------------------------
struct Dims {
    int S, H1, H2, Ha, KC, KCp;  // KCp = rup(H1 + Ha, 64): row length of the first-layer activation buffers
    int Ns, Np, n_sets;
    long theta_size, stats_size;
};

// BN tables of `len` features: inv = g / sqrt(var + eps), sh = be - mean * inv, rs = 1/sqrt(var + eps), mean
__global__ void bn_tables_kernel(const float* th, const float* st, long set_th, long set_st, int g_off, int be_off,
                                 int mm_off, int mv_off, int len, float* inv, float* sh, float* rs, float* mean, long set_tab,
                                 int t_off, int pad_to) {
    const int set = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= pad_to) return;
    float iv = 0.f, s = 0.f, r = 0.f, m = 0.f;
    if (k < len) {
        const float* t = th + (long)set * set_th;
        const float* q = st + (long)set * set_st;
        r = 1.0f / sqrtf(q[mv_off + k] + BN_EPS);
        iv = r * t[g_off + k];
        m = q[mm_off + k];
        s = t[be_off + k] - m * iv;
    }
    const long o = (long)set * set_tab + t_off + k;
    inv[o] = iv, sh[o] = s, rs[o] = r, mean[o] = m;
}

// Layer-2 weights W[K][N] (f32, Keras layout) -> WT[N][Kp] = bf16(inv[k] * W[k][n]) (forward B operand) and
// Wn[K][N] = bf16(W[k][n]) (dX B operand: rows = features, reduction over n). 32x32 LDS transpose tiles.
// cfn != NULL: Wn[k][n] = bf16(cfn[n] * W[k][n]) -- the output layer's coefficients folded into the input gradient's operand (rank-one form)
__global__ void prep_w2_kernel(const float* th, long set_th, int w_off, int K, int N, int Kp, const float* inv, long set_tab,
                               bf16* WT, long set_wt, bf16* Wn, long set_wn, int perm, const float* cfn) {
    __shared__ float tile[32][33];
    const int set = blockIdx.z, k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const float* W = th + (long)set * set_th + w_off;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, n = n0 + tx;
        float w = 0.f;
        if (k < K && n < N) w = W[(long)k * N + n];
        tile[r][tx] = w;
        if (Wn && k < K && n < N) Wn[(long)set * set_wn + (long)k * N + n] = (bf16)(cfn ? w * cfn[(long)set * N + n] : w);
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        if (n < N && k < Kp) {
            const float iv = (k < K) ? inv[(long)set * set_tab + k] : 0.f;
            // perm: bits 2 and 3 of k swapped (the k order of fw::fwd_gen_kernel's generated operand)
            const int kd = perm ? ((k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1)) : k;
            WT[(long)set * set_wt + (long)n * Kp + kd] = (bf16)(tile[tx][r] * iv);
        }
    }
}

// bias2[n] = b2[n] + sum_k sh[k] * W[k][n]: one block per 64 columns, 16 k-groups of 64 threads (coalesced rows of W), LDS reduction
__global__ __launch_bounds__(1024) void bias2_kernel(const float* th, long set_th, int w_off, int b_off, int K, int N, const float* sh,
                                                     long set_tab, float* bias, long set_bias) {
    __shared__ float part[16][64];
    const int set = blockIdx.y, c = threadIdx.x & 63, g = threadIdx.x >> 6, n = blockIdx.x * 64 + c;
    const float* t = th + (long)set * set_th;
    float acc = 0.f;
    if (n < N)
        for (int k = g; k < K; k += 16) acc = fmaf(sh[(long)set * set_tab + k], t[w_off + (long)k * N + n], acc);
    part[g][c] = acc;
    __syncthreads();
    if (g == 0 && n < N) {
        float sum = t[b_off + n];
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += part[i][c];
        bias[(long)set * set_bias + n] = sum;
    }
}

// Rank-one backward, after fw::dw_gen_kernel: the gradient slab holds G[k][n] = inv[k] sum_rows d y1[k] mask[n], `cs` holds
// S2[n] = sum_rows d mask[n]. With relu(z2) = mask (y1 . (inv (.) W2) + bias2):
//   u[n] = sum_rows d relu(z2)[n] = sum_k W2[k][n] G[k][n] + bias2[n] S2[n]
//   dW2[k][n] = cf[n] (G[k][n] + sh[k] S2[n]),   db2[n] = cf[n] S2[n]  (written over S2)
// One block per 64 columns, 16 k-groups of 64 threads (coalesced rows), LDS reduction -- bias2_kernel's shape.
__global__ __launch_bounds__(1024) void w2_post_kernel(const float* th, long set_th, int w_off, int K, int N, const float* sh, long set_tab,
                                                       const float* bias2, const float* cf, float* g, long set_g, float* u, float* cs,
                                                       long set_u) {
    __shared__ float part[16][64];
    const int set = blockIdx.y, c = threadIdx.x & 63, gq = threadIdx.x >> 6, n = blockIdx.x * 64 + c;
    const float* W = th + (long)set * set_th + w_off;
    float* G = g + (long)set * set_g + w_off;
    const float s2 = n < N ? cs[(long)set * set_u + n] : 0.f, cfv = n < N ? cf[(long)set * N + n] : 0.f;
    float acc = 0.f;
    if (n < N)
        for (int k = gq; k < K; k += 16) {
            const long o = (long)k * N + n;
            const float gv = G[o];
            acc = fmaf(W[o], gv, acc);
            G[o] = cfv * fmaf(sh[(long)set * set_tab + k], s2, gv);
        }
    part[gq][c] = acc;
    __syncthreads();  // (every thread has read S2 before thread group 0 overwrites it)
    if (gq == 0 && n < N) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += part[i][c];
        u[(long)set * set_u + n] = fmaf(bias2[(long)set * N + n], s2, sum);
        cs[(long)set * set_u + n] = cfv * s2;
    }
}

// First layer of a branch: C[n][c0 + k] = bf16(relu(sum_j X[n][j] W[j][k] + b[k])) for k < H (pad columns up to Hpad get 0),
// and CT[c0 + k][n] (or NULL). X: f32 [sets][Ns][KIN]. One block = 64 rows x 64 columns.
template <int KIN>
__global__ __launch_bounds__(256) void l1_fwd_kernel(const float* X, long set_x, const float* th, long set_th, int w_off,
                                                      int b_off, int H, int Hpad, int c0, int Ns, int Np, bf16* C,
                                                      long ldc, long set_c, bf16* CT, long ldct, long set_ct) {
    __shared__ float sx[64][KIN];
    __shared__ bf16 so[64][66];
    const int set = blockIdx.z, n0 = blockIdx.y * 64, k0 = blockIdx.x * 64, tid = threadIdx.x;
    const float* t = th + (long)set * set_th;
    for (int i = tid; i < 64 * KIN; i += 256) {
        const int n = n0 + i / KIN;
        sx[i / KIN][i % KIN] = (n < Ns) ? X[(long)set * set_x + (long)n * KIN + (i % KIN)] : 0.f;
    }
    __syncthreads();
    const int kc = tid & 63, k = k0 + kc;
    float w[KIN], b = 0.f;
#pragma unroll
    for (int j = 0; j < KIN; ++j) w[j] = (k < H) ? t[w_off + (long)j * H + k] : 0.f;
    if (k < H) b = t[b_off + k];
    for (int r = tid >> 6; r < 64; r += 4) {
        float acc = b;
#pragma unroll
        for (int j = 0; j < KIN; ++j) acc = fmaf(sx[r][j], w[j], acc);
        const int n = n0 + r;
        const bf16 o = (bf16)((k < H && n < Ns) ? fmaxf(acc, 0.f) : 0.f);
        if (k < Hpad && n < Np) C[(long)set * set_c + (long)n * ldc + c0 + k] = o;
        so[r][kc] = o;
    }
    if (CT) {
        __syncthreads();
        const int nr = tid & 63;
        for (int c = tid >> 6; c < 64; c += 4) {
            const int kk = k0 + c, n = n0 + nr;
            if (kk < Hpad && n < Np) CT[(long)set * set_ct + (long)(c0 + kk) * ldct + n] = so[nr][c];
        }
    }
}

// Output layer, width 1: z[n] = sum_k P[n][k] * c[k] + c0  with c = inv (.) w3 and c0 = b3 + sh . w3 (BN folded).
// mode 0: out[n] = z. mode 1 (actor): out[n] = tanh(z) * high, tout[n] = tanh(z).
// One wave per row, 4 rows per block.
__global__ __launch_bounds__(256) void row_dot_kernel(const bf16* P, long ldp, long set_p, int c_off, int K, const float* cvec,
                                                       long set_c, const float* c0v, int Ns, int mode, float high, float* out,
                                                       float* tout, long set_o) {
    const int set = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= Ns) return;
    const bf16* row = P + (long)set * set_p + (long)n * ldp + c_off;
    const float* c = cvec + (long)set * set_c;
    float acc = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const bf16x4 v = *(const bf16x4*)(row + k);
        const f32x4 cv = *(const f32x4*)(c + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fmaf((float)v[e], cv[e], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        const float z = acc + (c0v ? c0v[set] : 0.f);
        if (mode == 0) {
            out[(long)set * set_o + n] = z;
        } else {
            const float t = tanhf(z);
            out[(long)set * set_o + n] = t * high;
            tout[(long)set * set_o + n] = t;
        }
    }
}

// Output layer backward + the ReLU/BN below it: dZ2[n][k] = (P2[n][k] > 0) ? d[n] * c[k] : 0 with c = w3 (.) inv3
// (also transposed), u[k] += sum_n P2[n][k] d[n], cs[k] += sum_n dZ2[n][k].  Block = 64 rows x 64 columns, 8-byte accesses.
__global__ __launch_bounds__(256) void out_bwd_kernel(const bf16* P2, long ldp, long set_p, const float* d, long set_d,
                                                       const float* cvec, long set_c, int H2, int Ns, int Np, bf16* dZ,
                                                       bf16* dZT, long ldt, long set_t, float* u, float* cs, long set_u, int perm) {
    __shared__ bf16 so[64][68];
    __shared__ float sd[64];
    __shared__ float red[2][16][64];
    const int set = blockIdx.z, n0 = blockIdx.y * 64, k0 = blockIdx.x * 64, tid = threadIdx.x;
    if (tid < 64) sd[tid] = (n0 + tid < Ns) ? d[(long)set * set_d + n0 + tid] : 0.f;
    __syncthreads();
    const int cg = tid & 15, rg = tid >> 4, k = k0 + 4 * cg;  // H2 % 64 == 0: k + 3 < H2
    const f32x4 c = *(const f32x4*)(cvec + (long)set * set_c + k);
    float su[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = rg + 16 * rr, n = n0 + r;
        bf16x4 pv = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, o;
        if (n < Ns) pv = *(const bf16x4*)(P2 + (long)set * set_p + (long)n * ldp + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float p = fmaxf((float)pv[e], 0.f);  // (the fused forward may leave the sign on: relu commutes with the bf16 rounding)
            const float dz = (p > 0.f) ? sd[r] * c[e] : 0.f;
            su[e] = fmaf(p, sd[r], su[e]);
            sc[e] += dz;
            o[e] = (bf16)dz;
        }
        if (n < Np) *(bf16x4*)(dZ + (long)set * set_p + (long)n * ldp + k) = o;
        *(bf16x4*)(&so[r][4 * cg]) = o;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[0][rg][4 * cg + e] = su[e], red[1][rg][4 * cg + e] = sc[e];
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, kc = tid & 63;
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += red[which][g][kc];
        float* dst = which ? cs : u;
        if (dst) atomicAdd(dst + (long)set * set_u + k0 + kc, s);
    }
    if (dZT) {
        const int ng = tid & 15;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            // perm: rows in the k order of fw::dw_gen_kernel's generated operand (bits 2 and 3 of n swapped: the second and third
            // group of four of every 16 rows trade places)
            const int q4 = ng & 3, qs = perm ? ((q4 == 1) ? 2 : (q4 == 2) ? 1 : q4) : q4;
            const int cidx = (tid >> 4) + 16 * cc, n = n0 + 16 * (ng >> 2) + 4 * qs;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = so[4 * ng + e][cidx];  // (rows 4 ng .. 4 ng + 3 of the block, stored at position n)
            if (n < Np) *(bf16x4*)(dZT + (long)set * set_t + (long)(k0 + cidx) * ldt + n) = o;
        }
    }
}

// First-layer gradients: dW[j][k] += sum_n X[n][j] dZ[n][c0 + k], db[k] += sum_n dZ[n][c0 + k] (scaled).
// Block = 64 columns x `rows_per_block` rows; a thread owns 4 consecutive columns (8-byte loads) of every 16th row.
template <int KIN>
__global__ __launch_bounds__(256) void l1_grads_kernel(const float* X, long set_x, const bf16* dZ, long ldz, long set_z, int c0,
                                                        int H, int Ns, int rows_per_block, float scale, float* g, long set_g,
                                                        int w_off, int b_off) {
    __shared__ float red[16][64][KIN + 1];
    const int set = blockIdx.z, cg = threadIdx.x & 15, rg = threadIdx.x >> 4, k = blockIdx.x * 64 + 4 * cg;
    const int nb = blockIdx.y * rows_per_block, ne = min(nb + rows_per_block, Ns);
    float acc[4][KIN + 1];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j <= KIN; ++j) acc[e][j] = 0.f;
    if (k < H) {  // H % 16 == 0: all 4 columns are valid
        for (int n = nb + rg; n < ne; n += 16) {
            const bf16x4 dv = *(const bf16x4*)(dZ + (long)set * set_z + (long)n * ldz + c0 + k);
            const float* x = X + (long)set * set_x + (long)n * KIN;
            float xv[KIN];
#pragma unroll
            for (int j = 0; j < KIN; ++j) xv[j] = x[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dz = (float)dv[e];
#pragma unroll
                for (int j = 0; j < KIN; ++j) acc[e][j] = fmaf(xv[j], dz, acc[e][j]);
                acc[e][KIN] += dz;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j <= KIN; ++j) red[rg][4 * cg + e][j] = acc[e][j];
    __syncthreads();
    const int c = threadIdx.x & 63, jq = threadIdx.x >> 6;  // 4 threads per column share the KIN + 1 outputs
    const int kk = blockIdx.x * 64 + c;
    if (kk < H) {
        float* gs = g + (long)set * set_g;
        for (int j = jq; j <= KIN; j += 4) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += red[q][c][j];
            atomicAdd(j < KIN ? gs + w_off + (long)j * H + kk : gs + b_off + kk, sum * scale);
        }
    }
}

// elementwise glue over rows -----------------------------------------------------------------
// mode 0: y = r + gamma * q            (TD target, no done mask: workers/trainer.py:494)
// mode 1: d = 2 (q - y) / N, acc[set][0] += (y - q)^2, acc[set][1] += d      (critic loss seed)
// mode 2: d = -1 / N,        acc[set][2] += q                                   (actor loss seed)
// mode 3: d = da * high * (1 - t^2), acc[set][3] += d                           (through tanh * high)
// z[set][n] = c0[set] (the output layer's constant term; the forward GEMM's epilogue adds the dot products)
__global__ void fill_rows_kernel(float* z, long set_z, const float* c0, int Np) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < Np) z[(long)blockIdx.y * set_z + n] = c0[blockIdx.y];
}
// actor head: a = tanh(z) * high, t = tanh(z)
__global__ void tanh_rows_kernel(const float* z, long set_z, int Ns, float high, float* a, float* t) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= Ns) return;
    const long o = (long)blockIdx.y * set_z + n;
    const float th = tanhf(z[o]);
    a[o] = th * high, t[o] = th;
}

constexpr int ROWS_PER_BLOCK = 256 * 16;  // (a block walks 16 x 256 rows: its two atomics meet 1/16 as many others on the set's address)
__global__ __launch_bounds__(256) void rows_kernel(int mode, int Ns, long set_o, const float* q, const float* y_or_t,
                                                    const float* r_or_da, float gamma_or_high, float* out, float* acc,
                                                    const float* row_weight) {
    const int set = blockIdx.y;
    float a0 = 0.f, a1 = 0.f;
    for (int i = 0; i < ROWS_PER_BLOCK / 256; ++i) {
        const int n = blockIdx.x * ROWS_PER_BLOCK + i * 256 + threadIdx.x;
        if (n >= Ns) break;
        const long o = (long)set * set_o + n;
        // weighted federated mean (src/server/federated.py:99-118): each row's loss seed carries its platoon's weight
        const float rw = row_weight ? row_weight[(long)set * Ns + n] : 1.0f;
        if (mode == 0) {
            out[o] = fmaf(gamma_or_high, q[o], r_or_da[o]);
        } else if (mode == 1) {
            const float e = y_or_t[o] - q[o], d = -2.0f * e * rw / (float)Ns;
            out[o] = d, a0 += e * e, a1 += d;
        } else if (mode == 2) {
            out[o] = -rw / (float)Ns, a0 += q[o];
        } else {
            const float t = y_or_t[o], d = r_or_da[o] * gamma_or_high * (1.0f - t * t);
            out[o] = d, a0 += d;
        }
    }
    if (mode == 0) return;
    __shared__ float part[2][4];
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) a0 += __shfl_xor(a0, s), a1 += __shfl_xor(a1, s);
    if ((threadIdx.x & 63) == 0) part[0][threadIdx.x >> 6] = a0, part[1][threadIdx.x >> 6] = a1;
    __syncthreads();
    if (threadIdx.x == 0) {  // one atomic per block and quantity: thousands of waves on one address serialise
        a0 = part[0][0] + part[0][1] + part[0][2] + part[0][3], a1 = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        if (mode == 1) atomicAdd(acc + set * 4 + 0, a0), atomicAdd(acc + set * 4 + 1, a1);
        if (mode == 2) atomicAdd(acc + set * 4 + 2, a0);
        if (mode == 3) atomicAdd(acc + set * 4 + 3, a0);
    }
}

// output-layer coefficient vectors of one net: cf[k] = inv[k] * w3[k] (forward), cb[k] = w3[k] * inv[k] (same),
// c0 = b3 + sum_k sh[k] w3[k]
__global__ void out_coefs_kernel(const float* th, long set_th, int w3_off, int b3_off, int H2, const float* inv, const float* sh,
                                 long set_tab, float* cf, float* c0, long set_c) {
    __shared__ float part[256];
    const int set = blockIdx.x;
    const float* t = th + (long)set * set_th;
    float s = 0.f;
    for (int k = threadIdx.x; k < H2; k += 256) {
        const float w = t[w3_off + k];
        cf[(long)set * set_c + k] = inv[(long)set * set_tab + k] * w;
        s = fmaf(sh[(long)set * set_tab + k], w, s);
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) c0[set] = part[0] + t[b3_off];
}

// small gradients of the layers around the output: with sdq = sum_n d[n], u[k] = sum_n P2[n][k] d[n]:
//   dW3[k] = inv[k] u[k] + sh[k] sdq; db3 = sdq; dgamma[k] = w3[k] rs[k] (u[k] - mean[k] sdq); dbeta[k] = w3[k] sdq;
//   db2[k] = cs[k]
__global__ void out_grads_kernel(const float* th, long set_th, int w3_off, int H2, const float* inv, const float* sh,
                                 const float* rs, const float* mean, long set_tab, const float* u, const float* cs, long set_u,
                                 const float* acc, int acc_idx, float* g, long set_g, int gw3, int gb3, int gg, int gbe, int gb2,
                                 int write_w3) {
    const int set = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= H2) return;
    const float sdq = acc[set * 4 + acc_idx];
    const long tb = (long)set * set_tab + k;
    const float w3 = th[(long)set * set_th + w3_off + k], uk = u[(long)set * set_u + k];
    float* gs = g + (long)set * set_g;
    if (write_w3) {
        gs[gw3 + k] = fmaf(inv[tb], uk, sh[tb] * sdq);
        if (k == 0) gs[gb3] = sdq;
    }
    gs[gg + k] = w3 * rs[tb] * (uk - mean[tb] * sdq);
    gs[gbe + k] = w3 * sdq;
    gs[gb2 + k] = cs[(long)set * set_u + k];
}

// first-layer dgamma / dbeta: sum the NSLICE partial tables into the gradient slab
__global__ void bn1_flush_kernel(const float* dg, const float* dbe, long set_tab, long slice_stride, int t_off, int len, float* g,
                                 long set_g, int gg, int gbe) {
    const int set = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= len) return;
    float a = 0.f, b = 0.f;
    for (int sl = 0; sl < NSLICE; ++sl) {
        const long o = (long)sl * slice_stride + (long)set * set_tab + t_off + k;
        a += dg[o], b += dbe[o];
    }
    g[(long)set * set_g + gg + k] = a, g[(long)set * set_g + gbe + k] = b;
}

__global__ void losses_kernel(const float* acc, int Ns, int n_sets, float* losses) {
    const int set = threadIdx.x;
    if (set < n_sets && losses) {
        losses[set * 2 + 0] = acc[set * 4 + 0] / (float)Ns;   // mean((y - q)^2)   (trainer.py:496)
        losses[set * 2 + 1] = -acc[set * 4 + 2] / (float)Ns;  // -mean(q1)         (trainer.py:504)
    }
}

}  // namespace wide
}  // namespace avd

using namespace avd;
using namespace avd::wide;

extern "C" int avd_gemm_bt_bf16(int M, int Nc, int K, const void* A, long lda, const void* B, long ldb, float* D, long ldd,
                                void* stream) {
    AVD_REQUIRE(M > 0 && Nc > 0 && K > 0 && K % BK == 0 && lda >= K && ldb >= K && ldd >= Nc && ldd % 4 == 0,
                "avd_gemm_bt_bf16: M=%d Nc=%d K=%d (K %% 64 == 0) lda=%ld ldb=%ld ldd=%ld", M, Nc, K, lda, ldb, ldd);
    AVD_REQUIRE(A && B && D, "avd_gemm_bt_bf16: null pointer");
    GemmP p = {(const bf16*)A, (const bf16*)B, lda, ldb, 0, 0, M, Nc, K, 1};
    EpiStoreF32 e = {D, ldd, 0};
    return launch_gemm(p, e, 1, (hipStream_t)stream, "avd_gemm_bt_bf16");
}

// ------------------------------------------------------------------------------------------
// Fused forward pass for second layers of n x 512 columns (BASELINE config 5: hidden 1024): first layer -> second layer ->
// output-layer dot in ONE persistent kernel; the [N x K] first-layer activation matrix never exists in memory.
//   * unit of work = 128 rows x 512 columns on 4 waves (one per SIMD, 64 rows x 256 columns each = 256 accumulator registers,
//     pinned to the AGPRs);
//   * the A operand is GENERATED: per 32-feature chunk one v_mfma_f32_32x32x16_bf16 per 32 rows evaluates x . W1 + b1 from
//     bf16 hi/lo pairs of x, W1 and b1 in the 16 k slots (2^-16); relu + bf16 pack turns its accumulator tile
//     [feature][row] into the B operand of the two k-steps of the main product (k order = accumulator order: the weight
//     image is stored with bits 2 and 3 of k swapped);
//   * the other operand, bf16(inv (.) W2)^T, streams L2 -> LDS by global_load_lds in chunks of 32 k (+ the chunk's 1 KiB of
//     first-layer fragments) through FOUR stages = three chunks (99 KiB) in flight per CU: the stream is periodic in K and
//     carries on across the row tiles of a (set, column block) pair, whose tiles are dealt over ALL workgroups, so the chip
//     streams one 1.1 MB weight block at a time (resident in every XCD's L2). 64-byte image rows, 16-byte chunks XOR-swizzled by
//     bits 2..3 of the row: conflict-free for the lane groups ds_read_b128 is served in;
//   * the loop is rotated: the fragment reads of the next chunk, the counted vmcnt wait and the barrier sit between two
//     blocks of 16 MFMAs whose operands are already in registers.
// 128 FLOP per byte moved from L2 into LDS. The width-1 output layer rides along (rows complete in the kernel at 512 columns;
// one f32 atomic per row and column block beyond).
// ------------------------------------------------------------------------------------------
namespace avd { namespace fset { int cu_count(); } }
namespace fw {
constexpr int FR = 128, FC = 512, FK = 32, FSTG = 4, FT = 512;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split_bf(float x, bf16& hi, bf16& lo) {
    hi = (bf16)x;
    lo = (bf16)(x - (float)hi);
}
__host__ __device__ constexpr int kpos(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }  // swap bits 2 and 3

// k slots of the first-layer MFMA (lane half h holds slots 8h .. 8h + 7):
//   states : x = [xh0..3 | xl0..3 | xh0..3 | 1 1 0 0],  w = [wh0..3 | wh0..3 | wl0..3 | bh bl 0 0]
//   action : the same slots with x = (a, 0, 0, 0)
__device__ __forceinline__ bf16x8 x_frag_state(const float (&x)[4], bool live, int h) {
    bf16 hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_bf(live ? x[j] : 0.f, hi[j], lo[j]);
    const bf16 one = (bf16)(live ? 1.f : 0.f), zero = (bf16)0.f;
    bf16x8 v;
    v[0] = hi[0], v[1] = hi[1], v[2] = hi[2], v[3] = hi[3];
    v[4] = h ? one : lo[0], v[5] = h ? one : lo[1], v[6] = h ? zero : lo[2], v[7] = h ? zero : lo[3];
    return v;
}
// (the action branch uses the same slots with the action in input 0 and zeros elsewhere)
__device__ __forceinline__ bf16x8 x_frag_action(float a, bool live, int h) {
    const float x[4] = {a, 0.f, 0.f, 0.f};
    return x_frag_state(x, live, h);
}

// first-layer weight fragments [sets][nft][64 lanes] (tile t < nfs: 32 state features, else 32 action features)
__global__ void prep_wf1_kernel(const float* th, long set_th, int S, int ws_off, int bs_off, int H1, int wa_off, int ba_off, int Ha,
                                int nfs, int nft, bf16x8* wf1) {
    const int set = blockIdx.y, t = blockIdx.x, lane = threadIdx.x, f = lane & 31, h = lane >> 5;
    const float* T = th + (long)set * set_th;
    const bf16 zero = (bf16)0.f;
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = zero;
    if (t < nfs) {
        const int k = 32 * t + f;
        bf16 wh[4], wl[4], bh, bl;
#pragma unroll
        for (int j = 0; j < 4; ++j) split_bf((j < S && k < H1) ? T[ws_off + (long)j * H1 + k] : 0.f, wh[j], wl[j]);
        split_bf(k < H1 ? T[bs_off + k] : 0.f, bh, bl);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = h ? wl[j] : wh[j];
        // (slot 6 of the upper half repeats bh: the forward's x holds [1 1 0 0] there -- bh + bl --, the weight-gradient kernel's
        //  rank-one form [dh dh dl 0] with |d| = dh + dl the row's loss seed: dh bh + dh bl + dl bh)
        v[4] = h ? bh : wh[0], v[5] = h ? bl : wh[1], v[6] = h ? bh : wh[2], v[7] = h ? zero : wh[3];
    } else {
        const int k = 32 * (t - nfs) + f;
        bf16 wh, wl, bh, bl;
        split_bf(k < Ha ? T[wa_off + k] : 0.f, wh, wl);
        split_bf(k < Ha ? T[ba_off + k] : 0.f, bh, bl);
        v[0] = h ? wl : wh, v[4] = h ? bh : wh, v[5] = h ? bl : zero, v[6] = h ? bh : zero;  // the state layout with one input
    }
    wf1[((long)set * nft + t) * 64 + lane] = v;
}

// ------------------------------------------------------------------------------------------
// Rank-one backward of the output layer (r06). The output layers are one unit wide, so dZ2[n][c] = d[n] cf[c] [z2[n][c] > 0]
// with d the row's loss seed and cf = inv3 (.) w3: the matrix is never formed (out_bwd_kernel read one activation matrix and wrote
// one gradient matrix per backward pass: 2 x 1.08 ms at BASELINE config 5). The forward kernel stores the relu MASK as the exact
// bf16 image {1, 0}; d[n] is folded into the generated first layer of the weight gradient (|d| scales the row's inputs and bias
// slots, its sign is XORed into the packed relu'd tile) and multiplies the rows of the input gradient; cf[c] is folded into the
// input gradient's weight operand (prep_w2_kernel) and applied to dW2 at the end (w2_post_kernel), where u[c] = sum_n d relu(z2)
// and db2 also come from: relu(z2) = mask (y1 . W2' + bias2), so u[c] = sum_f W2'[f][c] G[f][c] + bias2[c] S2[c] with
// G = sum_n d y1 mask (the raw weight-gradient sums) and S2[c] = sum_n d[n] mask[n][c] (summed beside them in dw_gen_kernel).
//
// One record per 32-row chunk j of a set (j = 0 .. Np / 32: one more than there are chunks), everything dw_gen_kernel wants of
// the rows besides the mask, in the form it is consumed in -- ONE 2304-byte piece of its stream per chunk:
//   [   0, 1024)  the chunk's state fragments, lane (r, h) 16 B: x = [xh0..3 | xl0..3] (h = 0), [xh0..3 | dh dh dl 0] (h = 1), x = |d| s
//   [1024, 2048)  its action fragments (critic; the same slots with the action as input 0)
//   [2048, 2112)  sign words [h][k-step] x 4: bit 15 / 31 = sign of d of the two rows a packed pair of the relu'd tile holds
//   [2112, 2240)  d of the 32 rows of chunk j - 1 (the mask chunk that travels in the same stage), f32
// Rows >= Ns give zero fragments and d = 0 (their mask rows are whatever the forward made of a zero input).
// ------------------------------------------------------------------------------------------
constexpr int AUX_REC = 2304, AUX_ACT = 1024, AUX_SGN = 2048, AUX_D = 2112;
// dw_gen_kernel streams a record as one UNMASKED wave instruction per wave: wave w moves bytes [256 w, 256 w + 1024) -- the eight
// overlapping KiB cover the record and 512 B of the next one, overlaps carry identical bytes (an exec-masked instruction of 18 lanes per
// wave put a branch into the prepare phase)
constexpr int AUX_LDS = 7 * 256 + 1024;
__global__ __launch_bounds__(64) void aux_pack_kernel(const float* X, long setX, const float* act, long setAct, const float* d, long setD,
                                                      int Ns, int Np, unsigned char* aux, long setAux, float* dclean) {
    __shared__ float sd[32];
    const int set = blockIdx.y, j = blockIdx.x, lane = threadIdx.x, r = lane & 31, h = lane >> 5, nch = Np / FK;
    const long n = (long)j * FK + r;
    const bool live = j < nch && n < Ns;
    const float dv = live ? d[(long)set * setD + n] : 0.f, ad = fabsf(dv);
    if (h == 0) {
        sd[r] = dv;
        if (j < nch) dclean[(long)set * Np + n] = dv;  // the seed with zeros in the padding rows (dx_gen_kernel's row factor)
    }
    bf16 dh, dl;
    split_bf(ad, dh, dl);
    const bf16 zero = (bf16)0.f;
    auto frag = [&](const float (&x)[4]) {
        bf16 hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split_bf(x[i], hi[i], lo[i]);
        bf16x8 v;
        v[0] = hi[0], v[1] = hi[1], v[2] = hi[2], v[3] = hi[3];
        v[4] = h ? dh : lo[0], v[5] = h ? dh : lo[1], v[6] = h ? dl : lo[2], v[7] = h ? zero : lo[3];
        return v;
    };
    float xs[4] = {0.f, 0.f, 0.f, 0.f}, xa[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const wide::f32x4 xv = *(const wide::f32x4*)(X + (long)set * setX + n * 4);
        xs[0] = xv[0] * ad, xs[1] = xv[1] * ad, xs[2] = xv[2] * ad, xs[3] = xv[3] * ad;
        if (act) xa[0] = act[(long)set * setAct + n] * ad;
    }
    unsigned char* rec = aux + (long)set * setAux + (long)j * AUX_REC;
    ((bf16x8*)rec)[lane] = frag(xs);
    ((bf16x8*)(rec + AUX_ACT))[lane] = frag(xa);
    __syncthreads();
    if (lane < 16) {  // word (hh, ks, q): rows 16 ks + 8 (q >> 1) + 4 hh + 2 (q & 1) + {0, 1} -- dword q of the packed tile of k-step ks
        const int hh = lane >> 3, ks = (lane >> 2) & 1, q = lane & 3, row0 = 16 * ks + 8 * (q >> 1) + 4 * hh + 2 * (q & 1);
        ((unsigned*)(rec + AUX_SGN))[lane] = ((__float_as_uint(sd[row0]) >> 31) << 15) | ((__float_as_uint(sd[row0 + 1]) >> 31) << 31);
    } else if (lane < 48) {
        const long nn = (long)(j - 1) * FK + (lane - 16);
        ((float*)(rec + AUX_D))[lane - 16] = (j >= 1 && nn < Ns) ? d[(long)set * setD + nn] : 0.f;
    } else {
        ((unsigned*)(rec + AUX_D + 128))[lane - 48] = 0u;
    }
}

struct FwdP {
    const float* X;      // [sets][Ns][S]
    long setX;
    const float* act;    // critic: [sets][setAct] actions, else NULL
    long setAct;
    const bf16x8* wf1;   // [sets][nft][64]
    int nft, nfs;        // 32-feature tiles: all / state
    const bf16* WT;      // [sets][512][ldw], k permuted by kpos within groups of 16
    long setWT, ldw;
    const float *bias, *cf, *c0;  // [sets][512], [sets][512], [sets]
    bf16* P2;            // [sets][Np][H2] or NULL
    long setP2;
    int store_pre;       // != 0: the stored activations keep their sign (bf16 of z2, not of relu(z2)): fwd_delta_kernel continues from them
    int mask_out;        // != 0: the relu MASK of the (rounded) activations is stored instead, as the exact bf16 image {1, 0}: the
                         // rank-one backward's operand (nothing else reads the actor's activations)
    float dz_scale;      // != 0: store dZ2 = (relu(z2) > 0) * dz_scale * rw[row] * cf[col] there INSTEAD of the activations (the
    const float* rw;     // output-layer backward of a pass whose seed is the constant -1/N: critic(s, mu)); rw [sets][Ns] or NULL
    float* z;            // [sets][setZ]; H2 > 512: pre-filled with c0, every 512-column block adds its part (f32 atomics)
    long setZ;
    int Ns, Np, H2, n_sets;  // Np: padded rows (multiple of 128); H2: multiple of 512
    // EPI 4 (critic(s, a) and critic(s, mu) in one pass: see fwd_gen_kernel): the second action per row, its q sums, the action gradient
    const float* mu;     // [sets][setMu]
    long setMu;
    float* z2;           // [sets][setZ], pre-filled with c0: q(s, mu), one f32 atomic per row, column block and wave
    float* da;           // [sets][setDa], zeroed by the caller: dq(s, mu) / d mu per row (f32 atomics)
    long setDa;
    const float* wa;     // critic parameters + offset of the action layer's weights Wa[k]; set stride setTh
    long setTh;
    int Ha;
    unsigned long long* stamp;  // -DAVD_FW_STAMP: [8 waves][4] s_memtime sums of workgroup 16 (prepare, barrier, multiply, barrier)
    int dbg;                 // diagnostics build (AVD_FW_DBG bits: 1 no refill DMA, 2 no relu / pack, 4 no phase barriers, 8 no epilogue,
                             // 16 four of the 16 accumulating MFMAs, 32 no fragment reads): wrong results
};

// diagnostics build (-DAVD_FW_DBG): AVD_FW_DBG=<bits> switches pieces of the kernel off (tools/c5_dbg.sh @ tag r06-pre-prune; results wrong)
#ifdef AVD_FW_DBG
#define FW_DBG(bit) (p.dbg & (bit))
#else
#define FW_DBG(bit) false
#endif
// One stage = the 32 k x 512 column chunk (32 KiB) + the first-layer fragments of that chunk's 32 features (1 KiB).
// Tile of a workgroup: GR rows x GC columns (r06c: 256 x 256, was 128 x 512 -- same 64 x 128 per wave, same accumulators, HALF the
// weight bytes streamed per row: the kernel is bound by that stream, not by the matrix pipe -- with the refill compiled out a pass took
// 1.59 ms, with half its bytes 1.79, with all of them 2.46; the ablation libraries of docs/ENGINEERING_NOTES_r06.md)
constexpr int GR = 256, GC = 256, NRQ = GR / 64, NCH = GC / 128;  // row quarters x column halves = the 8 waves
constexpr int STG_BYTES = GC * FK * 2 + 1024;
constexpr int L_ZS = FSTG * STG_BYTES, L_BIAS = L_ZS + 2 * NCH * GR * 4, L_CF = L_BIAS + GC * 4, L_XR = L_CF + GC * 4, L_AR = L_XR + GR * 16,
              L_XF = L_AR + GR * 4, L_ZERO = L_XF + 2 * NRQ * 4 * 1024, L_TOTAL = L_ZERO + 256,
              L_MK = L_TOTAL, L_MU = L_MK + 256, L_TOTAL_DUAL = L_MU + 2 * GR * 4;  // (EPI 4: Wa[64]; the rows' second actions, two tiles deep)
static_assert(L_TOTAL_DUAL <= 160 * 1024, "fwd_gen_kernel: LDS");
// XF: [2 buffers][4 row quarters][state, action][2 row tiles][64 lanes] x 16 B; ZERO: 256 B of zeros (the output-layer MFMA's idle A rows)
//
// EPI picks the tile epilogue at compile time (r06b; the r03 epilogue took every decision per group of four elements at run time --
// uniform branches on p.dz_scale / p.store_pre / p.mask_out --, quieted every element with v_max_f32 x, x, converted one element per
// v_cvt_pk_bf16_f32 and packed pairs with shifts: ~3400 instructions per wave and tile, with BOTH waves of a SIMD in their epilogues
// at the same time and the matrix pipe idle):
//   0  the r03 code, run-time flags: relu'd activations or dZ2 out (the layer-wise backward's operands; no BASELINE shape runs it)
//   1  nothing stored (target networks)      2  bf16(z2) with its sign (critic(s, a): fwd_delta_kernel continues from it)
//   3  the relu mask {1, 0} (actor(s): the rank-one backward's operand)
// 1..3: one v_cvt_pk_bf16_f32 + one v_pk_max_i16 per PAIR, and the width-1 output layer on the matrix pipe: the packed relu'd tile is
// the B operand (k = its 16 columns per k-step) of an MFMA whose A operand holds cf as a bf16 pair in rows 0 (hi) and 1 (lo) and zeros
// below -- 16 MFMAs per wave and tile replace 128 shifts + 128 FMAs + a cross-lane add.
//   4  (critic) critic(s, a) AND critic(s, mu) of a learn step in one pass: the two share the states and the weights, so
//      z2(mu) = z2(a) + W2[action features] . (f(mu) - f(a)), f = the relu'd first layer of the action branch (r05: fwd_delta_kernel, on
//      the signed activations this kernel stored -- one matrix written, read and written again as a mask). Here the chunk order is
//      rotated: the action branch's chunks are the FIRST of the period (physical chunk = (c + nfs) mod nk), so at a tile boundary,
//      where chunks 0 and 1 of the next period have landed, the stages hold exactly the weights the second pass needs:
//        epilogue A  q(s, a) sums; the relu mask of z2(a) stored (the rank-one backward's operand);
//        delta       acc += W2[action chunk] . bf16(relu(p1(mu)) - relu(p1(a))) for both action tiles (8 + 32 MFMAs);
//        epilogue B  q(s, mu) sums (atomics); the action gradient da[n] = drow sum_k [p1(mu) > 0] Wa[k] sum_c [z2(mu) > 0] cf[c] W2'[k][c]:
//                    one more product over the columns, B = the mask tile times bf16(cf) as it stands, A = the action chunks read
//                    TRANSPOSED from the same stages (ds_read_b64_tr_b16; the image rows are the contraction index here).
template <bool CRITIC, int EPI>
__global__ __launch_bounds__(FT) __attribute__((amdgpu_waves_per_eu(2, 2))) void fwd_gen_kernel(FwdP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* zs = (float*)(smem_raw + L_ZS);      // [2 tiles][NCH column halves][GR]
    float* sbias = (float*)(smem_raw + L_BIAS);  // [GC]
    float* scf = (float*)(smem_raw + L_CF);      // [GC] (EPI 0) or the cf fragments
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int rh = wave & 3, cq = wave >> 2;  // 64-row quarter, 128-column half (cq = the wave's ping-pong group)
    const int nk = p.nft, ncb = p.H2 / GC, ntile = p.Np / GR;
    constexpr bool DUAL = EPI == 4;
    static_assert(!DUAL || CRITIC, "EPI 4 is the critic's pass");
    // DUAL: the action branch's chunks lead the period (chunk c of the period = physical 32-feature chunk phys(c) of the image / fragments)
    const int nact = DUAL ? nk - p.nfs : 0;
    auto phys = [&](int c) { return DUAL ? (c < nact ? c + p.nfs : c - nact) : c; };
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mfma = [](bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); };
    // relu + bf16 of half a first-layer tile: k-step ks of the chunk <- registers 8 ks .. 8 ks + 7
    // (one v_cvt_pk_bf16_f32 + one v_pk_max_i16 per pair: a negative bf16 is a negative int16)
    auto pack = [&](const f32x16& p1, int ks, bf16x8& b) {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 f = {p1[8 * ks + 2 * i], p1[8 * ks + 2 * i + 1]};
            const bf16x2 v = __builtin_convertvector(f, bf16x2);  // one v_cvt_pk_bf16_f32
            const s16x2 z = {0, 0};
            w[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
        }
        wide::u32x4 o;
        o[0] = w[0], o[1] = w[1], o[2] = w[2], o[3] = w[3];
        b = __builtin_bit_cast(bf16x8, o);
    };
    // fragment reads: image row = 128 cq + 32 ct + r, logical 16-byte chunk 2 ks + h, swizzled by bits 2..3 of the row
    const int sw = (r >> 2) & 3;
    const int rd0 = ((128 * cq + r) * FK + (((0 + h) ^ sw) << 3)) * 2, rd1 = ((128 * cq + r) * FK + (((2 + h) ^ sw) << 3)) * 2;
    auto read_frags = [&](int stg, int ks, bf16x8 (&a)[4]) {
        const unsigned char* b = smem_raw + stg * STG_BYTES + (ks ? rd1 : rd0);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) a[ct] = *(const bf16x8*)(b + ct * 32 * FK * 2);
    };
    auto wf_at = [&](int stg) { return *(const bf16x8*)(smem_raw + stg * STG_BYTES + GC * FK * 2 + lane * 16); };
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // Workgroups start up to one row tile apart (16 steps of ~2 us): in step, all 256 would write their 128 KiB of activations at
    // the same moment, every tile boundary a 32 MB burst that the stores sit out in the issue queue
    if (p.P2)
        for (int i = 0; i < (int)(blockIdx.x & 15); ++i) __builtin_amdgcn_s_sleep(72);

    // every workgroup walks the (set, 512-column block) pairs in the same order and takes its share of each pair's row tiles:
    // at any time the chip streams ONE 512 x K weight block (1.1 MB: every XCD's L2 holds it)
    for (int pair = 0; pair < p.n_sets * ncb; ++pair) {
        const int set = pair / ncb, cb = pair - set * ncb;
        int tile = blockIdx.x;
        if (tile >= ntile) continue;  // (uniform per workgroup)
        // wave w fills image rows (= output columns) [32 w, 32 w + 32), 16 rows per instruction (lane -> row l / 4, slot l % 4),
        // and dwords [64 (w % 4), ..) of the chunk's first-layer fragments (waves 4..7 repeat what waves 0..3 write: one
        // instruction count for everybody). Addresses are wave-uniform base + one 32-bit lane offset (scalar-base form).
        const char* ubw = (const char*)(p.WT + (long)set * p.setWT + (long)(GC * cb + 32 * wv) * p.ldw);
        const unsigned vow = (unsigned)(((lane >> 2) * p.ldw + (((lane & 3) ^ ((lane >> 4) & 3)) << 3)) * 2);
        const long g16 = 32 * p.ldw;  // bytes between instructions (16 image rows)
        const char* ubf = (const char*)((const float*)(p.wf1 + (long)set * nk * 64) + 64 * (wv & 3));
        const unsigned vof = (unsigned)lane * 4u;
        // one chunk = 3 wave-instructions, issued in two parts (0: one image instruction, 1: the other + the fragments)
        auto dma = [&](int stg, int kc, int part) {
            unsigned char* l = smem_raw + stg * STG_BYTES;
            unsigned vw = vow;
            asm volatile("" : "+v"(vw));  // (opaque per call: keeps the scalar-base + lane-offset address form)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (i == part)
                    __builtin_amdgcn_global_load_lds((gptr_t)(ubw + i * g16 + phys(kc) * FK * 2 + vw), (lptr_t)(l + (32 * wv + 16 * i) * FK * 2), 16, 0, 0);
            if (part == 1) {  // the first-layer fragments of the FOLLOWING chunk travel with this one
                const int kf = phys(kc + 1 == nk ? 0 : kc + 1);
                __builtin_amdgcn_global_load_lds((gptr_t)(ubf + (long)kf * 1024 + (vw & 0u) + vof), (lptr_t)(l + GC * FK * 2 + 256 * (wv & 3)), 4, 0, 0);
            }
        };
        // raw inputs of a row tile into LDS buffer b (every wave issues the same 4 / 8 / 12 instructions: uniform counts). Rows beyond
        // Ns are clamped to row Ns - 1 here and zeroed when the fragments are built.
        auto dma_x = [&](int t) {
            const int mub = t / (int)gridDim.x & 1;  // (of the tile asked for: a clamped request must not land in the running tile's buffer)
            if (t >= ntile) t = ntile - 1;
#pragma unroll
            for (int half = 0; half < NRQ; ++half) {
                int n = t * GR + 64 * half + lane;
                n = n < p.Ns ? n : p.Ns - 1;
                __builtin_amdgcn_global_load_lds((gptr_t)(p.X + (long)set * p.setX + (long)n * 4), (lptr_t)(smem_raw + L_XR + (64 * half) * 16), 16, 0, 0);
                if (CRITIC)
                    __builtin_amdgcn_global_load_lds((gptr_t)(p.act + (long)set * p.setAct + n), (lptr_t)(smem_raw + L_AR + (64 * half) * 4), 4, 0, 0);
                if (DUAL)  // (two tiles deep: the tile's epilogue reads its rows while the next tile's are already here)
                    __builtin_amdgcn_global_load_lds((gptr_t)(p.mu + (long)set * p.setMu + n),
                                                     (lptr_t)(smem_raw + L_MU + (mub * GR + 64 * half) * 4), 4, 0, 0);
            }
        };
        // fragments of row tile t from the raw rows, into fragment buffer b (the four waves with cq == 0 do it for their row quarter)
        auto build_x = [&](int t, int b) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int row = rh * 64 + 32 * rt + r, n = t * GR + row;
                const bool live = t < ntile && n < p.Ns;
                const f32x4 xv = *(const f32x4*)(smem_raw + L_XR + row * 16);
                const float x[4] = {xv[0], xv[1], xv[2], xv[3]};
                bf16x8* d = (bf16x8*)(smem_raw + L_XF + ((b * NRQ + rh) * 4 + rt) * 1024) + lane;
                d[0] = x_frag_state(x, live, h);
                if (CRITIC) d[128] = x_frag_action(*(const float*)(smem_raw + L_AR + row * 4), live, h);
            }
        };
        auto x_frag = [&](int b, bool action, int rt) {
            return *((const bf16x8*)(smem_raw + L_XF + ((b * NRQ + rh) * 4 + (action ? 2 : 0) + rt) * 1024) + lane);
        };
        __syncthreads();  // (the previous pair's LDS reads are done)
        if (tid < GC) sbias[tid] = p.bias[(long)set * p.H2 + GC * cb + tid];
        if constexpr (EPI == 0) {
            if (tid < GC) scf[tid] = p.cf[(long)set * p.H2 + GC * cb + tid];
        } else {
            // cf as A fragments: [column quarter][hi, lo][32-column tile][k-step][lane half] x 16 B; slot i of lane half hh holds the
            // column that register 8 ks + i of the accumulator tile holds there: 16 ks + 8 (i >> 2) + 4 hh + (i & 3)
            bf16 chi, clo;
            split_bf(p.cf[(long)set * p.H2 + GC * cb + (tid & (GC - 1))], chi, clo);
            const int c32 = tid & 31, slot = ((((tid >> 5) & 3) * 2 + (c32 >> 4)) * 2 + ((c32 >> 2) & 1)) * 8 + ((c32 >> 3) & 1) * 4 + (c32 & 3);
            bf16* t = (bf16*)(smem_raw + L_CF) + (tid >> 7) * 256;
            if (tid < GC) t[slot] = chi, t[128 + slot] = clo;
            if (tid < 64) ((unsigned*)(smem_raw + L_ZERO))[tid] = 0u;
            if (DUAL && tid < 64) ((float*)(smem_raw + L_MK))[tid] = tid < p.Ha ? p.wa[(long)set * p.setTh + tid] : 0.f;
        }
        // (chunk 0's first-layer fragments travel with chunk nk - 1: not in a stage yet. Requested HERE, ahead of the stream's first
        //  chunks, the load's latency is covered by the wait below instead of standing alone behind it -- once per pair)
        const bf16x8 wf0 = p.wf1[((long)set * nk + phys(0)) * 64 + lane];
        dma_x(tile);
        // ---- start the stream: chunks 0 .. 2 into stages 0 .. 2 (a stage also carries the first-layer fragments of the NEXT chunk)
#pragma unroll
        for (int c = 0; c < FSTG - 1; ++c) dma(c, c % nk, 0), dma(c, c % nk, 1);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();
        if (cq == 0) build_x(tile, 0);
        __syncthreads();
        // PING-PONG: waves 0..3 (group 0, one per SIMD) and waves 4..7 (group 1) alternate between two kinds of phase, a barrier
        // after each: "prepare" (fragment reads, relu / pack of the first layer, the refill DMA, epilogues) and "multiply"
        // (16 + 2 MFMAs at raised priority). Group 1 runs one phase behind, so on every SIMD one wave multiplies while the
        // other prepares -- left in step (one barrier per chunk, same program) the two waves of a SIMD ran their MFMAs
        // together and their VALU / DMA together and the pipe idled through the latter (41-45 % MfmaUtil).
        //   prepare(k) : A <- stage of chunk k (both k-steps); wfn <- that stage's fragments of chunk k + 1; pack(p1n) -> bfr;
        //                request chunk k + 3 into the stage of chunk k - 1 (read by both groups two barriers ago)
        //   multiply(k): acc += A . bfr; p1n = first layer of chunk k + 1
        // Every wave waits for vmcnt(6) before every barrier: its share of every chunk but the two youngest has landed, which
        // covers whatever anybody reads in the next phase.
        const int grp = wave >> 2;
        bf16x8 bfr[2][2];  // the chunk's B operand (k-step ks of row tile rt: bfr[rt][ks]): relu'd first layer, built one multiply phase ahead
        {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x16 p1 = mfma(wf0, x_frag(0, DUAL, rt), zero16);
                pack(p1, 0, bfr[rt][0]), pack(p1, 1, bfr[rt][1]);
            }
        }
        int stg = 0;   // stage of the chunk being prepared / multiplied (the stream is periodic in nk, across row tiles)
        int xb = 0;    // fragment buffer of the current tile's inputs
        // output-layer sums: every wave leaves its 128-column partial of tile t in zs[t & 1]; the four waves with cq == 1 add the
        // two partials a whole tile later (both groups have written and passed barriers by then)
        auto z_flush = [&](int zt_) {
            if (zt_ >= 0 && cq == 1) {
                const int row = rh * 64 + lane;
                float* zd = p.z + (long)set * p.setZ + (long)zt_ * GR + row;
                const float* zq = zs + (zt_ / (int)gridDim.x & 1) * NCH * GR;
                const float zt = zq[row] + zq[GR + row];
                if (ncb == 1)
                    *zd = p.c0[set] + zt;
                else
                    atomicAdd(zd, zt);
            }
        };
#ifdef AVD_FW_STAMP
        unsigned long long facc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, flast = __builtin_amdgcn_s_memtime();
#define FW_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); facc[i] += t_ - flast; flast = t_; }
#else
#define FW_STAMP(i)
#endif
        f32x16 acc[2][4];
        auto init_acc = [&]() {  // the folded bias: tile (rt, ct), register 4 g + j <-> column 128 cq + 32 ct + 8 g + 4 h + j
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *(const f32x4*)(sbias + 128 * cq + 32 * ct + 8 * g + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[0][ct][4 * g + j] = bv[j], acc[1][ct][4 * g + j] = bv[j];
                }
        };
        // relu, bf16, output-layer dot on the stored (rounded) activations, row-major store
        auto epilogue = [&](int pt) {
            if constexpr (EPI != 0) {
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                // (lane-derived addresses from an OPAQUE copy of the lane id: hoisted out of the tile loop -- once per kernel -- they are state the
                //  allocator spills over the K loop and reloads here, and a scratch reload issued behind the tile's stores waits for them)
                int lane = tid & 63;
                asm volatile("" : "+v"(lane));
                const int r = lane & 31, h = lane >> 5;
                const int rh = wv & 3, cq = wv >> 2;  // (the wave's tile from the SCALAR wave id: address parts in SGPRs, not hoisted vector registers)
                const int cfb = r < 2 ? L_CF + (cq * 2 + r) * 256 + h * 16 : L_ZERO + h * 16;  // this lane's row of the cf fragments
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const long n = (long)pt * GR + rh * 64 + 32 * rt + r;
                    // (four lanes per row: lane (r, h) stores row r & 15 (+ 16 in the second instruction), columns 16 (r >> 4) + 8 h .. + 8 of every tile)
                    bf16* dst4 = EPI >= 2 ? p.P2 + (long)set * p.setP2 + (n - (r & 16)) * p.H2 + GC * cb + 128 * cq + (r & 16) + 8 * h : nullptr;
                    f32x16 E = zero16;  // rows 0 / 1 (registers 0 / 1 of the lower lane half): sum over this wave's columns of o . cf_hi / o . cf_lo
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        unsigned raw[4][2], rl[4][2];
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const f32x2 f = {acc[rt][ct][4 * g + 2 * e], acc[rt][ct][4 * g + 2 * e + 1]};
                                const s16x2 v = __builtin_bit_cast(s16x2, __builtin_convertvector(f, bf16x2)), z = {0, 0};
                                raw[g][e] = __builtin_bit_cast(unsigned, v);
                                rl[g][e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, z));
                            }
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            wide::u32x4 bw;
                            bw[0] = rl[2 * ks][0], bw[1] = rl[2 * ks][1], bw[2] = rl[2 * ks + 1][0], bw[3] = rl[2 * ks + 1][1];
                            E = mfma(*(const bf16x8*)(smem_raw + cfb + (ct * 2 + ks) * 32), __builtin_bit_cast(bf16x8, bw), E);
                        }
                        if constexpr (EPI >= 2) {
                            // 16-byte row-major pieces: after the half swaps a row's two lanes hold 32 contiguous bytes twice (columns 0-15
                            // and 16-31 of the tile); the 16-lane swaps then hand the second half of rows 0-15 to lanes 16-31 and the first
                            // half of rows 16-31 to lanes 0-15: FOUR lanes per row, 64 contiguous bytes, 16 rows per store instruction (the
                            // store tail is bound by the segments an instruction touches: 32 x 32 B took twice as long)
                            wide::u32x4 ov[2];
#pragma unroll
                            for (int gg = 0; gg < 2; ++gg) {
                                unsigned o[2][2];
#pragma unroll
                                for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                                    for (int e = 0; e < 2; ++e) o[g2][e] = EPI == 2 ? raw[2 * gg + g2][e] : wide::relu_mask2(rl[2 * gg + g2][e]);
                                const auto s0 = __builtin_amdgcn_permlane32_swap(o[0][0], o[1][0], false, false);
                                const auto s1 = __builtin_amdgcn_permlane32_swap(o[0][1], o[1][1], false, false);
                                ov[gg][0] = s0[0], ov[gg][1] = s1[0], ov[gg][2] = s0[1], ov[gg][3] = s1[1];
                            }
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                const auto t = __builtin_amdgcn_permlane16_swap(ov[0][d], ov[1][d], false, false);
                                ov[0][d] = t[0], ov[1][d] = t[1];
                            }
                            *(wide::u32x4*)(dst4 + 32 * ct) = ov[0];
                            *(wide::u32x4*)(dst4 + 16 * p.H2 + 32 * ct) = ov[1];
                        }
                    }
                    if (h == 0) zs[((pt / (int)gridDim.x & 1) * NCH + cq) * GR + rh * 64 + 32 * rt + r] = E[0] + E[1];
                }
            } else {
            float zp[2] = {0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const long n = (long)pt * GR + rh * 64 + 32 * rt + r;
                bf16* dst = p.P2 ? p.P2 + (long)set * p.setP2 + n * p.H2 + GC * cb + 128 * cq + 8 * h : nullptr;
                const float drow = (p.dz_scale != 0.f && n < p.Ns) ? p.dz_scale * (p.rw ? p.rw[(long)set * p.Ns + n] : 1.f) : 0.f;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    unsigned pk[4][2];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 cv = *(const f32x4*)(scf + 128 * cq + 32 * ct + 8 * g + 4 * h);
                        bf16 o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            o[j] = (bf16)fmaxf(acc[rt][ct][4 * g + j], 0.f);
                            zp[rt] = fmaf((float)o[j], cv[j], zp[rt]);
                        }
                        if (p.dz_scale != 0.f) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = (bf16)((float)o[j] > 0.f ? drow * cv[j] : 0.f);
                        } else if (p.store_pre) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = (bf16)acc[rt][ct][4 * g + j];
                        }
                        pk[g][0] = (unsigned)__builtin_bit_cast(unsigned short, o[0]) | ((unsigned)__builtin_bit_cast(unsigned short, o[1]) << 16);
                        pk[g][1] = (unsigned)__builtin_bit_cast(unsigned short, o[2]) | ((unsigned)__builtin_bit_cast(unsigned short, o[3]) << 16);
                        if (p.mask_out) pk[g][0] = wide::relu_mask2(pk[g][0]), pk[g][1] = wide::relu_mask2(pk[g][1]);
                    }
                    if (dst) {
#pragma unroll
                        for (int gg = 0; gg < 2; ++gg) {  // 16-byte row-major pieces: the row's two lanes cover 32 contiguous bytes
                            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gg][0], pk[2 * gg + 1][0], false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gg][1], pk[2 * gg + 1][1], false, false);
                            wide::u32x4 o;
                            o[0] = s0[0], o[1] = s1[0], o[2] = s0[1], o[3] = s1[1];
                            *(wide::u32x4*)(dst + 32 * ct + 16 * gg) = o;
                        }
                    }
                }
                zp[rt] += __shfl_xor(zp[rt], 32);
                if (h == 0) zs[((pt / (int)gridDim.x & 1) * NCH + cq) * GR + rh * 64 + 32 * rt + r] = zp[rt];
            }
            }
        };
        // EPI 4: the tile's second pass -- critic(s, mu) on top of the accumulators of critic(s, a) (the kernel's header). `stg` is the stage
        // of chunk 0 of the next period; rwv = the rows' weights, loaded before epilogue A's stores (a load behind them would wait for them)
        auto dual_tail = [&](int pt, const float (&rwv)[2]) {
            if constexpr (DUAL) {
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                // (every wave waited for vmcnt(0) on its way here: behind this barrier chunks 0 and 1 of the next period are complete)
                __builtin_amdgcn_s_barrier();
                // (lane-derived addresses from an OPAQUE copy of the lane id: hoisted out of the tile loop they are more loop state for the
                //  allocator to spill -- and a scratch reload behind epilogue A's stores waits for those stores)
                int lane = tid & 63;
                asm volatile("" : "+v"(lane));
                const int r = lane & 31, h = lane >> 5;
                const int rh = wv & 3, cq = wv >> 2;
                const int cfb = r < 2 ? L_CF + (cq * 2 + r) * 256 + h * 16 : L_ZERO + h * 16;
                auto x_frag = [&](int b, bool action, int rt) {
                    return *((const bf16x8*)(smem_raw + L_XF + ((b * NRQ + rh) * 4 + (action ? 2 : 0) + rt) * 1024) + lane);
                };
                auto wf_at = [&](int stg_) { return *(const bf16x8*)(smem_raw + stg_ * STG_BYTES + GC * FK * 2 + lane * 16); };
                const int sw = (r >> 2) & 3;
                const int rd0 = ((128 * cq + r) * FK + (((0 + h) ^ sw) << 3)) * 2, rd1 = ((128 * cq + r) * FK + (((2 + h) ^ sw) << 3)) * 2;
                auto read_frags = [&](int stg_, int ks, bf16x8 (&a)[4]) {
                    const unsigned char* b = smem_raw + stg_ * STG_BYTES + (ks ? rd1 : rd0);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) a[ct] = *(const bf16x8*)(b + ct * 32 * FK * 2);
                };
                const int xbf = xb ^ 1;  // the finished tile's fragments (xb already points at the next tile's)
                const float* mur = (const float*)(smem_raw + L_MU) + (pt / (int)gridDim.x & 1) * GR + rh * 64 + r;
                // the action tiles' first-layer fragments: a stage carries those of the FOLLOWING chunk (tile 0: with chunk nk - 1, tile 1: with chunk 0)
                const bf16x8 wfa[2] = {wf_at((stg + FSTG - 1) & (FSTG - 1)), wf_at(stg)};
                unsigned pos[2] = {0u, 0u};  // [rt] bit 16 ta + i: p1(mu) > 0 in register i of action tile ta
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) {
                    bf16x8 dfr[2][2];  // [rt][ks]
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        const long n = (long)pt * GR + rh * 64 + 32 * rt + r;
                        const f32x16 pa = mfma(wfa[ta], x_frag(xbf, true, rt), zero16);
                        const f32x16 pm = mfma(wfa[ta], x_frag_action(mur[32 * rt], n < p.Ns, h), zero16);
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            wide::u32x4 o;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const f32x2 f = {wide::relu1(pm[8 * ks + 2 * i]) - wide::relu1(pa[8 * ks + 2 * i]),
                                                 wide::relu1(pm[8 * ks + 2 * i + 1]) - wide::relu1(pa[8 * ks + 2 * i + 1])};
                                o[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
                            }
                            dfr[rt][ks] = __builtin_bit_cast(bf16x8, o);
                        }
#pragma unroll
                        for (int i = 0; i < 16; ++i) pos[rt] |= pm[i] > 0.f ? 1u << (16 * ta + i) : 0u;
                    }
                    __builtin_amdgcn_sched_barrier(0);  // (the first layers before the weight fragments: registers)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        bf16x8 A2[4];
                        read_frags((stg + ta) & (FSTG - 1), ks, A2);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                            for (int rt = 0; rt < 2; ++rt) acc[rt][ct] = mfma(A2[ct], dfr[rt][ks], acc[rt][ct]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // (the masks are complete HERE. Left to the compiler, the 32 compares of row tile 1 were deferred to the start of its sweep,
                //  with the first-layer tiles live across row tile 0's sweeps -- and that build returned wrong action gradients for part
                //  of those rows: tests/test_gpu_wide.py's oracle comparison at hidden 1024 caught it, the dual-vs-delta row check pins it)
                asm volatile("" : "+v"(pos[0]), "+v"(pos[1]));
                // ---- epilogue B. Transposed reads of the action chunks (A = [feature][column] from image rows = columns): of a 16-lane group
                // (mm = which 16 features, rows by lane half h) lane 4 q + pp addresses row c0 + q, features 16 mm + 4 pp .. + 3 -- image position
                // 16 mm + 4 swap(pp) (bits 2 and 3 of k are swapped in the image), 16-byte piece XOR-swizzled by bits 2..3 of the row = (2 g + h) & 3;
                // lane i of the group gets feature 16 mm + i of rows c0 .. c0 + 3 = slots 4 g .. 4 g + 3 with c0 = 16 ks + 8 g + 4 h of the tile
                const int tq = (lane & 15) >> 2, tp = lane & 3, mm = (lane >> 4) & 1, sp = ((tp & 1) << 1) | (tp >> 1), lch = 2 * mm + (sp >> 1);
                const int trow = (128 * cq + 4 * h + tq) * FK * 2 + (sp & 1) * 8;
                const unsigned tr0 = (unsigned)(trow + ((lch ^ h) << 4)), tr1 = (unsigned)(trow + 8 * FK * 2 + ((lch ^ (2 + h)) << 4));
                const unsigned sb0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem_raw + stg * STG_BYTES),
                               sb1 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem_raw + ((stg + 1) & (FSTG - 1)) * STG_BYTES);
                const unsigned a00 = sb0 + tr0, a01 = sb0 + tr1, a10 = sb1 + tr0, a11 = sb1 + tr1;  // [action tile][g]
                const int cfh = L_CF + cq * 2 * 256 + h * 16;  // bf16(cf) of this lane half's columns, in slot order (the hi row of the cf fragments)
                const float* mk = (const float*)(smem_raw + L_MK);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const long n = (long)pt * GR + rh * 64 + 32 * rt + r;
                    const bool live = n < p.Ns;
                    // the relu'd tile as packed bf16 first (32 registers for the row tile's 64 accumulators, dead from here on), then two sweeps
                    // over it -- q(s, mu), then the action gradient: with the accumulators and both products' sums live at the same time the
                    // allocator spilled the loop's state around the tail, and a scratch reload behind epilogue A's stores waits for them
                    unsigned rla[4][4][2];
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const f32x2 f = {acc[rt][ct][4 * g + 2 * e], acc[rt][ct][4 * g + 2 * e + 1]};
                                const s16x2 v = __builtin_bit_cast(s16x2, __builtin_convertvector(f, bf16x2)), z = {0, 0};
                                rla[ct][g][e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, z));
                            }
                    __builtin_amdgcn_sched_barrier(0);
                    float qv;
                    {
                        f32x16 Q = zero16;
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) {
                            const unsigned (&rl)[4][2] = rla[ct];
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks) {
                                wide::u32x4 bw;
                                bw[0] = rl[2 * ks][0], bw[1] = rl[2 * ks][1], bw[2] = rl[2 * ks + 1][0], bw[3] = rl[2 * ks + 1][1];
                                Q = mfma(*(const bf16x8*)(smem_raw + cfb + (ct * 2 + ks) * 32), __builtin_bit_cast(bf16x8, bw), Q);
                            }
                        }
                        qv = Q[0] + Q[1];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    f32x16 E[2] = {zero16, zero16};
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const unsigned (&rl)[4][2] = rla[ct];
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                            u32x2_t t[2][2];  // [action tile][g]
                            {   // (the builtin: here no LDS-DMA is outstanding by the compiler's count -- the boundary's explicit vmcnt(0) --, so it puts no
                                //  vm wait in front of the reads; dw_gen_kernel's inline-asm form, consumed by MFMAs right behind its own
                                //  lgkmcnt(0), gave wrong action gradients here)
                                typedef short s16x4_t __attribute__((ext_vector_type(4)));
                                typedef __attribute__((address_space(3))) s16x4_t* lp_t;
                                const int off = ct * 32 * FK * 2 + ks * 16 * FK * 2;
                                t[0][0] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(size_t)(a00 + off)));
                                t[0][1] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(size_t)(a01 + off)));
                                t[1][0] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(size_t)(a10 + off)));
                                t[1][1] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(size_t)(a11 + off)));
                            }
                            // cf as its bf16 pair, one product each: the rounding error of a single bf16(cf[c]) is the SAME for all 48 features, so
                            // it adds up coherently over k -- a row-independent offset of 1-3 % of the action gradient with the hi part alone
#pragma unroll
                            for (int part = 0; part < 2; ++part) {
                                wide::u32x4 bz;
                                const wide::u32x4 cv = *(const wide::u32x4*)(smem_raw + cfh + part * 256 + (ct * 2 + ks) * 32);
                                bz[0] = wide::relu_select2(rl[2 * ks][0], cv[0]), bz[1] = wide::relu_select2(rl[2 * ks][1], cv[1]);
                                bz[2] = wide::relu_select2(rl[2 * ks + 1][0], cv[2]), bz[3] = wide::relu_select2(rl[2 * ks + 1][1], cv[3]);
#pragma unroll
                                for (int mt = 0; mt < 2; ++mt) {
                                    wide::u32x4 af;
                                    af[0] = t[mt][0][0], af[1] = t[mt][0][1], af[2] = t[mt][1][0], af[3] = t[mt][1][1];
                                    E[mt] = mfma(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bz), E[mt]);
                                }
                            }
                        }
                    }
                    // E[mt]: lane = row, register 4 g + j <-> action feature 32 mt + 8 g + 4 h + j (the layout of p1(mu) and of `pos`)
                    float dap = 0.f;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 mv = *(const f32x4*)(mk + 32 * mt + 8 * g + 4 * h);
#pragma unroll
                            for (int j = 0; j < 4; ++j) dap += ((pos[rt] >> (16 * mt + 4 * g + j)) & 1u) ? mv[j] * E[mt][4 * g + j] : 0.f;
                        }
                    dap += __shfl_xor(dap, 32);
                    if (h == 0) {
                        atomicAdd(p.z2 + (long)set * p.setZ + n, qv);
                        if (live) atomicAdd(p.da + (long)set * p.setDa + n, p.dz_scale * rwv[rt] * dap);
                    }
                }
                // nobody is still reading this tile's second actions or the stages' fragments when the first wave of group 0 starts the next
                // tile (its first requests -- the next second actions, chunk 3 into the stage chunk nk - 1 and action tile 0's fragments
                // sit in -- are issued before its first phase barrier): without this barrier slow waves read the tile after next's actions
                __builtin_amdgcn_s_barrier();
            }
        };
        init_acc();
        int prev = -1;  // tile whose partial output-layer sums were written in the previous epilogue
        for (; tile < ntile; tile += gridDim.x) {
            // K loop a phase apart, tile boundary level (as in dx_gen_kernel): group 1 drops a phase behind here, group 0 waits for it
            // after the loop -- the two epilogues run side by side instead of one after the other beside an idle matrix pipe
            if (grp == 1) __builtin_amdgcn_s_barrier();
            dma_x(tile + gridDim.x);  // the next tile's raw inputs (fragments are built in step nk - 2, used in step nk - 1)
            // one step = prepare(kt), barrier, multiply(kt), barrier. The first two steps of a tile wait for no vm operation (see the
            // tile boundary below), the others for all but the two youngest chunks
            auto step = [&](int kt, auto wait_c) {
                constexpr bool WAIT = decltype(wait_c)::value;
                // ================= prepare(kt)
                FW_STAMP(3);
                bf16x8 A[2][4];
                if (!FW_DBG(32)) {
                    read_frags(stg, 0, A[0]);
                    read_frags(stg, 1, A[1]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) A[0][i] = A[1][i] = x_frag(0, false, 0);
                }
                const bf16x8 wfn = wf_at(stg);
                const bool act_next = DUAL ? (kt + 1 == nk || kt + 1 < nact) : (CRITIC && kt + 1 < nk && kt + 1 >= p.nfs);
                const int xbuf = kt + 1 == nk ? xb ^ 1 : xb;
                const bf16x8 xf[2] = {x_frag(xbuf, act_next, 0), x_frag(xbuf, act_next, 1)};
                // (scalar, out of line; read from the next prepare phase on. Peeling step nk - 2 out of the loop instead -- no branch in
                //  the loop body -- made the kernel 7-9 % SLOWER, r06: three copies of the step, 100 B more scratch)
                if (__builtin_expect(kt + 2 == nk && wv < NRQ, 0)) build_x(tile + gridDim.x, xb ^ 1);
                if (!FW_DBG(1)) {
                    int kc = kt + FSTG - 1;
                    kc -= kc >= nk ? nk : 0;
                    kc -= kc >= nk ? nk : 0;
                    const int sd = (stg + FSTG - 1) & (FSTG - 1);
                    dma(sd, kc, 0), dma(sd, kc, 1);
                }
                // (vm operations complete in issue order per kind only: the tile boundary drains the stream before its stores, so
                // the first two steps of a tile have nothing to wait for, and from step 2 on the counted wait also retires those stores)
                if (WAIT)
                    __builtin_amdgcn_s_waitcnt(0x0076);  // vmcnt(6) lgkmcnt(0)
                else
                    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
                FW_STAMP(0);
                if (!FW_DBG(4)) __builtin_amdgcn_s_barrier();
                FW_STAMP(1);
                __builtin_amdgcn_sched_barrier(0);
                // ================= multiply(kt)
                __builtin_amdgcn_s_setprio(3);
                // the first layer of chunk kt + 1 first; its relu / pack (32 VALU) rides in the gaps of the 16 accumulating MFMAs
                f32x16 p1n[2];
                bf16x8 bnx[2][2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) p1n[rt] = mfma(wfn, xf[rt], zero16);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) acc[rt][ct] = mfma(A[ks][ct], bfr[rt][ks], acc[rt][ct]);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) pack(p1n[rt], 0, bnx[rt][0]), pack(p1n[rt], 1, bnx[rt][1]);
                // (r06b, measured and not kept: row tile 1's first layer half a phase behind row tile 0's -- the two tiles not live at the
                //  same time, 16 registers -- took 30-50 B of scratch out of every variant and made every variant 4-9 % slower)
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#pragma unroll
                for (int i = 0; i < 11; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) bfr[rt][0] = bnx[rt][0], bfr[rt][1] = bnx[rt][1];
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (WAIT) __builtin_amdgcn_s_waitcnt(0x0F76);  // vmcnt(6)
                FW_STAMP(2);
                if (!FW_DBG(4)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                stg = (stg + 1) & (FSTG - 1);
            };
            step(0, std::false_type{});
            step(1, std::false_type{});
            for (int kt = 2; kt < nk; ++kt) step(kt, std::true_type{});
            xb ^= 1;
            // (this group is in a prepare phase here: the tile's epilogue and the next tile's start share it with prepare(0))
            FW_STAMP(3);
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): chunks 0 and 1 of the next period have landed; nothing older than the stores
            if (grp == 0) __builtin_amdgcn_s_barrier();  // level (pairs with group 1's last barrier of the loop)
            FW_STAMP(4);
#ifdef AVD_FW_STAMP
            // (when do the workgroups reach their tile boundaries? the 100 MHz real-time counter at the start of this workgroup's epilogues
            //  of pair 0, and at their end: [256 workgroups][8 tiles][2])
            if (p.stamp && pair == 0 && wave == 4 && lane == 0 && tile / (int)gridDim.x < 8)
                p.stamp[64 + (blockIdx.x * 8 + tile / (int)gridDim.x) * 2] = __builtin_amdgcn_s_memrealtime();
#endif
            float rwv[2] = {1.f, 1.f};
            if (DUAL && p.rw) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const long n = (long)tile * GR + rh * 64 + 32 * rt + r;
                    rwv[rt] = p.rw[(long)set * p.Ns + (n < p.Ns ? n : p.Ns - 1)];
                }
            }
            if (!FW_DBG(8)) epilogue(tile);
            dual_tail(tile, rwv);
            if constexpr (DUAL) {
                // the next tile's first B operand again (the last step built it too): recomputed here it is not live across the tail -- 16
                // registers the allocator otherwise finds by spilling the loop's state. Chunk 0's fragments are still in the stage of chunk nk - 1.
                const bf16x8 wf0 = wf_at((stg + FSTG - 1) & (FSTG - 1));
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const f32x16 p1 = mfma(wf0, x_frag(xb, true, rt), zero16);
                    pack(p1, 0, bfr[rt][0]), pack(p1, 1, bfr[rt][1]);
                }
            }
            FW_STAMP(5);
#ifdef AVD_FW_STAMP
            if (p.stamp && pair == 0 && wave == 4 && lane == 0 && tile / (int)gridDim.x < 8)
                p.stamp[64 + (blockIdx.x * 8 + tile / (int)gridDim.x) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
            z_flush(prev);
            prev = tile;
            init_acc();
            FW_STAMP(6);
        }
#ifdef AVD_FW_STAMP
        if (blockIdx.x == 16 && lane == 0 && p.stamp && pair == 0)
            for (int i = 0; i < 8; ++i) p.stamp[wave * 8 + i] = facc[i];
#endif
        // ---- the last output-layer sums (the groups are level)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the stream (vmcnt 0)
        __syncthreads();
        z_flush(prev);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the stream before the next pair re-uses the stages / the LDS is released
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the second layer in the same style, rank-one form (r06):
//   G[f][c] = inv[f] sum_n (d[n] y1[n][f]) mask[n][c]          dW2 = cf[c] (G + sh[f] S2[c])  (w2_post_kernel)
// with the first-layer activations y1 GENERATED per 32-row chunk (one MFMA per 32 rows x 32 features from the chunk's prepared
// fragments, whose inputs and bias slots carry |d|; the sign of d is XORed into the packed relu'd tile) instead of read from a
// transposed activation matrix, and the relu mask of the second layer -- the exact bf16 image {1, 0}, row-major, stored by the
// actor's forward kernel in place of its activations and by fw::fwd_delta_kernel beside critic(s, mu) (taking the mask from the
// sign of the stored activations HERE, 64 packed VALU per step, cost dw_gen 60 %: its prepare phase has no VALU issue slots to
// spare; a second image from the critic's forward kernel cost that kernel 1.2 ms) -- streamed L2 -> LDS through the four stages
// and transposed on the way out of LDS. Transposed product
// D[c][f] = sum_n mask[n][c] (d y1)[n][f]: A = mask fragments (LDS), B = relu'd first-layer tile (lane = feature, registers = rows).
// A work item = 128 features x 512 columns x one eighth (or less) of a set's rows, dealt over the persistent workgroups so that
// the feature blocks of a (set, column block, row range) run side by side on the same stream; f32 atomics at the end of an item.
// S2[c] = sum_n d[n] mask[n][c] rides along: thread t of the item with feature block fb < 32 / RPB adds rows RPB fb .. + RPB of every
// chunk for column t (the stream's feature blocks share the rows between them). Same ping-pong of the two wave groups as
// fwd_gen_kernel.
struct DwP {
    const unsigned char* aux;  // [sets][Np / 32 + 1][AUX_REC]: aux_pack_kernel's records
    long setAux;               // bytes per set
    const bf16x8* wf1;         // [sets][nft][64] (the same fragments: lane = feature)
    int nft, nfs;
    const bf16* ZT;            // [sets][Np][ldz]: the relu mask {1, 0} of the second layer, row-major (ldz = H2)
    long setZT, ldz;
    const float* inv;          // [sets][setTab] first-layer BN table (feature index)
    long setTab;
    float* dW;                 // [sets][setW] + offset of W2: [K][H2], receives G
    long setW;
    float* s2;                 // [sets][H2], zeroed by the caller
    int Ns, Np, H2, K, n_sets, nsplit;
};

// stage = the chunk's 32 mask rows (512 columns each, row stride 1088 B: the transposed reads below are conflict-free) + the record
// of the NEXT chunk's fragments
constexpr int DW_ROWB = FC * 2 + 64, DW_STG = FK * DW_ROWB + AUX_LDS;
// RPB = mask rows of every 32-row chunk an S2-summing feature block takes: 32 / min(8, state feature blocks) -- 4 at H1 >= 1024,
// 8 at H1 = 512, 16 at H1 = 256 (the first 32 / RPB blocks of a stream share the rows; a compile-time trip count: a runtime loop
// would put a branch into the prepare phase)
template <bool CRITIC, int RPB>
__global__ __launch_bounds__(FT) __attribute__((amdgpu_waves_per_eu(2, 2))) void dw_gen_kernel(DwP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int rh = wave & 1, cq = wave >> 1, grp = wave >> 2;  // 64-feature half, 128-column quarter, ping-pong group
    const int ncb = p.H2 / FC, nfb = (p.nft * 32 + 127) / 128;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mfma = [](bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); };
    // relu + bf16 of half a first-layer tile, the rows' signs on top (one v_xor per pair: a relu'd bf16 is non-negative)
    auto pack = [&](const f32x16& p1, int ks, const wide::u32x4& sg, bf16x8& b) {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        wide::u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 f = {p1[8 * ks + 2 * i], p1[8 * ks + 2 * i + 1]};
            const bf16x2 v = __builtin_convertvector(f, bf16x2);
            const s16x2 z = {0, 0};
            o[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z)) ^ sg[i];
        }
        b = __builtin_bit_cast(bf16x8, o);
    };
    // A fragments (lane = column of the mask, k = rows) from the row-major chunk by ds_read_b64_tr_b16: per 16 lanes a block of 4 rows x
    // 16 columns comes back column-major; lane 4 q + pp of the group addresses row q, columns 4 pp .. 4 pp + 3. Two reads per fragment:
    // element 4 g + j of lane half h is row 8 g + 4 h + j of the k-step -- the k order of the generated operand as it stands.
    const int tq = (lane & 15) >> 2, tp = lane & 3, tmb = 16 * ((lane >> 4) & 1);
    const int rdt = (4 * h + tq) * DW_ROWB + (128 * cq + tmb + 4 * tp) * 2;
    auto read_frags = [&](int stg, int ks, bf16x8 (&a)[4]) {
        const unsigned char* b = smem_raw + stg * DW_STG + rdt + 16 * ks * DW_ROWB;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                // (inline asm, not __builtin_amdgcn_ds_read_tr16_b64: behind the builtin hipcc puts s_waitcnt vmcnt(0) -- it cannot tell
                //  the read from the LDS-DMA writes in flight -- and the stream ran one chunk deep instead of three. The explicit
                //  lgkmcnt(0) before the phase barrier covers these reads: their first use is behind it.)
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                u32x2_t t;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(b + 8 * g * DW_ROWB)), "i"(ct * 64));
                const bf16x4 tb = __builtin_bit_cast(bf16x4, t);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[ct][4 * g + j] = tb[j];
            }
    };
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int rows_per = p.Np / p.nsplit, nk = rows_per / FK;  // chunks of an item (Np is a multiple of 256, nsplit of 8: whole chunks)
    // A stream = (set, column block, row range); its nfb feature blocks are the items that read it. Workgroup ids go round the
    // eight XCDs, and every XCD has its own L2: XCD x takes streams x, x + 8, .. and its workgroups walk that list stream by
    // stream, so the feature blocks of a stream run side by side behind ONE L2 (dealt by plain item number they landed on all
    // eight, and every XCD fetched every stream from memory: 2.7 GB became ~20)
    const int nstreams = p.n_sets * ncb * p.nsplit, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    for (int li = slot;; li += nslots) {
        const int stream = xcd + 8 * (li / nfb), fb = li % nfb;
        if (stream >= nstreams) break;
        const int split = stream % p.nsplit, rest2 = stream / p.nsplit, cb = rest2 % ncb, set = rest2 / ncb;
        const int row_base = split * rows_per;
        // the wave's two feature tiles and their (fixed) first-layer fragments: B operand, lane = feature
        const int t0 = fb * 4 + rh * 2;
        const bool action = CRITIC && t0 >= p.nfs;
        bf16x8 wf[2];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            const bf16 zb = (bf16)0.f;
            wf[ft] = (bf16x8){zb, zb, zb, zb, zb, zb, zb, zb};
            if (t0 + ft < p.nft) wf[ft] = p.wf1[((long)set * p.nft + t0 + ft) * 64 + lane];
        }
        // stream: rows 4 w .. 4 w + 3 of the chunk's mask rows (512 columns = 1 KiB per instruction) + this wave's KiB of the
        // record that holds the NEXT chunk's fragments: 5 wave-instructions per chunk
        const char* ubw = (const char*)(p.ZT + (long)set * p.setZT + (long)(row_base + 4 * wv) * p.ldz + FC * cb);
        const unsigned char* rec0 = p.aux + (long)set * p.setAux + (long)(row_base / FK) * AUX_REC;  // record of the item's chunk 0
        const unsigned vow = (unsigned)lane * 16u;
        auto dma = [&](int stg, int kc) {
            unsigned char* l = smem_raw + stg * DW_STG;
            unsigned vw = vow;
            asm volatile("" : "+v"(vw));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(ubw + ((long)kc * FK + i) * p.ldz * 2 + vw), (lptr_t)(l + (4 * wv + i) * DW_ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(rec0 + (long)(kc + 1) * AUX_REC + 256 * wv + vw), (lptr_t)(l + FK * DW_ROWB + 256 * wv), 16, 0, 0);
        };
        constexpr int NDMA = 5;
        const int fofs = FK * DW_ROWB + (action ? AUX_ACT : 0) + lane * 16, sofs = FK * DW_ROWB + AUX_SGN + h * 32;
        __syncthreads();  // (the previous item's LDS reads are done)
#pragma unroll
        for (int c = 0; c < FSTG - 1; ++c) dma(c, c < nk ? c : nk - 1);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();
        f32x16 acc[2][4];  // [feature tile][column tile]: lane = feature, register 4 g + j <-> column 128 cq + 32 ct + 8 g + 4 h + j
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ft][ct] = zero16;
        bf16x8 bfr[2][2];  // relu'd first layer of the chunk: [feature tile][k-step]
        {
            // chunk 0's own fragments are not in a stage (a stage carries the NEXT chunk's): read them from its record
            const bf16x8 x0 = *(const bf16x8*)(rec0 + (action ? AUX_ACT : 0) + lane * 16);
            const wide::u32x4 sg0 = *(const wide::u32x4*)(rec0 + AUX_SGN + h * 32), sg1 = *(const wide::u32x4*)(rec0 + AUX_SGN + h * 32 + 16);
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                const f32x16 p1 = mfma(x0, wf[ft], zero16);
                pack(p1, 0, sg0, bfr[ft][0]), pack(p1, 1, sg1, bfr[ft][1]);
            }
        }
        if (grp == 1) {  // one phase behind
            __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * NDMA));
            __builtin_amdgcn_s_barrier();
        }
        int stg = 0;
        float s2acc = 0.f;  // sum_n d[n] mask[n][column tid] over this item's share of the rows
        // (a ninth feature block -- the critic's -- has no rows left to sum: it goes through the same motions on rows 0..3 and keeps the
        //  result to itself. Without them it ran a few per cent ahead of the stream's other blocks, out of the window in which they
        //  share the chunks in L2: 8.2 GB fetched for 2.7)
        constexpr int NSH = 32 / RPB;
        const bool s2on = fb < NSH;
        const int fbr = fb & (NSH - 1);
        auto step = [&](int kt, auto wait_c) {
            constexpr bool WAIT = decltype(wait_c)::value;
            // ================= prepare(kt)
            bf16x8 A[2][4];
            const unsigned char* l = smem_raw + stg * DW_STG;
            // fragment + sign words of chunk kt + 1 (its record rides in this stage); the four mask elements and seeds of the S2 sum.
            // (All requested BEFORE the transposed fragment reads -- inline asm the compiler cannot count: behind them, every one of
            //  these reads got its own lgkmcnt(0).)
            const bf16x8 xn = *(const bf16x8*)(l + fofs);
            const wide::u32x4 sg0 = *(const wide::u32x4*)(l + sofs), sg1 = *(const wide::u32x4*)(l + sofs + 16);
            wide::f32x4 dv[RPB / 4];
#pragma unroll
            for (int q = 0; q < RPB / 4; ++q) dv[q] = *(const wide::f32x4*)(l + FK * DW_ROWB + AUX_D + 4 * RPB * fbr + 16 * q);
            unsigned mrow[RPB];
#pragma unroll
            for (int i = 0; i < RPB; ++i) mrow[i] = *(const unsigned short*)(l + (RPB * fbr + i) * DW_ROWB + tid * 2);
            read_frags(stg, 0, A[0]);
            read_frags(stg, 1, A[1]);
#pragma unroll
            for (int i = 0; i < RPB; ++i) s2acc = fmaf(__uint_as_float(mrow[i] << 16), dv[i >> 2][i & 3], s2acc);
            {
                int kc = kt + FSTG - 1;
                kc = kc < nk ? kc : nk - 1;  // (tail: harmless re-loads keep the vmcnt arithmetic uniform)
                dma((stg + FSTG - 1) & (FSTG - 1), kc);
            }
            if (WAIT)
                __builtin_amdgcn_s_waitcnt(0x0070 | (2 * NDMA));  // vmcnt(2 chunks) lgkmcnt(0)
            else
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ================= multiply(kt)
            __builtin_amdgcn_s_setprio(3);
            f32x16 p1n[2];
            bf16x8 bnx[2][2];
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) p1n[ft] = mfma(xn, wf[ft], zero16);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft) acc[ft][ct] = mfma(A[ks][ct], bfr[ft][ks], acc[ft][ct]);
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) pack(p1n[ft], 0, sg0, bnx[ft][0]), pack(p1n[ft], 1, sg1, bnx[ft][1]);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) bfr[ft][0] = bnx[ft][0], bfr[ft][1] = bnx[ft][1];
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (WAIT) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * NDMA));
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stg = (stg + 1) & (FSTG - 1);
        };
        step(0, std::false_type{});
        step(1, std::false_type{});
        for (int kt = 2; kt < nk; ++kt) step(kt, std::true_type{});
        if (grp == 0) {  // make up the phase this group is ahead
            __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * NDMA));
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the stream
        if (s2on) atomicAdd(p.s2 + (long)set * p.H2 + FC * cb + tid, s2acc);
        // ---- G[f][c] += inv[f] acc, f32 atomics. The accumulator holds a feature per lane (rows of dW2, 4 KB apart): added as it
        // stands, every atomic instruction touched 32 cache lines with two floats each, and the epilogues of an item cost as much
        // as 260 of its 1024 steps (0.23 ms: a fifth of the kernel). Each 32 x 32 tile goes through a private 4.5 KB of LDS
        // instead and is added with a COLUMN per lane: 2 lines x 32 floats per instruction.
        __syncthreads();  // (the stages are scratch from here: every wave is past its last fragment / record read)
        float* tr = (float*)(smem_raw + wv * (32 * 36 * 4));  // [32 features][36]: 16-byte rows, private to the wave
        const int tc = lane & 31, tf = lane >> 5;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            const int f = 32 * (t0 + ft) + r;
            const float iv = f < p.K ? p.inv[(long)set * p.setTab + f] : 0.f;
            float* o = p.dW + (long)set * p.setW + (long)(32 * (t0 + ft) + tf) * p.H2 + FC * cb + 128 * cq + tc;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = iv * acc[ft][ct][4 * g + j];
                    *(f32x4*)(tr + r * 36 + 8 * g + 4 * h) = v;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {  // features 2 i + tf of the tile, column tc
                    const float v = tr[(2 * i + tf) * 36 + tc];
                    if (32 * (t0 + ft) + 2 * i + tf < p.K) atomicAdd(o + (long)(2 * i) * p.H2 + 32 * ct, v);
                }
            }
        }
        __syncthreads();  // (before the next item's stream lands in the stages)
    }
}

// Input gradient of the second layer + everything behind it, without an activation or gradient matrix of the first layer:
//   dy[n][f] = d[n] sum_c mask[n][c] (cf[c] W2[f][c])   (MFMA, K = H2; D[row][feature]: lane = feature, registers = rows; the row
//                                                        factor d[n] multiplies the accumulator tile in the epilogue: r06, rank-one form)
//   p[n][f]  = x[n] . W1[:, f] + b1[f]              (regenerated: one MFMA per 32 x 32 tile, same layout)
//   dgamma1[f] = rs (S1 - mean S2), dbeta1[f] = S2, db1[f] = inv S3, dW1[s][f] = inv S4[s]   with the per-feature sums over rows
//   S1 = sum dy relu(p), S2 = sum dy, S3 = sum [p > 0] dy, S4[s] = sum x[n][s] [p > 0] dy
// (S1 as sum [p > 0] dy p; S3 rides on S4's product as a row of ones in x^T: one compare, one select, one fma, one add per element)
// -- sums over the REGISTERS of a lane (rows), kept in registers across all row tiles of a workgroup and added to the gradient
// slab once at the end (f32 atomics). Replaces the input-gradient GEMM with its BN / ReLU epilogue (which read C and wrote dZ1),
// l1_fwd (C), l1_grads and the BN flush. A workgroup = 256 rows x 256 features per step of 32 c: both operands stream through
// LDS (dZ2 rows from memory, the W2 block from L2), 8 waves = 4 row quarters x 2 feature halves (the two ping-pong groups).
// Workgroup ids -> XCD x = id % 8, slot id / 8 -> (row group, feature block): the feature blocks of a row group share an L2.
struct DxP {
    const float* X;     // [sets][Ns][4]
    long setX;
    const float* act;   // critic: [sets][setAct]
    long setAct;
    const bf16x8* wf1;  // [sets][nft][64]
    int nft, nfs;
    const bf16* dZ;     // [sets][Np][H2]: the relu mask of the second layer as a bf16 image {1, 0} (rank-one form: dZ2 = d (x) cf (.) mask)
    long setDZ;
    const float* d;     // [sets][setD]: the rows' loss seeds, zeros in the padding rows (aux_pack_kernel's dclean)
    long setD;
    const bf16* Wn;     // [sets][>= 256 nfb][H2]: bf16(cf[c] W2[f][c]), row = feature
    long setWn;
    const float *inv, *rs, *mean;  // [sets][setTab]
    long setTab;
    float* g;           // [sets][setG] gradient slab of the net
    long setG;
    int w_off[2], b_off[2], g_off[2], be_off[2];  // [0] state branch, [1] action branch
    int Ns, Np, H2, H1, Ha, n_sets, nfb, groups_per_xcd;
    int abl;  // diagnostic build only (AVD_WIDE_DX_ABL, results wrong by design): 1 = every mask tile fetched from row tile 0 (served by L2)
};

constexpr int DX_STG = 2 * 256 * FK * 2, DX_NS = 3;  // A tile + B tile; three stages (two chunks in flight)
// raw inputs three tiles deep; the running sums of a wave's 128 features (7 per feature) live in LDS between tile epilogues
// + the first-layer fragments of every wave's four feature tiles (per set: the epilogue read them from memory one after the other).
// CRITIC (r06): a stage also carries 32 of the critic's 64 (padded) action features x 32 c (2 KiB), the raw buffers the actions.
template <bool CRITIC>
struct DxL {
    static constexpr int STG = DX_STG + (CRITIC ? 32 * FK * 2 : 0);
    static constexpr int X = DX_NS * STG, A = X + 3 * 256 * 16, ACT = A + 3 * 256 * 4, S = ACT + (CRITIC ? 3 * 256 * 4 : 0),
                         W = S + 8 * 28 * 32 * 4, TOTAL = W + 2 * 4 * 64 * 16;
};

template <bool CRITIC>
__global__ __launch_bounds__(FT) __attribute__((amdgpu_waves_per_eu(2, 2))) void dx_gen_kernel(DxP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef DxL<CRITIC> LY;
    constexpr int DXL_X = LY::X, DXL_A = LY::A, DXL_ACT = LY::ACT, DXL_S = LY::S, DXL_W = LY::W, STG_B = LY::STG;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int rq = wave & 3, fh = wave >> 2, grp = fh;  // 64-row quarter, 128-feature half = ping-pong group
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mfma = [](bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); };
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int nk = p.H2 / FK, ntile = p.Np / 256;
    // slot -> (row group, feature block)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int rg = slot / p.nfb, fb = slot - rg * p.nfb;
    if (rg >= p.groups_per_xcd) return;
    const int ngroups = 8 * p.groups_per_xcd, group = xcd * p.groups_per_xcd + rg;
    const int total = p.n_sets * ntile;  // (set, row tile) pairs, dealt in contiguous runs to the row groups
    const int t_begin = (int)((long)total * group / ngroups), t_end = (int)((long)total * (group + 1) / ngroups);
    if (t_begin >= t_end) return;
    // fragment read offsets (64-byte image rows, 16-byte pieces swizzled by bits 2..3 of the row)
    const int sw = (r >> 2) & 3;
    const int ra0 = ((64 * rq + r) * FK + (((0 + h) ^ sw) << 3)) * 2, ra1 = ((64 * rq + r) * FK + (((2 + h) ^ sw) << 3)) * 2;
    const int rb0 = 256 * FK * 2 + ((128 * fh + r) * FK + (((0 + h) ^ sw) << 3)) * 2, rb1 = 256 * FK * 2 + ((128 * fh + r) * FK + (((2 + h) ^ sw) << 3)) * 2;
    const unsigned vo_a = (unsigned)(((lane >> 2) * p.H2 + (((lane & 3) ^ ((lane >> 4) & 3)) << 3)) * 2);
    // CRITIC: the critic's 64 (48 + padding) action features ride along as ONE more MFMA per wave and step. A row group's four feature
    // blocks x two feature halves are eight wave classes for the eight (action tile, 32-row tile, k-step) products of a 64-row quarter:
    // class (fb, fh) takes action tile fb & 1, row tile fb >> 1 and k-step fh of EVERY chunk -- a partial product over half of c, which is
    // all the epilogue needs: everything behind dy is linear in dy (the mask comes from the regenerated first layer), so the two
    // partial sums of a (rows, feature) pair are folded independently and meet in the gradient slab. (Replaces, for H1 = 1024, the
    // 64-column GEMM over the mask with its activation / gradient matrices: l1_fwd, gemm_bt<EpiDx>, l1_grads, bn1_flush: 0.93 ms.)
    const int at = fb & 1, rtx = (fb >> 1) & 1;
    const int rax = (fh ? ra1 : ra0) + rtx * 32 * FK * 2, rbx = 2 * 256 * FK * 2 + (r * FK + (((2 * fh + h) ^ sw) << 3)) * 2;
    float asum[4] = {0.f, 0.f, 0.f, 0.f};  // the action tile's running sums S1, S2, S3, S4[0] (halves already added; lanes h == 0)
    bf16x8 wfa;                            // its first-layer fragment

    // running sums per feature, halves already added: [wave][7 ft-major quantities x 4 ft][32 lanes] floats in LDS
    float* sums = (float*)(smem_raw + DXL_S) + wave * 28 * 32 + r;
    auto clear_sums = [&]() {
        if (h == 0)
#pragma unroll
            for (int q = 0; q < 28; ++q) sums[q * 32] = 0.f;
    };
    clear_sums();
    const int f0 = fb * 256 + fh * 128;               // the wave's first feature
    const bool action = CRITIC && f0 >= p.H1;          // (H1 is a multiple of 128: a wave's features lie on one side)
    int set = t_begin / ntile;
    // per-set state: first-layer fragments (B operand of the regenerated first layer), stream bases
    const char *ub_a = nullptr, *ub_b = nullptr, *ub_x = nullptr;
    auto load_set = [&]() {
        ub_a = (const char*)(p.dZ + (long)set * p.setDZ + (long)(32 * wv) * p.H2);             // + row tile, chunk
        ub_b = (const char*)(p.Wn + (long)set * p.setWn + (long)(fb * 256 + 32 * wv) * p.H2);  // + chunk
        if constexpr (CRITIC) {  // rows H1 + 32 at .. + 32 of Wn, 16 per instruction: waves 2..7 repeat waves 0 and 1 (uniform counts)
            ub_x = (const char*)(p.Wn + (long)set * p.setWn + (long)(p.H1 + 32 * at + 16 * (wv & 1)) * p.H2);
            wfa = p.wf1[((long)set * p.nft + p.nfs + at) * 64 + lane];
        }
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {  // the four first-layer fragments of the wave's feature half (its four row quarters write the same bytes)
            const bf16 zb = (bf16)0.f;
            bf16x8 v = {zb, zb, zb, zb, zb, zb, zb, zb};
            const int wt = (fb * 256 + fh * 128) / 32 + ft;
            if (wt < p.nft) v = p.wf1[((long)set * p.nft + wt) * 64 + lane];
            *(bf16x8*)(smem_raw + DXL_W + ((fh * 4 + ft) * 64 + lane) * 16) = v;
        }
    };
    // chunk kc of row tile tl into stage stg: wave w fills image rows [32 w, 32 w + 32) of both tiles (2 + 2 instructions)
    auto dma = [&](int stg, int tl, int kc) {
        unsigned char* l = smem_raw + stg * STG_B;
        unsigned vw = vo_a;
        asm volatile("" : "+v"(vw));
#ifdef AVD_DIAG
        if (p.abl & 1) tl = 0;
#endif
        const long ta = (long)tl * 256 * p.H2 * 2 + (long)kc * FK * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(ub_a + ta + (long)i * 32 * p.H2 + vw), (lptr_t)(l + (32 * wv + 16 * i) * FK * 2), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(ub_b + (long)kc * FK * 2 + (long)i * 32 * p.H2 + vw), (lptr_t)(l + 256 * FK * 2 + (32 * wv + 16 * i) * FK * 2), 16, 0, 0);
        }
        if constexpr (CRITIC)
            __builtin_amdgcn_global_load_lds((gptr_t)(ub_x + (long)kc * FK * 2 + vw), (lptr_t)(l + 2 * 256 * FK * 2 + 16 * (wv & 1) * FK * 2), 16, 0, 0);
    };
    // (r06, measured and not kept: the mask rows come from HBM and three stages give a chunk about one step to arrive -- with every tile
    //  fetch redirected to one tile, diagnostic build AVD_WIDE_DX_ABL=1, the kernel runs 8-12 % faster. An L2 prefetch of the chunk four
    //  steps ahead as one more LDS-DMA instruction per wave and step -- 4 bytes per lane of its 32 rows into a scratch word, counted
    //  behind the refills -- made it 23 % SLOWER: 32 distinct lines per instruction on the in-order vm queue. A fourth stage does not
    //  fit the LDS beside the running sums.)
    // raw inputs of row tile tl into buffer b: 64 rows per instruction, waves 4..7 repeat waves 0..3 (uniform counts)
    auto dma_x = [&](int tl, int b) {
        int n = tl * 256 + 64 * (wv & 3) + lane;
        n = n < p.Ns ? n : p.Ns - 1;
        __builtin_amdgcn_global_load_lds((gptr_t)(p.X + (long)set * p.setX + (long)n * 4), (lptr_t)(smem_raw + DXL_X + (b * 256 + 64 * (wv & 3)) * 16), 16, 0, 0);
        // (the seed comes unclamped: its padding rows hold zeros)
        __builtin_amdgcn_global_load_lds((gptr_t)(p.d + (long)set * p.setD + tl * 256 + 64 * (wv & 3) + lane), (lptr_t)(smem_raw + DXL_A + (b * 256 + 64 * (wv & 3)) * 4), 4, 0, 0);
        if constexpr (CRITIC)
            __builtin_amdgcn_global_load_lds((gptr_t)(p.act + (long)set * p.setAct + n), (lptr_t)(smem_raw + DXL_ACT + (b * 256 + 64 * (wv & 3)) * 4), 4, 0, 0);
    };
    constexpr int NDMA = CRITIC ? 5 : 4;  // per chunk; the counted waits leave ONE chunk in flight (three stages)
    // the sums of the finished set go to the gradient slab
    auto flush = [&]() {
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const float s1 = sums[(7 * ft + 0) * 32], s2 = sums[(7 * ft + 1) * 32], s3 = sums[(7 * ft + 2) * 32];
            const float s4[4] = {sums[(7 * ft + 3) * 32], sums[(7 * ft + 4) * 32], sums[(7 * ft + 5) * 32], sums[(7 * ft + 6) * 32]};
            const int f = f0 + 32 * ft + r, br = action ? 1 : 0, fl = action ? f - p.H1 : f, Hn = action ? p.Ha : p.H1;
            if (h == 0 && fl < Hn) {
                const long tb = (long)set * p.setTab + f;
                const float iv = p.inv[tb], rsv = p.rs[tb], mv = p.mean[tb];
                float* g = p.g + (long)set * p.setG;
                atomicAdd(g + p.g_off[br] + fl, rsv * (s1 - mv * s2));
                atomicAdd(g + p.be_off[br] + fl, s2);
                atomicAdd(g + p.b_off[br] + fl, iv * s3);
                const int ns = action ? 1 : 4;  // (action: input 0 is the action, the others are zeros)
                for (int sI = 0; sI < ns; ++sI) atomicAdd(g + p.w_off[br] + (long)sI * Hn + fl, iv * s4[sI]);
            }
        }
        if constexpr (CRITIC) {
            const int fl = 32 * at + r;
            if (h == 0 && fl < p.Ha) {
                const long tb = (long)set * p.setTab + p.H1 + fl;
                const float iv = p.inv[tb], rsv = p.rs[tb], mv = p.mean[tb];
                float* g = p.g + (long)set * p.setG;
                atomicAdd(g + p.g_off[1] + fl, rsv * (asum[0] - mv * asum[1]));
                atomicAdd(g + p.be_off[1] + fl, asum[1]);
                atomicAdd(g + p.b_off[1] + fl, iv * asum[2]);
                atomicAdd(g + p.w_off[1] + fl, iv * asum[3]);
            }
            asum[0] = asum[1] = asum[2] = asum[3] = 0.f;
        }
        clear_sums();
    };
    load_set();
    // ---- start the stream: chunks 0 .. 2 of the first row tile
    {
        const int tl = t_begin - set * ntile;
        dma_x(tl, 0);
#pragma unroll
        for (int c = 0; c < DX_NS - 1; ++c) dma(c, tl, c);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    int stg = 0, xb = 0;
    f32x16 acc[2][4], accx = zero16;
    for (int t = t_begin; t < t_end; ++t) {
        // The two groups run the K loop a phase apart and the tile epilogue level: group 1 drops a phase behind here and group 0
        // waits for it after the loop. (With the offset kept across tiles each group's epilogue ran beside the other's LAST or FIRST
        // multiply phase only -- 12 k cycles of latency-bound work, twice per tile, with the matrix pipes idle.)
        if (grp == 1) __builtin_amdgcn_s_barrier();
        const int tl = t - set * ntile;
        // where the stream goes after this tile: the next tile of the run (same set or the next one), or nowhere
        const int tn = t + 1 < t_end ? t + 1 : t;
        const int set_n = tn / ntile, tln = tn - set_n * ntile;
        const bool same_set = set_n == set;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) acc[rt][ft] = zero16;
        if constexpr (CRITIC) accx = zero16;
        auto step = [&](int kt, auto wait_c, auto tail_c) {
            constexpr bool WAIT = decltype(wait_c)::value, TAIL = decltype(tail_c)::value;
            // ================= prepare(kt)
            bf16x8 A[2][2], B[2][4];
            const unsigned char* l = smem_raw + stg * STG_B;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) A[0][rt] = *(const bf16x8*)(l + ra0 + rt * 32 * FK * 2), A[1][rt] = *(const bf16x8*)(l + ra1 + rt * 32 * FK * 2);
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) B[0][ft] = *(const bf16x8*)(l + rb0 + ft * 32 * FK * 2), B[1][ft] = *(const bf16x8*)(l + rb1 + ft * 32 * FK * 2);
            bf16x8 Ax, Bx;
            if constexpr (CRITIC) Ax = *(const bf16x8*)(l + rax), Bx = *(const bf16x8*)(l + rbx);
            {   // refill: chunk kt + 3 of this tile, or the first chunks of the next tile (a new set only after the flush: see below)
                const int kc = kt + DX_NS - 1, sd = (stg + DX_NS - 1) % DX_NS;
                if (!TAIL)  // (the loop body proper carries no branch: the last two steps are peeled below)
                    dma(sd, tl, kc);
                else if (same_set)
                    dma(sd, tln, kc - nk);
                else
                    dma(sd, tl, nk - 1);  // (harmless re-load: uniform counts; the next set's stream starts at the tile boundary)
            }
            if (WAIT)
                __builtin_amdgcn_s_waitcnt(0x0070 | NDMA);
            else
                __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ================= multiply(kt)
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ft = 0; ft < 4; ++ft)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) acc[rt][ft] = mfma(A[ks][rt], B[ks][ft], acc[rt][ft]);
            if constexpr (CRITIC) accx = mfma(Ax, Bx, accx);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (WAIT) __builtin_amdgcn_s_waitcnt(0x0F70 | NDMA);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stg = stg + 1 == DX_NS ? 0 : stg + 1;
        };
        // (three buffers: the other group is still folding the previous tile while this one requests the tile after next)
        const int xbn = xb == 2 ? 0 : xb + 1;
        if (same_set && tn != t) dma_x(tln, xbn);
        // (three stages = two chunks in flight: only chunks 0 and 1 of a tile have landed at its start. Step 1 therefore waits
        //  like every later step -- it must retire chunk 2, requested in step 0 and read in step 2. With both first steps unwaited, as
        //  in the four-stage kernels, step 2 read whatever had arrived: right almost always, garbage from the previous tenant of the
        //  stage once in a few hundred launches: tools/determinism_c5.py)
        step(0, std::false_type{}, std::false_type{});  // (nk >= 4: H2 >= 128)
        step(1, std::true_type{}, std::false_type{});
        for (int kt = 2; kt < nk - 2; ++kt) step(kt, std::true_type{}, std::false_type{});
        step(nk - 2, std::true_type{}, std::true_type{});
        step(nk - 1, std::true_type{}, std::true_type{});
        __builtin_amdgcn_s_waitcnt(0x0F70);  // drain: nothing older than what follows
        if (grp == 0) __builtin_amdgcn_s_barrier();  // level again (pairs with group 1's last barrier of the loop)
        // ---- tile epilogue (a prepare phase of this group): regenerate the first layer, fold the tile into the sums.
        // S1..S3 on the VALU (5 per element); S4[s] = sum_rows x[row][s] mdy[row][f] is one more product over the rows: A = x^T as
        // bf16 hi (m = s) and lo (m = 4 + s) rows in the accumulator's row order, B = bf16(mdy) as it stands (lane = feature).
        auto raw_x1 = [&](int row, int sI) {  // input sI of a row (the kernel serves the state features; action features: dx() + l1_grads)
            return *(const float*)(smem_raw + DXL_X + (xb * 256 + row) * 16 + 4 * sI);
        };
        bf16x8 xT[2][2];  // [rt][ks]: lane m = lane & 31 (m < 4: hi of input m, m < 8: lo of input m - 4, else 0), element e <-> row below
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16 zb = (bf16)0.f;
                bf16x8 v = {zb, zb, zb, zb, zb, zb, zb, zb};
                if (r < 8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int row = 64 * rq + 32 * rt + 16 * ks + 8 * (e >> 2) + 4 * h + (e & 3);
                        const float xs = raw_x1(row, r & 3);
                        const bf16 hi = (bf16)xs;
                        v[e] = r < 4 ? hi : (bf16)(xs - (float)hi);
                    }
                } else if (r == 8) {  // m = 8: a row of ones -- its product is S3 = sum of the masked tile over the rows, for free
                    const bf16 ob = (bf16)1.f;
                    v = (bf16x8){ob, ob, ob, ob, ob, ob, ob, ob};
                }
                xT[rt][ks] = v;
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const bf16x8 wf = *(const bf16x8*)(smem_raw + DXL_W + ((fh * 4 + ft) * 64 + lane) * 16);
            f32x16 G = zero16;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int row = 64 * rq + 32 * rt + r;
                const float x0[4] = {raw_x1(row, 0), raw_x1(row, 1), raw_x1(row, 2), raw_x1(row, 3)};
                const f32x16 p1 = mfma(x_frag_state(x0, tl * 256 + row < p.Ns, h), wf, zero16);  // [row (registers)][feature (lane)]
                float md[16];
                const float* dsc = (const float*)(smem_raw + DXL_A) + xb * 256 + 64 * rq + 32 * rt + 4 * h;  // register 4 g + j <-> row 8 g + 4 h + j
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float dy = acc[rt][ft][i] * dsc[8 * (i >> 2) + (i & 3)];
                    md[i] = p1[i] > 0.f ? dy : 0.f;
                    s1 = fmaf(md[i], p1[i], s1);  // = dy relu(p)
                    s2 += dy;
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 mb;
#pragma unroll
                    for (int e = 0; e < 8; ++e) mb[e] = (bf16)md[8 * ks + e];
                    G = mfma(xT[rt][ks], mb, G);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // G[m][f]: register 4 g + j <-> m = 8 g + 4 h + j: registers 0..3 of the h = 0 half are the hi products of inputs 0..3,
            // of the h = 1 half the lo products (the halves are added in flush())
            // (register 4 of the h = 0 half: m = 8, the ones row = S3; of the h = 1 half: m = 12, zero)
            float q7[7] = {s1, s2, G[4], G[0], G[1], G[2], G[3]};
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                q7[q] += __shfl_xor(q7[q], 32);  // (the two halves hold different rows of the same feature; G: hi + lo products)
                if (h == 0) sums[(7 * ft + q) * 32] += q7[q];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (CRITIC) {
            // the wave's action tile: rows 64 rq + 32 rtx, features H1 + 32 at .. + 32, the partial dy over k-step fh of every chunk
            const int rowb = 64 * rq + 32 * rtx;
            auto raw_a = [&](int row) { return *(const float*)(smem_raw + DXL_ACT + (xb * 256 + row) * 4); };
            bf16x8 xTa[2];  // lane m = 0: hi of the action, m = 4: lo, m = 8: ones (the state layout with one input)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16 zb = (bf16)0.f;
                bf16x8 v = {zb, zb, zb, zb, zb, zb, zb, zb};
                if (r == 0 || r == 4) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xs = raw_a(rowb + 16 * ks + 8 * (e >> 2) + 4 * h + (e & 3));
                        const bf16 hi = (bf16)xs;
                        v[e] = r == 0 ? hi : (bf16)(xs - (float)hi);
                    }
                } else if (r == 8) {
                    const bf16 ob = (bf16)1.f;
                    v = (bf16x8){ob, ob, ob, ob, ob, ob, ob, ob};
                }
                xTa[ks] = v;
            }
            const f32x16 p1 = mfma(x_frag_action(raw_a(rowb + r), tl * 256 + rowb + r < p.Ns, h), wfa, zero16);
            const float* dsc = (const float*)(smem_raw + DXL_A) + xb * 256 + rowb + 4 * h;
            float md[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float dy = accx[i] * dsc[8 * (i >> 2) + (i & 3)];
                md[i] = p1[i] > 0.f ? dy : 0.f;
                s1 = fmaf(md[i], p1[i], s1);
                s2 += dy;
            }
            f32x16 G = zero16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 mb;
#pragma unroll
                for (int e = 0; e < 8; ++e) mb[e] = (bf16)md[8 * ks + e];
                G = mfma(xTa[ks], mb, G);
            }
            float q4[4] = {s1, s2, G[4], G[0]};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                q4[q] += __shfl_xor(q4[q], 32);
                if (h == 0) asum[q] += q4[q];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        xb = xbn;
        if (!same_set || tn == t) {
            flush();
            if (tn != t) {
                // a new set: its fragments, bases and stream start here (the groups are level: see the top of the tile loop).
                // load_set() overwrites the first-layer fragments in LDS that a slower wave of the same feature half may still be
                // reading in its epilogue: everybody is past it first
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_s_barrier();
                set = set_n;
                load_set();
                dma_x(tln, xb);
#pragma unroll
                for (int c = 0; c < DX_NS - 1; ++c) dma((stg + c) % DX_NS, tln, c);
                __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): the stream's first chunks and the fragments load_set() stored
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
}

// ------------------------------------------------------------------------------------------
// critic(s, mu) as a delta on critic(s, a): the two passes share the states and the weights, so
//   z2(mu) = z2(a) + W2[action features] . (f(mu) - f(a)),   f = the relu'd first layer of the action branch (64 padded features)
// -- 4 k-steps instead of the pass's 68. z2(a) comes back from the activations critic(s, a) stored with their sign (bf16), the
// 64 action columns of the weight block stay in LDS, and the epilogue is fwd_gen_kernel's for this pass: q partial sums and
// dZ2 = [z2 > 0] (-w/N) cf -- which never leaves the kernel: the only thing the pass wants from it is the action gradient
// (below). HBM-bound: reads one activation matrix.
struct DeltaP {
    const bf16* Zin;    // [sets][Np][H2]
    long setZ;
    const float *a, *mu;  // [sets][setA], [sets][setMu]
    long setA, setMu;
    const bf16x8* wf1;  // [sets][nft][64]
    int nft, nfs;
    const bf16* WT;     // [sets][H2n][ldw], k permuted
    long setWT, ldw;
    const float* cf;    // [sets][H2]
    float* z;           // [sets][setQ], pre-filled with c0: one f32 atomic per row and column block
    long setQ;
    float dz_scale;
    const float* rw;
    int Ns, Np, H2, H1, n_sets;
    // the action gradient in the same kernel: dC[n][k] = sum_c dZ2[n][c] W2[H1 + k][c] (one more product, B = the dZ2 tile as it stands),
    // da[n] = sum_k [p1(mu)[n][k] > 0] inv[k] Wa[k] dC[n][k] -- linear in dC, so every wave adds its 128 columns' part (f32 atomics)
    const bf16* Wn;     // [sets][..][H2]: bf16(W2), row = feature -- or bf16(cf[c] W2[f][c]) (cf_in_wn: the rank-one backward's operand)
    long setWn;
    int cf_in_wn;
    const float* inv;   // [sets][setTab] (index H1 + k)
    long setTab;
    const float* th;    // [sets][setTh] critic parameters (wa_off: Wa[k])
    long setTh;
    int wa_off, Ha;
    float* da;          // [sets][setDa], zeroed by the caller
    long setDa;
    bf16* Mk;           // [sets][Np][H2] or NULL: the relu mask {1, 0} of the z2(a) read here (critic(s, a)'s mask: the rank-one backward's
                        // operand) -- written by this pass, which reads every element anyway and is bound by neither memory nor VALU
};
constexpr int DL_ROWN = FC * 2 + 16;  // row stride of the 64 action rows of W2 in LDS (1040 B: conflict-free 16-byte reads down the rows)
constexpr int DL_W = 0, DL_CF = DL_W + FC * 64 * 2, DL_ZS = DL_CF + FC * 4, DL_WN = DL_ZS + 4 * FR * 4, DL_MK = DL_WN + 64 * DL_ROWN,
              DL_TOTAL = DL_MK + 64 * 4;

__global__ __launch_bounds__(FT) __attribute__((amdgpu_waves_per_eu(2, 2))) void fwd_delta_kernel(DeltaP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* sWa = (bf16*)(smem_raw + DL_W);   // [512 columns][64 k]: 128-byte rows, 16-byte pieces XOR-swizzled by bits 1..3 of the row
    float* scf = (float*)(smem_raw + DL_CF);
    float* zs = (float*)(smem_raw + DL_ZS);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int rh = wave & 1, cq = wave >> 1, ncb = p.H2 / FC, ntile = p.Np / FR;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mfma = [](bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); };
    for (int pair = 0; pair < p.n_sets * ncb; ++pair) {
        const int set = pair / ncb, cb = pair - set * ncb;
        if ((int)blockIdx.x >= ntile) continue;
        __syncthreads();
        {   // the weight block's 64 action columns (k = H1 .. H1 + 63) of image row (= output column) tid
            const bf16* src = p.WT + (long)set * p.setWT + (long)(FC * cb + tid) * p.ldw + p.H1;
#pragma unroll
            for (int c = 0; c < 8; ++c) *(wide::u32x4*)(sWa + tid * 64 + ((c ^ ((tid >> 1) & 7)) << 3)) = *(const wide::u32x4*)(src + 8 * c);
            scf[tid] = p.cf[(long)set * p.H2 + FC * cb + tid];
            // the 64 action rows of W2 (natural k = c order -> the accumulator's k order: the middle groups of four of every 16 swap)
            const int fr = tid >> 3, part = tid & 7;  // row, 64-column part
            const bf16* wsrc = p.Wn + (long)set * p.setWn + (long)(p.H1 + fr) * p.H2 + FC * cb + 64 * part;
            const unsigned keep = fr < p.Ha ? 0xffffffffu : 0u;  // (rows past the action layer's width: allocated, never written)
#pragma unroll 2
            for (int q16 = 0; q16 < 4; ++q16) {
                const wide::u32x4 lo = *(const wide::u32x4*)(wsrc + 16 * q16), hi = *(const wide::u32x4*)(wsrc + 16 * q16 + 8);
                unsigned char* d = smem_raw + DL_WN + fr * DL_ROWN + (64 * part + 16 * q16) * 2;
                *(wide::u32x4*)d = wide::u32x4{lo[0] & keep, lo[1] & keep, hi[0] & keep, hi[1] & keep};         // columns 0-3, 8-11
                *(wide::u32x4*)(d + 16) = wide::u32x4{lo[2] & keep, lo[3] & keep, hi[2] & keep, hi[3] & keep};  // columns 4-7, 12-15
            }
            if (tid < 64) ((float*)(smem_raw + DL_MK))[tid] = tid < p.Ha ? p.inv[(long)set * p.setTab + p.H1 + tid] * p.th[(long)set * p.setTh + p.wa_off + tid] : 0.f;
        }
        bf16x8 wf[2];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
            const bf16 zb = (bf16)0.f;
            wf[ta] = (bf16x8){zb, zb, zb, zb, zb, zb, zb, zb};
            if (p.nfs + ta < p.nft) wf[ta] = p.wf1[((long)set * p.nft + p.nfs + ta) * 64 + lane];
        }
        __syncthreads();
        // Half tiles (32 rows per wave) in a register pipeline: the next half tile's z2(a) rows (8 x 16 B per lane) and action pair
        // are requested before this one's products and epilogue. Loading, multiplying and storing one tile after the other
        // left the kernel waiting on memory: 1.4 TB/s with eight waves per CU.
        const int nu = (int)blockIdx.x < ntile ? 2 * ((ntile - 1 - (int)blockIdx.x) / (int)gridDim.x + 1) : 0;
        wide::u32x4 raw[8];
        float av = 0.f, mv = 0.f;
        auto request = [&](int u) {
            const int tile = blockIdx.x + (u >> 1) * gridDim.x, rt = u & 1;
            const long n = (long)tile * FR + rh * 64 + 32 * rt + r;
            // z2(a), this wave's 128 columns of the row: 16-byte pieces, the row's two lanes swap halves below (the store's inverse)
            const bf16* src = p.Zin + (long)set * p.setZ + n * p.H2 + FC * cb + 128 * cq + 8 * h;
#pragma unroll
            for (int i = 0; i < 8; ++i) raw[i] = *(const wide::u32x4*)(src + 16 * i);
            const long nc = n < p.Ns ? n : p.Ns - 1;  // (rows past the batch: any row's pair, the fragments are built dead)
            av = p.a[(long)set * p.setA + nc];
            mv = p.mu[(long)set * p.setMu + nc];
        };
        if (nu > 0) request(0);
        const float* mk = (const float*)(smem_raw + DL_MK);
        for (int u = 0; u < nu; ++u) {
            const int tile = blockIdx.x + (u >> 1) * gridDim.x, rt = u & 1;
            const long n = (long)tile * FR + rh * 64 + 32 * rt + r;
            const bool live = n < p.Ns;
            f32x16 acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const wide::u32x4 v = raw[2 * ct + gg];
                    const auto t0 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);
                    const auto t1 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
                    const unsigned pk[2][2] = {{t0[0], t1[0]}, {t0[1], t1[1]}};
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            acc[ct][4 * (2 * gg + g2) + 2 * e] = __uint_as_float(pk[g2][e] << 16);
                            acc[ct][4 * (2 * gg + g2) + 2 * e + 1] = __uint_as_float(pk[g2][e] & 0xffff0000u);
                        }
                }
            if (p.Mk) {  // critic(s, a)'s relu mask: same pieces, same places as the z2(a) image; before the next request overwrites them.
                // (Stores count in vmcnt like loads, so the wait for the next rows also waits for these: 1.20 -> 1.66 ms. Held in
                //  registers and stored behind the next request they spill -- 256 B of scratch, 2.48 ms.)
                bf16* md = p.Mk + (long)set * p.setZ + n * p.H2 + FC * cb + 128 * cq + 8 * h;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    wide::u32x4 v = raw[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = wide::sign_mask2(v[e]);
                    *(wide::u32x4*)(md + 16 * i) = v;
                }
            }
            const bf16x8 xa = x_frag_action(av, live, h), xm = x_frag_action(mv, live, h);
            bf16x8 dfr[2][2];  // [action tile][k-step]: bf16 of relu(p1(mu)) - relu(p1(a))
            unsigned long long pos = 0;  // relu mask of p1(mu), bit 16 ta + register
#pragma unroll
            for (int ta = 0; ta < 2; ++ta) {
                const f32x16 pa = mfma(wf[ta], xa, zero16), pm = mfma(wf[ta], xm, zero16);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 8; ++i) dfr[ta][ks][i] = (bf16)(fmaxf(pm[8 * ks + i], 0.f) - fmaxf(pa[8 * ks + i], 0.f));
#pragma unroll
                for (int i = 0; i < 16; ++i) pos |= (unsigned long long)(pm[i] > 0.f) << (16 * ta + i);
            }
            // (after the last use of what the allocator spills in this loop: a scratch reload is a vm operation, and waiting for
            //  it would wait for everything requested before it)
            __builtin_amdgcn_sched_barrier(0);
            if (u + 1 < nu) request(u + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const int row = 128 * cq + 32 * ct + r, c = 4 * ta + 2 * ks + h;
                        const bf16x8 aw = *(const bf16x8*)(sWa + row * 64 + ((c ^ ((row >> 1) & 7)) << 3));
                        acc[ct] = mfma(aw, dfr[ta][ks], acc[ct]);
                    }
                }
            // ---- epilogue (as fwd_gen_kernel's for this pass) + the action gradient
            const float drow = live ? p.dz_scale * (p.rw ? p.rw[(long)set * p.Ns + n] : 1.f) : 0.f;
            float zp = 0.f;
            f32x16 E[2] = {zero16, zero16};  // [action feature tile]: dC partial over this wave's 128 columns (lane = row, registers = features)
            asm volatile("" ::: "memory");  // (the coefficient tables are read from LDS every time: hoisted out of the loop they take 96 registers)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                __builtin_amdgcn_sched_barrier(0);
                unsigned pk[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 cv = *(const f32x4*)(scf + 128 * cq + 32 * ct + 8 * g + 4 * h);
                    bf16 o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = (bf16)fmaxf(acc[ct][4 * g + j], 0.f);
                        zp = fmaf((float)o[j], cv[j], zp);
                        o[j] = (bf16)((float)o[j] > 0.f ? (p.cf_in_wn ? drow : drow * cv[j]) : 0.f);
                    }
                    pk[g][0] = (unsigned)__builtin_bit_cast(unsigned short, o[0]) | ((unsigned)__builtin_bit_cast(unsigned short, o[1]) << 16);
                    pk[g][1] = (unsigned)__builtin_bit_cast(unsigned short, o[2]) | ((unsigned)__builtin_bit_cast(unsigned short, o[3]) << 16);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {  // the dZ2 tile as the B operand: k = its columns (registers 8 ks .. 8 ks + 7), n = row
                    wide::u32x4 bw;
                    bw[0] = pk[2 * ks][0], bw[1] = pk[2 * ks][1], bw[2] = pk[2 * ks + 1][0], bw[3] = pk[2 * ks + 1][1];
                    const bf16x8 bz = __builtin_bit_cast(bf16x8, bw);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const bf16x8 wn = *(const bf16x8*)(smem_raw + DL_WN + (32 * mt + r) * DL_ROWN + (128 * cq + 32 * ct + 16 * ks + 8 * h) * 2);
                        E[mt] = mfma(wn, bz, E[mt]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            float dap = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 mvv = *(const f32x4*)(mk + 32 * mt + 8 * g + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) dap += ((pos >> (16 * mt + 4 * g + j)) & 1) ? mvv[j] * E[mt][4 * g + j] : 0.f;
                }
            dap += __shfl_xor(dap, 32);
            if (h == 0 && live) atomicAdd(p.da + (long)set * p.setDa + n, dap);
            zp += __shfl_xor(zp, 32);
            if (h == 0) zs[cq * FR + rh * 64 + 32 * rt + r] = zp;
            if (rt == 1) {
                __syncthreads();
                if (tid < FR) atomicAdd(p.z + (long)set * p.setQ + (long)tile * FR + tid, (zs[tid] + zs[FR + tid]) + (zs[2 * FR + tid] + zs[3 * FR + tid]));
                __syncthreads();
            }
        }
    }
}
}  // namespace fw

// ---- workspace plan -------------------------------------------------------------------------
namespace {
struct Plan {
    Dims d;
    // byte offsets into the workspace
    size_t C, CT, P2, dZ2, dZ2T, dZ1;                // activations (all sets)
    size_t aC, aCT, aP2;                             // the online actor's activations, kept from pass 2 to pass 3
    size_t WT[4], Wn[2], bias[4];                    // weights: 0 actor, 1 critic, 2 target actor, 3 target critic
    size_t tabs[4];                                  // per net: inv/sh/rs/mean tables of [KCp + H2] floats x 4
    size_t cf[4], c0[4];                             // output-layer coefficient vectors
    size_t q, y, dq, a1, tt, da, zbuf, dcl;          // row vectors [sets][Np] (dcl: the current backward pass's seed, zero padding rows)
    size_t u, cs, acc;                               // [sets][H2] x 2, [sets][4]
    size_t bnacc;                                    // first-layer dgamma | dbeta partial tables [2][NSLICE][sets][ldT]
    size_t wf1[4];                                   // fused forward: first-layer fragments [sets][KCp / 32][64] x 16 B per net
    size_t total;
    long ldT;  // table stride per set: KCp + H2
};

static Plan make_plan(const avd_mlp_layout& L, int n_agents, int n_sets, int rows_per_agent = 0) {
    Plan p;
    Dims& d = p.d;
    d.S = L.S, d.H1 = L.H1, d.H2 = L.H2, d.Ha = L.Ha, d.KC = L.H1 + L.Ha, d.KCp = (int)rup(d.KC, 64);
    d.n_sets = n_sets, d.Ns = (n_agents / n_sets) * (rows_per_agent ? rows_per_agent : L.B), d.Np = (int)rup(d.Ns, 256);
    d.theta_size = L.theta_size, d.stats_size = L.stats_size;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += rup((long)bytes, 256);
        return at;
    };
    const size_t sets = n_sets, Np = d.Np;
    const size_t KCn = rup(d.KC, 256) + 256, H2n = rup(d.H2, 256), KCp = d.KCp;  // + a tile: dX tiles may start at column H1
    p.C = take(sets * Np * KCp * 2), p.CT = take(sets * KCn * Np * 2);
    p.P2 = take(sets * Np * d.H2 * 2), p.dZ2 = take(sets * Np * d.H2 * 2), p.dZ2T = take(sets * H2n * Np * 2);
    p.dZ1 = take(sets * Np * KCp * 2);
    p.aC = take(sets * Np * KCp * 2), p.aCT = take(sets * KCn * Np * 2), p.aP2 = take(sets * Np * d.H2 * 2);
    for (int i = 0; i < 4; ++i) p.WT[i] = take(sets * H2n * KCp * 2), p.bias[i] = take(sets * d.H2 * 4);
    for (int i = 0; i < 2; ++i) p.Wn[i] = take(sets * KCn * d.H2 * 2);
    p.ldT = KCp + d.H2;
    for (int i = 0; i < 4; ++i) p.tabs[i] = take(sets * p.ldT * 4 * 4), p.cf[i] = take(sets * d.H2 * 4), p.c0[i] = take(sets * 4);
    p.q = take(sets * Np * 4), p.y = take(sets * Np * 4), p.dq = take(sets * Np * 4), p.a1 = take(sets * Np * 4);
    p.tt = take(sets * Np * 4), p.da = take(sets * Np * 4), p.zbuf = take(sets * Np * 4), p.dcl = take(sets * Np * 4);
    p.u = take(sets * d.H2 * 4), p.cs = take(sets * d.H2 * 4), p.acc = take(sets * 4 * 4);
    p.bnacc = take(2 * (size_t)NSLICE * sets * p.ldT * 4);
    for (int i = 0; i < 4; ++i) p.wf1[i] = take(sets * (KCp / 32 + 1) * 64 * 16);
    p.total = o;
    return p;
}

static int check_wide(const avd_mlp_layout* L, int n_agents, int n_sets, const char* who) {
    AVD_REQUIRE(L, "%s: null layout", who);
    AVD_REQUIRE(n_sets > 0 && n_agents > 0 && n_agents % n_sets == 0, "%s: n_agents=%d must be a multiple of n_sets=%d", who,
                n_agents, n_sets);
    if (L->A != 1 || (L->S != 3 && L->S != 4) || L->H1 % 64 || L->H2 % 64 || L->Ha % 16 || L->B % 64) {
        set_error("%s: the shared-set learner implements A == 1, S in {3, 4}, layer1/layer2 sizes that are multiples of 64, an "
                  "action layer size multiple of 16 and batch sizes multiple of 64 (got S=%d A=%d H1=%d H2=%d Ha=%d B=%d)",
                  who, L->S, L->A, L->H1, L->H2, L->Ha, L->B);
        return AVD_E_UNSUPPORTED;
    }
    return AVD_OK;
}
}  // namespace

extern "C" int avd_learn_shared_workspace(const avd_mlp_layout* lay, int n_agents, int n_sets, size_t* bytes) {
    int rc = check_wide(lay, n_agents, n_sets, "avd_learn_shared_workspace");
    if (rc) return rc;
    AVD_REQUIRE(bytes, "avd_learn_shared_workspace: null pointer");
    *bytes = make_plan(*lay, n_agents, n_sets).total;
    return AVD_OK;
}

namespace {
// one network's prepared operands
struct NetOps {
    const float *th, *st;  // [sets][theta_size] (actor block at 0, critic at actor_size), [sets][stats_size]
    float *inv, *sh, *rs, *mean;  // tables [sets][ldT]: first-layer features at [0, KCp), second layer at [KCp, KCp + H2)
    bf16 *WT, *Wn;
    float *bias, *cf, *c0;
    const void* wf1;  // fused forward: first-layer fragments
};
}  // namespace

#define WIDE_CHECK(call)         \
    do {                         \
        int rc_ = (call);        \
        if (rc_) return rc_;     \
    } while (0)

extern "C" int avd_learn_shared_bf16(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta,
                                     const float* stats, const float* theta_t, const float* stats_t, const float* s,
                                     const float* a, const float* r, const float* s2, const float* row_weight, float gamma,
                                     float high, float* grads, float* losses, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    int rc = check_wide(lay, n_agents, n_sets, "avd_learn_shared_bf16");
    if (rc) return rc;
    AVD_REQUIRE(theta && stats && theta_t && stats_t && s && a && r && s2 && grads && workspace,
                "avd_learn_shared_bf16: null pointer");
    const avd_mlp_layout& L = *lay;
    const Plan pl = make_plan(L, n_agents, n_sets);
    AVD_REQUIRE(workspace_bytes >= pl.total, "avd_learn_shared_bf16: workspace %zu B < %zu B", workspace_bytes, pl.total);
    const Dims& d = pl.d;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace;
    auto B16 = [&](size_t off) { return (bf16*)(ws + off); };
    auto F32 = [&](size_t off) { return (float*)(ws + off); };
    const int sets = n_sets, Ns = d.Ns, Np = d.Np, H1 = d.H1, H2 = d.H2, Ha = d.Ha, KC = d.KC, KCp = d.KCp;
    const long KCn = rup(KC, 256) + 256, H2n = rup(H2, 256), ldT = pl.ldT;
    const long setC = (long)Np * KCp, setCT = KCn * Np, setP2 = (long)Np * H2, setZT = H2n * Np;
    const long setWT = H2n * KCp, setWn = KCn * H2;
    const int asz = L.actor_size;

    // zero what is accumulated into or read as padding
    (void)hipMemsetAsync(grads, 0, sizeof(float) * (size_t)sets * L.theta_size, st);
    (void)hipMemsetAsync(ws + pl.acc, 0, sizeof(float) * sets * 4, st);
    // (activation buffers need no clearing: their producers write every row < Np and every column < KCp, zeros in the
    //  padding; rows/columns beyond that only ever feed output elements the GEMM epilogues do not store)

    // second layers of 512 n columns (config 5: 1024): the forward passes run fused (fw::fwd_gen_kernel); AVD_WIDE_FUSED_FWD=0: layer-wise
    static const char* ff_env = AVD_DIAG_ENV("WIDE_FUSED_FWD");
    const bool fused_fwd = L.S == 4 && H2 % fw::FC == 0 && H1 % 32 == 0 && KCp % 32 == 0 && KCp / 32 >= fw::FSTG && !(ff_env && ff_env[0] == '0');
    static const char* fd_env = AVD_DIAG_ENV("WIDE_FUSED_DW");
    static const char* fx_env = AVD_DIAG_ENV("WIDE_FUSED_DX");
    static const char* fl_env = AVD_DIAG_ENV("WIDE_FUSED_DELTA");
    const bool fused_delta = fused_fwd && H1 % 32 == 0 && KCp - H1 == 64 && !(fl_env && fl_env[0] == '0');
    // The fused backward kernels exist in the rank-one form only (r06: dZ2 = d (x) cf (.) mask is never materialised; the forward
    // kernels store the relu mask, fw::dw_gen_kernel / fw::dx_gen_kernel / fw::fwd_delta_kernel take d and cf on their other operands):
    // all of them or none -- without one of them the backward pass runs layer-wise from out_bwd_kernel's dZ2.
    const bool r1 = fused_fwd && fused_delta && Np % (8 * fw::FK) == 0 && Np / (8 * fw::FK) >= 2 &&  // (>= 2 chunks per row range)
                    Np % 256 == 0 && H1 % 256 == 0 && 32 % (H1 / 256) == 0 && !(fd_env && fd_env[0] == '0') && !(fx_env && fx_env[0] == '0');
    const bool fused_dw = r1, fused_dx = r1;
    // critic(s, a) and critic(s, mu) in ONE forward pass (fw::fwd_gen_kernel EPI 4) instead of a pass that stores its signed activations + the
    // delta pass over them: whenever the rank-one chain runs (the action branch is two chunks: fused_delta)
    static const char* du_env = AVD_DIAG_ENV("WIDE_DUAL");
    const bool dual = r1 && KCp / 32 - H1 / 32 == 2 && !(du_env && du_env[0] == '0');
    static const char* fa_env = AVD_DIAG_ENV("WIDE_ACT_IN_DX");
    const bool act_in_dx = fused_dx && H1 / 256 == 4 && Ha <= 64 && !(fa_env && fa_env[0] == '0');  // (fw::dx_gen_kernel<true>)
    constexpr size_t fw_lds = fw::L_TOTAL, dw_lds = (size_t)fw::FSTG * fw::DW_STG;  // (forward; weight gradient: four stages, its epilogue's scratch inside them)
    if (fused_fwd) {
        // the > 64 KB dynamic-LDS opt-in, once per DEVICE of this process (the attribute belongs to the device's copy of the function)
        static unsigned long long fw_attr_done = 0;  // bit = device ordinal
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 64 || !((fw_attr_done >> dev) & 1ull)) {
            hipError_t e = hipSuccess;
            auto opt_in = [&](const void* fn, size_t bytes) {
                if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            };
            opt_in((const void*)fw::fwd_gen_kernel<false, 0>, fw_lds), opt_in((const void*)fw::fwd_gen_kernel<true, 0>, fw_lds);
            opt_in((const void*)fw::fwd_gen_kernel<false, 1>, fw_lds), opt_in((const void*)fw::fwd_gen_kernel<true, 1>, fw_lds);
            opt_in((const void*)fw::fwd_gen_kernel<false, 3>, fw_lds), opt_in((const void*)fw::fwd_gen_kernel<true, 2>, fw_lds);
            opt_in((const void*)fw::fwd_gen_kernel<true, 4>, fw::L_TOTAL_DUAL);
            opt_in((const void*)fw::dw_gen_kernel<false, 4>, dw_lds), opt_in((const void*)fw::dw_gen_kernel<true, 4>, dw_lds);
            opt_in((const void*)fw::dw_gen_kernel<false, 8>, dw_lds), opt_in((const void*)fw::dw_gen_kernel<true, 8>, dw_lds);
            opt_in((const void*)fw::dw_gen_kernel<false, 16>, dw_lds), opt_in((const void*)fw::dw_gen_kernel<true, 16>, dw_lds);
            opt_in((const void*)fw::dx_gen_kernel<false>, fw::DxL<false>::TOTAL), opt_in((const void*)fw::dx_gen_kernel<true>, fw::DxL<true>::TOTAL);
            opt_in((const void*)fw::fwd_delta_kernel, fw::DL_TOTAL);
            if (e != hipSuccess) {
                set_error("avd_learn_shared_bf16: hipFuncSetAttribute(dynamic LDS %zu B) on device %d: %s", fw_lds, dev, hipGetErrorString(e));
                return AVD_E_LAUNCH;
            }
            if (dev < 64) fw_attr_done |= 1ull << dev;
        }
    }
    // ---- per-net operand preparation: BN tables, folded/transposed bf16 weights, output-layer vectors
    NetOps net[4];  // 0 actor, 1 critic, 2 target actor, 3 target critic
    for (int i = 0; i < 4; ++i) {
        NetOps& n = net[i];
        const bool critic = (i & 1), target = (i >= 2);
        n.th = (target ? theta_t : theta) + (critic ? asz : 0);
        n.st = target ? stats_t : stats;
        float* tab = F32(pl.tabs[i]);
        n.inv = tab, n.sh = tab + (long)sets * ldT, n.rs = tab + 2L * sets * ldT, n.mean = tab + 3L * sets * ldT;
        n.WT = B16(pl.WT[i]), n.Wn = target ? nullptr : B16(pl.Wn[i]);
        n.bias = F32(pl.bias[i]), n.cf = F32(pl.cf[i]), n.c0 = F32(pl.c0[i]), n.wf1 = ws + pl.wf1[i];
        const int K = critic ? KC : H1;
        auto tables = [&](int g, int be, int mm, int mv, int len, int t_off, int pad_to) {
            hipLaunchKernelGGL(bn_tables_kernel, dim3((unsigned)rup(pad_to, 256) / 256, sets), dim3(256), 0, st, n.th, n.st,
                               (long)L.theta_size, (long)L.stats_size, g, be, mm, mv, len, n.inv, n.sh, n.rs, n.mean, ldT, t_off,
                               pad_to);
        };
        if (critic) {
            tables(L.cgs, L.cbes, L.cmms, L.cmvs, H1, 0, H1);
            tables(L.cga, L.cbea, L.cmma, L.cmva, Ha, H1, KCp - H1);
            tables(L.cg3, L.cbe3, L.cmm3, L.cmv3, H2, KCp, H2);
        } else {
            tables(L.ag1, L.abe1, L.amm1, L.amv1, H1, 0, KCp);
            tables(L.ag2, L.abe2, L.amm2, L.amv2, H2, KCp, H2);
        }
        const int w2 = critic ? L.cW2 : L.aW2, b2 = critic ? L.cb2 : L.ab2, w3 = critic ? L.cW3 : L.aW3, b3 = critic ? L.cb3 : L.ab3;
        hipLaunchKernelGGL(out_coefs_kernel, dim3(sets), dim3(256), 0, st, n.th, (long)L.theta_size, w3, b3, H2, n.inv + KCp,
                           n.sh + KCp, ldT, n.cf, n.c0, (long)H2);
        hipLaunchKernelGGL(prep_w2_kernel, dim3((unsigned)rup(H2, 32) / 32, (unsigned)rup(KCp, 32) / 32, sets), dim3(256), 0, st, n.th,
                           (long)L.theta_size, w2, K, H2, KCp, n.inv, ldT, n.WT, setWT, n.Wn, setWn, fused_fwd ? 1 : 0,
                           (r1 && n.Wn) ? n.cf : (const float*)nullptr);
        if (fused_fwd) {
            const int nfs = H1 / 32, nft = critic ? KCp / 32 : nfs;
            hipLaunchKernelGGL(fw::prep_wf1_kernel, dim3((unsigned)nft, sets), dim3(64), 0, st, n.th, (long)L.theta_size, L.S,
                               critic ? L.cWs : L.aW1, critic ? L.cbs : L.ab1, H1, critic ? L.cWa : 0, critic ? L.cba : 0,
                               critic ? Ha : 0, nfs, nft, (bf16x8*)(ws + pl.wf1[i]));
        }
        hipLaunchKernelGGL(bias2_kernel, dim3((unsigned)rup(H2, 64) / 64, sets), dim3(1024), 0, st, n.th, (long)L.theta_size, w2, b2, K,
                           H2, n.sh, ldT, n.bias, (long)H2);
    }
    WIDE_CHECK(check_launch("avd_learn_shared_bf16: operand preparation"));

    bf16 *C = B16(pl.C), *CT = B16(pl.CT), *P2 = B16(pl.P2), *dZ2 = B16(pl.dZ2), *dZ2T = B16(pl.dZ2T), *dZ1 = B16(pl.dZ1);
    // the lambdas below work on whichever activation buffers C / CT / P2 currently point to
    auto use_actor_buffers = [&](bool yes) {
        C = B16(yes ? pl.aC : pl.C), CT = B16(yes ? pl.aCT : pl.CT), P2 = B16(yes ? pl.aP2 : pl.P2);
    };
    float *q = F32(pl.q), *y = F32(pl.y), *dq = F32(pl.dq), *a1 = F32(pl.a1), *tt = F32(pl.tt), *da = F32(pl.da);
    float *u = F32(pl.u), *cs = F32(pl.cs), *acc = F32(pl.acc), *zbuf = F32(pl.zbuf);
    const long setX = (long)Ns * L.S;
    const dim3 g64((unsigned)1, (unsigned)rup(Np, 64) / 64, sets);

    // first layer: states (S = 3 or 4) or actions (1) -> columns [c0, c0 + H)
    auto l1 = [&](const NetOps& n, bool critic, bool action, const float* X, long set_x, bool transpose) {
        const int H = action ? Ha : H1, c0 = action ? H1 : 0;
        const int Hpad = action ? (KCp - H1) : H1;
        const int w = critic ? (action ? L.cWa : L.cWs) : L.aW1, b = critic ? (action ? L.cba : L.cbs) : L.ab1;
        dim3 grid((unsigned)rup(Hpad, 64) / 64, g64.y, sets);
        bf16* ct = transpose ? CT : nullptr;
        if (action)
            hipLaunchKernelGGL((l1_fwd_kernel<1>), grid, dim3(256), 0, st, X, set_x, n.th, (long)L.theta_size, w, b, H, Hpad, c0, Ns,
                               Np, C, (long)KCp, setC, ct, (long)Np, setCT);
        else if (L.S == 4)
            hipLaunchKernelGGL((l1_fwd_kernel<4>), grid, dim3(256), 0, st, X, set_x, n.th, (long)L.theta_size, w, b, H, Hpad, c0, Ns,
                               Np, C, (long)KCp, setC, ct, (long)Np, setCT);
        else
            hipLaunchKernelGGL((l1_fwd_kernel<3>), grid, dim3(256), 0, st, X, set_x, n.th, (long)L.theta_size, w, b, H, Hpad, c0, Ns,
                               Np, C, (long)KCp, setC, ct, (long)Np, setCT);
    };
    // second layer forward: P2 = relu(C @ WT^T + bias)
    // fused: first layer + second layer + output-layer dot from the raw inputs (X, act): C is not read
    // mu2 != NULL (critic): critic(s, act) AND critic(s, mu2) in one pass (fw::fwd_gen_kernel EPI 4): q(s, act) -> q, its relu mask -> the
    // dZ2 buffer; q(s, mu2) -> zbuf, the action gradient -> da
    auto l2f = [&](const NetOps& n, bool critic, const float* X, const float* act, long set_act, bool keep_p2, bool dz_out = false,
                   bool store_pre = false, const float* mu2 = nullptr) {
        fw::FwdP f;
        f.mu = mu2, f.setMu = Np, f.z2 = zbuf, f.da = da, f.setDa = Np, f.wa = n.th + L.cWa, f.setTh = L.theta_size, f.Ha = Ha;
        f.X = X, f.setX = setX, f.act = critic ? act : nullptr, f.setAct = set_act;
        f.wf1 = (const bf16x8*)n.wf1, f.nfs = H1 / 32, f.nft = critic ? KCp / 32 : H1 / 32;
        f.WT = n.WT, f.setWT = setWT, f.ldw = KCp, f.bias = n.bias, f.cf = n.cf, f.c0 = n.c0;
        f.P2 = keep_p2 ? (dz_out ? dZ2 : P2) : nullptr, f.setP2 = setP2, f.z = critic ? q : zbuf, f.setZ = Np, f.Ns = Ns, f.Np = Np, f.H2 = H2, f.n_sets = sets;
        f.dz_scale = dz_out ? -1.0f / (float)Ns : 0.f, f.rw = row_weight, f.store_pre = store_pre ? 1 : 0;
        f.mask_out = (r1 && !critic && keep_p2) ? 1 : 0;  // (rank-one backward: the actor's pass leaves its relu mask, nothing reads its activations)
        static const char* dbg_env = AVD_DIAG_ENV("FW_DBG");
        f.dbg = dbg_env ? atoi(dbg_env) : 0;
        f.stamp = nullptr;
#ifdef AVD_FW_STAMP
        static unsigned long long* d_fst = nullptr;
        if (!d_fst) (void)hipMalloc(&d_fst, (64 + 256 * 16) * 8);
        f.stamp = d_fst;
#endif
        if (H2 > fw::GC)
            hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)rup(Np, 256) / 256, sets), dim3(256), 0, st, f.z, (long)Np, n.c0, Np);
        const dim3 grid((unsigned)std::min<long>(avd::fset::cu_count(), Np / fw::GR));
        // the epilogue is a compile-time choice (fw::fwd_gen_kernel): nothing stored / signed bf16(z2) (critic) / relu mask (actor); the
        // run-time-flag form for what is left (relu'd activations, dZ2 out: the layer-wise backward's operands)
        static const char* epi_env = AVD_DIAG_ENV("WIDE_FWD_EPI0");  // diagnostics: =1 the r03 epilogue everywhere (A/B)
        const bool epi0 = epi_env && epi_env[0] == '1';
        if (mu2) {
            f.P2 = dZ2, f.dz_scale = -1.0f / (float)Ns, f.store_pre = 0, f.mask_out = 1;
            hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)rup(Np, 256) / 256, sets), dim3(256), 0, st, f.z2, (long)Np, n.c0, Np);
            (void)hipMemsetAsync(da, 0, sizeof(float) * sets * Np, st);
            hipLaunchKernelGGL((fw::fwd_gen_kernel<true, 4>), grid, dim3(fw::FT), (size_t)fw::L_TOTAL_DUAL, st, f);
        } else if (!f.P2 && !epi0) {
            if (critic)
                hipLaunchKernelGGL((fw::fwd_gen_kernel<true, 1>), grid, dim3(fw::FT), fw_lds, st, f);
            else
                hipLaunchKernelGGL((fw::fwd_gen_kernel<false, 1>), grid, dim3(fw::FT), fw_lds, st, f);
        } else if (critic && f.store_pre && f.dz_scale == 0.f && !epi0) {
            hipLaunchKernelGGL((fw::fwd_gen_kernel<true, 2>), grid, dim3(fw::FT), fw_lds, st, f);
        } else if (!critic && f.mask_out && f.dz_scale == 0.f && !epi0) {
            hipLaunchKernelGGL((fw::fwd_gen_kernel<false, 3>), grid, dim3(fw::FT), fw_lds, st, f);
        } else if (critic) {
            hipLaunchKernelGGL((fw::fwd_gen_kernel<true, 0>), grid, dim3(fw::FT), fw_lds, st, f);
        } else {
            hipLaunchKernelGGL((fw::fwd_gen_kernel<false, 0>), grid, dim3(fw::FT), fw_lds, st, f);
        }
#ifdef AVD_FW_STAMP
        {
            static int printed = 0;
            if (printed >= 4 && printed < 8) {  // (the second learn call's four forward passes)
                static unsigned long long hst[64 + 256 * 16];
                (void)hipStreamSynchronize(st);
                (void)hipMemcpy(hst, f.stamp, sizeof(hst), hipMemcpyDeviceToHost);
                for (int w_ = 0; w_ < 8; ++w_)
                    fprintf(stderr, "fwd_gen wave %d: prepare %llu barrier %llu multiply %llu barrier %llu | boundary: drain %llu epilogue %llu flush+init %llu  (P2 %d) cycles, pair 0\n", w_,
                            hst[w_ * 8], hst[w_ * 8 + 1], hst[w_ * 8 + 2], hst[w_ * 8 + 3], hst[w_ * 8 + 4], hst[w_ * 8 + 5], hst[w_ * 8 + 6], f.P2 != nullptr);
                // tile boundaries of pair 0 on the real-time counter (10 ns units, relative to the first)
                unsigned long long t0 = ~0ull;
                const int nb = (int)grid.x;
                for (int b = 0; b < nb; ++b) t0 = hst[64 + b * 16] < t0 ? hst[64 + b * 16] : t0;
                for (int k = 2; k < 5; ++k) {
                    fprintf(stderr, "fwd_gen tile %d epilogue start (end) per workgroup, x10 ns:", k);
                    for (int b = 0; b < nb; b += 5) fprintf(stderr, " %llu(%llu)", hst[64 + (b * 8 + k) * 2] - t0, hst[64 + (b * 8 + k) * 2 + 1] - hst[64 + (b * 8 + k) * 2]);
                    fprintf(stderr, "\n");
                }
            }
            ++printed;
        }
#endif
        return check_launch("avd_learn_shared_bf16: fused forward");
    };
    auto l2 = [&](const NetOps& n, bool critic) {
        GemmP p = {C, n.WT, KCp, KCp, setC, setWT, Ns, H2, critic ? KCp : (int)rup(H1, 64), 1};
        // the output layer rides on the GEMM epilogue: q (critic) or z (actor, in `da`-free scratch `zbuf`) = c0 + P2 . cf
        float* zdst = critic ? q : zbuf;
        hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)rup(Np, 256) / 256, sets), dim3(256), 0, st, zdst, (long)Np, n.c0, Np);
        EpiFwd e = {P2, H2, setP2, n.bias, H2, n.cf, zdst, (long)Np};
        return launch_gemm(p, e, sets, st, "avd_learn_shared_bf16: forward GEMM");
    };
    auto out_layer = [&](const NetOps&, int mode, float* out) {
        if (mode == 1)  // actor head; the critic's q is complete once the GEMM has run
            hipLaunchKernelGGL(tanh_rows_kernel, dim3((unsigned)rup(Ns, 256) / 256, sets), dim3(256), 0, st, zbuf, (long)Np, Ns, high,
                               out, tt);
    };
    auto rows = [&](int mode, const float* qv, const float* yt, const float* rd, float gh, float* out) {
        hipLaunchKernelGGL(rows_kernel, dim3((unsigned)(rup(Ns, ROWS_PER_BLOCK) / ROWS_PER_BLOCK), sets), dim3(256), 0, st, mode, Ns, (long)Np, qv, yt, rd, gh,
                           out, acc, row_weight);
    };
    // Note on row-vector strides: r arrives as [sets][Ns] (stride Ns), internal vectors use stride Np. The TD kernel
    // reads r with the internal stride, so r is first copied into `da` (free at that point) with the padded stride.
    // `transpose` = a backward pass follows: the first-layer activations (and their transposes) are materialised for it
    auto actor_forward = [&](const NetOps& n, const float* X, bool transpose) {
        // (fused: the forward generates the first layer, dw_gen generates it again, dx_gen regenerates it: no copy in memory)
        if (!fused_fwd || (transpose && !(fused_dw && fused_dx))) l1(n, false, false, X, setX, transpose && !fused_dw);
        if (fused_fwd)
            WIDE_CHECK(l2f(n, false, X, nullptr, 0, transpose));
        else
            WIDE_CHECK(l2(n, false));
        out_layer(n, 1, a1);
        return AVD_OK;
    };
    auto critic_forward = [&](const NetOps& n, const float* X, const float* act, long set_act, bool transpose) {
        if (!fused_fwd || (transpose && !(fused_dw && fused_dx))) {
            l1(n, true, false, X, setX, transpose && !fused_dw);
            l1(n, true, true, act, set_act, transpose && !fused_dw);
        }
        if (fused_fwd)  // (the critic(s, a) pass of a learn keeps the sign on its stored activations: critic(s, mu) continues from them)
            WIDE_CHECK(l2f(n, true, X, act, set_act, transpose, false, transpose && fused_delta));
        else
            WIDE_CHECK(l2(n, true));
        out_layer(n, 0, q);
        return AVD_OK;
    };
    // backward of layers 3 and 2 of `n` given the seed d[n]; weight gradients when `wg`
    auto backward = [&](const NetOps& n, bool critic, const float* dvec, bool wg, int acc_idx, float* gnet, const float* bX = nullptr,
                        const float* bAct = nullptr, long bSetAct = 0) {
        (void)hipMemsetAsync(u, 0, sizeof(float) * sets * H2, st);
        (void)hipMemsetAsync(cs, 0, sizeof(float) * sets * H2, st);
        const int K = critic ? KC : H1;
        const int w3 = critic ? L.cW3 : L.aW3, b3 = critic ? L.cb3 : L.ab3, gg = critic ? L.cg3 : L.ag2,
                  gbe = critic ? L.cbe3 : L.abe2, gb2 = critic ? L.cb2 : L.ab2;
        if (!fused_dw)
            hipLaunchKernelGGL(out_bwd_kernel, dim3((unsigned)rup(H2, 64) / 64, g64.y, sets), dim3(256), 0, st, P2, (long)H2, setP2, dvec,
                               (long)Np, n.cf, (long)H2, H2, Ns, Np, dZ2, wg ? dZ2T : nullptr, (long)Np, setZT, u, cs, (long)H2, 0);
        if (wg) {
            if (!fused_dw)
                hipLaunchKernelGGL(out_grads_kernel, dim3((unsigned)rup(H2, 256) / 256, sets), dim3(256), 0, st, n.th, (long)L.theta_size, w3,
                                   H2, n.inv + KCp, n.sh + KCp, n.rs + KCp, n.mean + KCp, ldT, u, cs, (long)H2, acc, acc_idx, gnet,
                                   (long)L.theta_size, w3, b3, gg, gbe, gb2, 1);
            if (fused_dw) {
                // rank-one form: the rows' records (|d| x, sign words, d), then G and S2 from the relu mask (critic: left in the dZ2 buffer
                // by fw::fwd_delta_kernel; actor: stored by its forward pass in place of the activations) -- fw::dw_gen_kernel: first layer
                // generated per 32-row chunk, mask streamed --, then u, db2 and the cf / sh terms of dW2
                unsigned char* aux = ws + pl.dZ2T;  // (the transposed gradient matrix of the layer-wise path: free here)
                const long setAux = (long)(Np / fw::FK + 2) * fw::AUX_REC;  // (+ one record of slack: the stream reads 512 B past the last)
                hipLaunchKernelGGL(fw::aux_pack_kernel, dim3((unsigned)(Np / fw::FK + 1), sets), dim3(64), 0, st, bX, setX,
                                   critic ? bAct : (const float*)nullptr, bSetAct, dvec, (long)Np, Ns, Np, aux, setAux, F32(pl.dcl));
                fw::DwP d2;
                d2.aux = aux, d2.setAux = setAux;
                d2.wf1 = (const bf16x8*)n.wf1, d2.nfs = H1 / 32, d2.nft = critic ? KCp / 32 : H1 / 32;
                d2.ZT = critic ? dZ2 : P2, d2.setZT = setP2, d2.ldz = H2, d2.inv = n.inv, d2.setTab = ldT, d2.s2 = cs;
                d2.dW = gnet + (critic ? L.cW2 : L.aW2), d2.setW = L.theta_size;
                d2.Ns = Ns, d2.Np = Np, d2.H2 = H2, d2.K = K, d2.n_sets = sets, d2.nsplit = 8;
                {   // row ranges per (set, column block): the split whose items fill whole rounds of an XCD's workgroups best
                    // (rounds x chunks per item; the smaller split on a tie: fewer epilogues)
                    const int nfb_ = (d2.nft * 32 + 127) / 128, wgs = std::max(1, avd::fset::cu_count() / 8);
                    long best = -1;
                    for (int ns = 8; ns <= 32; ns *= 2) {
                        if (Np % (ns * 32) != 0 || Np / (ns * 32) < 2) continue;
                        const long items = ((long)sets * (H2 / fw::FC) * ns + 7) / 8 * nfb_;
                        const long cost = (items + wgs - 1) / wgs * (Np / (ns * 32));
                        if (best < 0 || cost < best) best = cost, d2.nsplit = ns;
                    }
                }
                const int nfb = (d2.nft * 32 + 127) / 128, items = sets * (H2 / fw::FC) * d2.nsplit * nfb;
                (void)items;
                const dim3 grid((unsigned)(avd::fset::cu_count() / 8 * 8));  // (a multiple of the XCD count: see the kernel's item map)
                // (the rows of a chunk are shared by the first min(8, H1 / 128) feature blocks of a stream for the S2 sum)
                const int nshare = std::min(8, H1 / 128);
#define AVD_DW_LAUNCH(C, R) hipLaunchKernelGGL((fw::dw_gen_kernel<C, R>), grid, dim3(fw::FT), dw_lds, st, d2)
                if (nshare == 8) {
                    if (critic) AVD_DW_LAUNCH(true, 4); else AVD_DW_LAUNCH(false, 4);
                } else if (nshare == 4) {
                    if (critic) AVD_DW_LAUNCH(true, 8); else AVD_DW_LAUNCH(false, 8);
                } else {
                    if (critic) AVD_DW_LAUNCH(true, 16); else AVD_DW_LAUNCH(false, 16);
                }
#undef AVD_DW_LAUNCH
                hipLaunchKernelGGL(w2_post_kernel, dim3((unsigned)rup(H2, 64) / 64, sets), dim3(1024), 0, st, n.th, (long)L.theta_size,
                                   critic ? L.cW2 : L.aW2, K, H2, n.sh, ldT, n.bias, n.cf, gnet, (long)L.theta_size, u, cs, (long)H2);
                hipLaunchKernelGGL(out_grads_kernel, dim3((unsigned)rup(H2, 256) / 256, sets), dim3(256), 0, st, n.th, (long)L.theta_size, w3,
                                   H2, n.inv + KCp, n.sh + KCp, n.rs + KCp, n.mean + KCp, ldT, u, cs, (long)H2, acc, acc_idx, gnet,
                                   (long)L.theta_size, w3, b3, gg, gbe, gb2, 1);
                return check_launch("avd_learn_shared_bf16: fused weight gradient");
            }
            // dW2 = inv (.) (C^T dZ2) + sh (x) db2: reduction over the rows, split into chunks with f32 atomics
            int ksplit = 1;
            while (Np / ksplit > 16384 && Np % (ksplit * 2 * BK) == 0) ksplit *= 2;  // f32 atomics cost ~ one MFMA K-chunk of 4096
            GemmP p = {CT, dZ2T, Np, Np, setCT, setZT, K, H2, Np / ksplit, ksplit};
            EpiDw e = {gnet + (critic ? L.cW2 : L.aW2), H2, (long)L.theta_size, n.inv, n.sh, cs, ldT, H2, 1.0f};
            WIDE_CHECK(launch_gemm(p, e, sets, st, "avd_learn_shared_bf16: weight-gradient GEMM"));
        }
        return AVD_OK;
    };
    // dX GEMM with the BN/ReLU backward of the first layer(s) over columns [c_begin, c_end)
    auto dx = [&](const NetOps& n, int c_begin, int c_end, bool wg, bool critic) {
        // (rank-one form: the dZ2 buffer holds the relu mask, Wn carries cf -- the row factor d completes the product)
        GemmP p = {dZ2, n.Wn + (long)c_begin * H2, H2, H2, setP2, setWn, Ns, c_end - c_begin, H2, 1};
        EpiDx e = {C, dZ1, KCp, setC, n.inv, n.rs, n.mean, nullptr, nullptr, ldT, (long)sets * ldT, c_begin, r1 ? F32(pl.dcl) : (const float*)nullptr, (long)Np};
        (void)critic;
        if (wg) {  // dgamma / dbeta of the first layers accumulate in sliced table-shaped scratch, summed by flush_bn1
            (void)hipMemsetAsync(ws + pl.bnacc, 0, sizeof(float) * 2 * NSLICE * sets * ldT, st);
            e.dgamma = F32(pl.bnacc);
            e.dbeta = e.dgamma + (long)NSLICE * sets * ldT;
        }
        return launch_gemm(p, e, sets, st, "avd_learn_shared_bf16: input-gradient GEMM");
    };
    auto l1_grads = [&](const float* X, long set_x, int kin, int c0, int H, float* gnet, int w_off, int b_off) {
        const int rpb = 2048;
        dim3 grid((unsigned)rup(H, 64) / 64, (unsigned)rup(Ns, rpb) / rpb, sets);
        if (kin == 1)
            hipLaunchKernelGGL((l1_grads_kernel<1>), grid, dim3(256), 0, st, X, set_x, dZ1, (long)KCp, setC, c0, H, Ns, rpb, 1.0f, gnet,
                               (long)L.theta_size, w_off, b_off);
        else if (kin == 4)
            hipLaunchKernelGGL((l1_grads_kernel<4>), grid, dim3(256), 0, st, X, set_x, dZ1, (long)KCp, setC, c0, H, Ns, rpb, 1.0f, gnet,
                               (long)L.theta_size, w_off, b_off);
        else
            hipLaunchKernelGGL((l1_grads_kernel<3>), grid, dim3(256), 0, st, X, set_x, dZ1, (long)KCp, setC, c0, H, Ns, rpb, 1.0f, gnet,
                               (long)L.theta_size, w_off, b_off);
    };
    // fused input gradient + first-layer gradients (fw::dx_gen_kernel)
    auto dx_fused = [&](const NetOps& n, bool critic, float* gnet, const float* X, const float* act, long set_act) {
        fw::DxP d3;
        d3.X = X, d3.setX = setX, d3.act = critic ? act : nullptr, d3.setAct = set_act;
        d3.wf1 = (const bf16x8*)n.wf1, d3.nfs = H1 / 32, d3.nft = critic ? KCp / 32 : H1 / 32;
        d3.dZ = critic ? dZ2 : P2, d3.setDZ = setP2, d3.d = F32(pl.dcl), d3.setD = Np;  // (the pass's relu mask and seed)
        d3.Wn = n.Wn, d3.setWn = setWn, d3.inv = n.inv, d3.rs = n.rs, d3.mean = n.mean, d3.setTab = ldT;
        d3.g = gnet, d3.setG = L.theta_size;
        d3.w_off[0] = critic ? L.cWs : L.aW1, d3.b_off[0] = critic ? L.cbs : L.ab1, d3.g_off[0] = critic ? L.cgs : L.ag1, d3.be_off[0] = critic ? L.cbes : L.abe1;
        d3.w_off[1] = L.cWa, d3.b_off[1] = L.cba, d3.g_off[1] = L.cga, d3.be_off[1] = L.cbea;
        d3.Ns = Ns, d3.Np = Np, d3.H2 = H2, d3.H1 = H1, d3.Ha = Ha, d3.n_sets = sets;
        static const char* dxabl = AVD_DIAG_ENV("WIDE_DX_ABL");
        d3.abl = dxabl ? atoi(dxabl) : 0;
        // one workgroup per CU, dealt round-robin over the 8 XCDs: slots per XCD = CUs / 8 (32 on MI355X), each row group = nfb slots
        const int slots = std::max(1, avd::fset::cu_count() / 8);
        d3.nfb = H1 / 256, d3.groups_per_xcd = std::max(1, slots / d3.nfb);  // (the state features; the critic's 48 action features go the GEMM way)
        // the <false> form serves the state features of both nets; the <true> form also carries the critic's action features (one more
        // MFMA per wave and step, dealt over a row group's FOUR feature blocks: H1 = 1024) -- elsewhere they go the GEMM way
        const dim3 grid((unsigned)(8 * d3.groups_per_xcd * d3.nfb));
        if (critic && act_in_dx)
            hipLaunchKernelGGL((fw::dx_gen_kernel<true>), grid, dim3(fw::FT), (size_t)fw::DxL<true>::TOTAL, st, d3);
        else
            hipLaunchKernelGGL((fw::dx_gen_kernel<false>), grid, dim3(fw::FT), (size_t)fw::DxL<false>::TOTAL, st, d3);
        return check_launch("avd_learn_shared_bf16: fused input gradient");
    };
    auto flush_bn1 = [&](bool critic, float* gnet, bool state_part = true) {
        const float* dg = F32(pl.bnacc);
        const float* dbe = dg + (long)NSLICE * sets * ldT;
        auto go = [&](int t_off, int len, int gg, int gbe) {
            hipLaunchKernelGGL(bn1_flush_kernel, dim3((unsigned)rup(len, 256) / 256, sets), dim3(256), 0, st, dg, dbe, ldT,
                               (long)sets * ldT, t_off, len, gnet, (long)L.theta_size, gg, gbe);
        };
        if (critic) {
            if (state_part) go(0, H1, L.cgs, L.cbes);
            go(H1, Ha, L.cga, L.cbea);
        } else {
            go(0, H1, L.ag1, L.abe1);
        }
    };

    // ---- pass 0: targets  y = r + gamma * Q'(s2, mu'(s2))                                   (trainer.py:493-494)
    WIDE_CHECK(actor_forward(net[2], s2, false));
    WIDE_CHECK(critic_forward(net[3], s2, a1, (long)Np, false));
    (void)hipMemcpy2DAsync(da, sizeof(float) * Np, r, sizeof(float) * Ns, sizeof(float) * Ns, sets, hipMemcpyDeviceToDevice, st);
    rows(0, q, nullptr, da, gamma, y);

    // ---- pass 1: critic loss and gradient                                                   (trainer.py:495-498)
    float* gcrit = grads + asz;
    if (!dual) {
        WIDE_CHECK(critic_forward(net[1], s, a, (long)Ns, true));
        rows(1, q, y, nullptr, 0.f, dq);
    }
    auto pass1_backward = [&]() {
        WIDE_CHECK(backward(net[1], true, dq, true, 1, gcrit, s, a, (long)Ns));
        if (fused_dx) {
            WIDE_CHECK(dx_fused(net[1], true, gcrit, s, a, (long)Ns));
            if (!act_in_dx) {
                // the action branch (48 features): its activations, the small GEMM with the BN / ReLU epilogue, first-layer gradients
                l1(net[1], true, true, a, (long)Ns, false);
                WIDE_CHECK(dx(net[1], H1, KC, true, true));
                l1_grads(a, (long)Ns, 1, H1, Ha, gcrit, L.cWa, L.cba);
                flush_bn1(true, gcrit, false);
            }
        } else {
            WIDE_CHECK(dx(net[1], 0, KC, true, true));
            l1_grads(s, setX, L.S, 0, H1, gcrit, L.cWs, L.cbs);
            l1_grads(a, (long)Ns, 1, H1, Ha, gcrit, L.cWa, L.cba);
            flush_bn1(true, gcrit);
        }
        return AVD_OK;
    };
    // rank-one form: the forward half of pass 2 runs BEFORE the critic's backward pass (all gradients are taken at the same
    // pre-update weights, trainer.py:492-506: the order is free) -- the delta kernel, which reads critic(s, a)'s signed activations
    // anyway, leaves their relu mask for it. The critic's seed stays in `dq`; the seeds of passes 2 / 3 then go to `y` (free after rows(1)).
    float* dq3 = r1 ? y : dq;
    if (!r1) WIDE_CHECK(pass1_backward());

    // ---- pass 2: actor through the critic, gradient w.r.t. the action                       (trainer.py:502-506)
    use_actor_buffers(true);
    WIDE_CHECK(actor_forward(net[0], s, true));  // activations (rank-one form: their relu mask) stay for pass 3
    use_actor_buffers(false);
    // same states, same critic as pass 1: the state columns of C are still valid, only the action branch changes
    if (dual) {
        // both critic passes at once: q(s, a) -> q and its relu mask -> dZ2 (pass 1's backward operand); q(s, mu) -> zbuf, dq / d mu -> da
        WIDE_CHECK(l2f(net[1], true, s, a, (long)Ns, true, false, false, a1));
        rows(1, q, y, nullptr, 0.f, dq);  // (the critic's seed; reads y before rows(2) below reuses it)
    } else if (!fused_delta) l1(net[1], true, true, a1, (long)Np, false);  // (the input-gradient epilogue reads the action columns of C)
    if (dual) {
    } else if (fused_delta) {  // z2(mu) = z2(a) + W2[action] (f(mu) - f(a)): 4 k-steps on top of the stored critic(s, a) activations
        hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)rup(Np, 256) / 256, sets), dim3(256), 0, st, q, (long)Np, net[1].c0, Np);
        fw::DeltaP dl;
        dl.Zin = P2, dl.setZ = setP2, dl.a = a, dl.mu = a1, dl.setA = Ns, dl.setMu = Np, dl.wf1 = (const bf16x8*)net[1].wf1, dl.nft = KCp / 32,
        dl.nfs = H1 / 32, dl.WT = net[1].WT, dl.setWT = setWT, dl.ldw = KCp, dl.cf = net[1].cf, dl.z = q, dl.setQ = Np;
        dl.dz_scale = -1.0f / (float)Ns, dl.rw = row_weight, dl.Ns = Ns, dl.Np = Np, dl.H2 = H2, dl.H1 = H1, dl.n_sets = sets;
        dl.Wn = net[1].Wn, dl.setWn = setWn, dl.cf_in_wn = r1 ? 1 : 0, dl.inv = net[1].inv, dl.setTab = ldT, dl.th = net[1].th, dl.setTh = L.theta_size, dl.wa_off = L.cWa, dl.Ha = Ha;
        dl.da = da, dl.setDa = Np;
        dl.Mk = r1 ? dZ2 : nullptr;  // (critic(s, a)'s relu mask, for pass1_backward below)
        (void)hipMemsetAsync(da, 0, sizeof(float) * sets * Np, st);
        hipLaunchKernelGGL(fw::fwd_delta_kernel, dim3((unsigned)std::min<long>(2 * avd::fset::cu_count(), Np / fw::FR)), dim3(fw::FT),
                           (size_t)fw::DL_TOTAL, st, dl);
        WIDE_CHECK(check_launch("avd_learn_shared_bf16: critic(s, mu) as a delta"));
    } else if (fused_fwd)  // (its output-layer backward too: the seed of this pass is the constant -1/N, so dZ2 leaves the forward kernel)
        WIDE_CHECK(l2f(net[1], true, s, a1, (long)Np, true, true));
    else
        WIDE_CHECK(l2(net[1], true));
    out_layer(net[1], 0, q);
#ifdef AVD_DIAG
    if (const char* dump = getenv("AVD_WIDE_DUMP")) {  // diagnostics: q(s, mu) and the action gradient of this call, [sets][Np] f32 each
        std::vector<float> hb(3 * (size_t)sets * Np);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hb.data(), dual ? zbuf : q, sizeof(float) * sets * Np, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hb.data() + (size_t)sets * Np, da, sizeof(float) * sets * Np, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hb.data() + 2 * (size_t)sets * Np, a1, sizeof(float) * sets * Np, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(dump, "wb")) fwrite(hb.data(), sizeof(float), hb.size(), f), fclose(f);
    }
#endif
    rows(2, dual ? zbuf : q, nullptr, nullptr, 0.f, dq3);
    if (!fused_fwd) WIDE_CHECK(backward(net[1], true, dq3, false, 0, nullptr));
    if (!fused_delta) {
        WIDE_CHECK(dx(net[1], H1, KC, false, true));
        // da[n] = sum_k dZ1[n][H1 + k] * Wa[0][k]
        hipLaunchKernelGGL(row_dot_kernel, dim3((unsigned)rup(Ns, 4) / 4, sets), dim3(256), 0, st, dZ1, (long)KCp, setC, H1, Ha,
                           net[1].th + L.cWa, (long)L.theta_size, (const float*)nullptr, Ns, 0, 0.f, da, (float*)nullptr, (long)Np);
    }
    if (r1) WIDE_CHECK(pass1_backward());

    // ---- pass 3: actor gradient from the activations (rank-one form: the mask) kept in pass 2
    use_actor_buffers(true);
    rows(3, nullptr, tt, da, high, dq3);
    WIDE_CHECK(backward(net[0], false, dq3, true, 3, grads, s));
    if (fused_dx) {
        WIDE_CHECK(dx_fused(net[0], false, grads, s, nullptr, 0));
    } else {
        WIDE_CHECK(dx(net[0], 0, H1, true, false));
        l1_grads(s, setX, L.S, 0, H1, grads, L.aW1, L.ab1);
        flush_bn1(false, grads);
    }
    hipLaunchKernelGGL(losses_kernel, dim3(1), dim3(64), 0, st, acc, Ns, sets, losses);
    return check_launch("avd_learn_shared_bf16");
}

// ---- acting with shared weight sets: actor(state) for every agent of a set as one GEMM chain ---------------
extern "C" int avd_actor_forward_shared_workspace(const avd_mlp_layout* lay, int n_agents, int n_sets, size_t* bytes) {
    int rc = check_wide(lay, n_agents, n_sets, "avd_actor_forward_shared_workspace");
    if (rc) return rc;
    AVD_REQUIRE(bytes, "avd_actor_forward_shared_workspace: null pointer");
    *bytes = make_plan(*lay, n_agents, n_sets, 1).total;
    return AVD_OK;
}

extern "C" int avd_actor_forward_shared_bf16(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta,
                                             const float* stats, const float* state, float high, float* out, void* workspace,
                                             size_t workspace_bytes, void* stream) {
    int rc = check_wide(lay, n_agents, n_sets, "avd_actor_forward_shared_bf16");
    if (rc) return rc;
    AVD_REQUIRE(theta && stats && state && out && workspace, "avd_actor_forward_shared_bf16: null pointer");
    const avd_mlp_layout& L = *lay;
    const Plan pl = make_plan(L, n_agents, n_sets, 1);
    AVD_REQUIRE(workspace_bytes >= pl.total, "avd_actor_forward_shared_bf16: workspace %zu B < %zu B", workspace_bytes, pl.total);
    const Dims& d = pl.d;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace;
    const int sets = n_sets, Ns = d.Ns, Np = d.Np, H1 = d.H1, H2 = d.H2, KCp = d.KCp;
    const long H2n = rup(H2, 256), ldT = pl.ldT, setC = (long)Np * KCp, setP2 = (long)Np * H2, setWT = H2n * KCp;
    float* tab = (float*)(ws + pl.tabs[0]);
    float *inv = tab, *sh = tab + (long)sets * ldT, *rs = tab + 2L * sets * ldT, *mean = tab + 3L * sets * ldT;
    bf16 *C = (bf16*)(ws + pl.C), *P2 = (bf16*)(ws + pl.P2), *WT = (bf16*)(ws + pl.WT[0]);
    float *bias = (float*)(ws + pl.bias[0]), *cf = (float*)(ws + pl.cf[0]), *c0 = (float*)(ws + pl.c0[0]), *tt = (float*)(ws + pl.tt);
    auto tables = [&](int g, int be, int mm, int mv, int len, int t_off, int pad_to) {
        hipLaunchKernelGGL(bn_tables_kernel, dim3((unsigned)rup(pad_to, 256) / 256, sets), dim3(256), 0, st, theta, stats,
                           (long)L.theta_size, (long)L.stats_size, g, be, mm, mv, len, inv, sh, rs, mean, ldT, t_off, pad_to);
    };
    tables(L.ag1, L.abe1, L.amm1, L.amv1, H1, 0, KCp);
    tables(L.ag2, L.abe2, L.amm2, L.amv2, H2, KCp, H2);
    hipLaunchKernelGGL(prep_w2_kernel, dim3((unsigned)rup(H2, 32) / 32, (unsigned)rup(KCp, 32) / 32, sets), dim3(256), 0, st, theta,
                       (long)L.theta_size, L.aW2, H1, H2, KCp, inv, ldT, WT, setWT, (bf16*)nullptr, 0L, 0, (const float*)nullptr);
    hipLaunchKernelGGL(bias2_kernel, dim3((unsigned)rup(H2, 64) / 64, sets), dim3(1024), 0, st, theta, (long)L.theta_size, L.aW2, L.ab2,
                       H1, H2, sh, ldT, bias, (long)H2);
    hipLaunchKernelGGL(out_coefs_kernel, dim3(sets), dim3(256), 0, st, theta, (long)L.theta_size, L.aW3, L.ab3, H2, inv + KCp, sh + KCp,
                       ldT, cf, c0, (long)H2);
    dim3 grid((unsigned)rup(H1, 64) / 64, (unsigned)rup(Np, 64) / 64, sets);
    if (L.S == 4)
        hipLaunchKernelGGL((l1_fwd_kernel<4>), grid, dim3(256), 0, st, state, (long)Ns * 4, theta, (long)L.theta_size, L.aW1, L.ab1, H1, H1,
                           0, Ns, Np, C, (long)KCp, setC, (bf16*)nullptr, 0L, 0L);
    else
        hipLaunchKernelGGL((l1_fwd_kernel<3>), grid, dim3(256), 0, st, state, (long)Ns * 3, theta, (long)L.theta_size, L.aW1, L.ab1, H1, H1,
                           0, Ns, Np, C, (long)KCp, setC, (bf16*)nullptr, 0L, 0L);
    GemmP p = {C, WT, KCp, KCp, setC, setWT, Ns, H2, (int)rup(H1, 64), 1};
    float* zbuf = (float*)(ws + pl.zbuf);
    hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)rup(Np, 256) / 256, sets), dim3(256), 0, st, zbuf, (long)Np, c0, Np);
    EpiFwd e = {P2, H2, setP2, bias, H2, cf, zbuf, (long)Np};
    WIDE_CHECK(launch_gemm(p, e, sets, st, "avd_actor_forward_shared_bf16: forward GEMM"));
    // out is tightly packed [sets][Ns]: strided copy through tanh
    for (int k = 0; k < sets; ++k)
        hipLaunchKernelGGL(tanh_rows_kernel, dim3((unsigned)rup(Ns, 256) / 256, 1), dim3(256), 0, st, zbuf + (long)k * Np, 0L, Ns, high,
                           out + (long)k * Ns, tt + (long)k * Np);
    return check_launch("avd_actor_forward_shared_bf16");
}
