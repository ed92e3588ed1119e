// Learn kernel and update pipeline of the centralized framework (src/environment.py:35-52, 234-236, 281; config.py:26, 95;
// agent/model.py hidd_mult): one actor / critic per platoon with S = 4 L states, A = L actions and the reference widths x 1.2
// (307 / 153 / 57, held zero-padded as 320 / 160 / 64 -- params.py), exact f32 on the matrix cores (v_mfma_f32_16x16x4_f32). Same
// four-pass plan and LDS layout as gen::learn_kernel_g (mlp.hip), which stays the kernel of every other shape; what is different (r04):
//   * the dimensions are compile-time constants: register arrays have their real sizes (the general kernel holds 512 registers and
//     620 bytes of scratch per lane), loops have no tails;
//   * EIGHT waves per workgroup = two per SIMD (LDS still holds one model's activations, 158 KB: one workgroup per CU);
//   * tile plans for eight waves at N = 160: forward 32 rows x 48 / 32 columns per wave (weights three k-blocks ahead in a register
//     ring, stages pinned), weight gradient 32 x 32 items (50 / 60 of them), input gradient 16-column tiles (20 / 24);
//   * the output layer's backward pass on the matrix cores (d y = D W3^T and d W3 = y^T D are A-wide GEMMs: 24 MFMAs per 16 columns
//     instead of a 64 x 8 FMA loop per column on 160 threads), with the BN / relu backward, d gamma, d beta and the bias gradient of
//     the layer below in its epilogue (no separate column-sum pass);
//   * what two passes compute twice is computed once: pass 3 takes the actor's layer-2 activations of pass 2 back from a scratch in
//     the model's own gradient row, the critic's pass 2 resumes from pass 1's accumulators after the state blocks;
//   * the update (avd_learn_update_f32) is a pipeline of two streams: learn kernels over chunks of 256 models in the caller's stream,
//     each chunk's Adam + Polyak pass (optim.hip adam_polyak_rows_kernel) on a side stream under the next chunk's learn kernel.
// Every sum has a fixed order: results are a function of the inputs only. 4096 models at L = 5: 9.5 -> ~6 ms per step.
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/avddpg_hip.h"
#include "learn_common.h"

// optim.hip: Adam + Polyak of whole slab rows, theta_in -> theta_out, by a persistent grid of n_groups workgroups
int launch_adam_polyak_rows(const avd_mlp_layout* lay, int n_sets, int n_groups, const float* theta_in, float* theta_out, float* theta_t,
                            float* m, float* v, const float* grads, const int32_t* step, float actor_lr, float critic_lr, double tau,
                            void* stream);

namespace avd {
namespace fset {
int cu_count();  // fset.hip: CUs of the current device (cached per device ordinal)
}
namespace cen {

constexpr int NT = 512, NW = NT / 64;
constexpr int H1 = 320, HA = 64, H2 = 160, KC = H1 + HA;
constexpr int LDA = ld_of(KC), LDB = ld_of(H2);
static_assert(LDA == 388 && LDB == 164, "row strides");
static_assert(H2 == 2 * 80 && H2 % 32 == 0 && H1 % 32 == 0 && KC % 32 == 0, "tile plans");

// Optimisation fence (learn_common.h opaque_zero): lane-derived addresses are rebuilt where they are used -- LLVM otherwise hoists the
// per-lane addresses of the whole pass loop above it and spills them.
__device__ __forceinline__ int tidx() { return (int)threadIdx.x + opaque_zero(); }

template <int S, int A>
struct Lds {
    static constexpr int bufA = 0, bufB = bufA + TILE * LDA, invA = bufB + TILE * LDB, shA = invA + KC, invB = shA + KC, shB = invB + H2,
                         rsB = shB + H2, mmB = rsB + H2, db = mmB + H2, sX = db + H2, sR = sX + TILE * S, sAct = sR + TILE,
                         sY = sAct + TILE * A, sQ = sY + TILE * A, sD = sQ + TILE * A, sA1 = sD + TILE * A, sT = sA1 + TILE * A,
                         sDa = sT + TILE * A, red = sDa + TILE * A, total = red + NW;
    static_assert(sizeof(float) * total <= 160 * 1024, "one workgroup's LDS");
};

// First layer of a branch: out[r][col0 + k] = relu(sum_j X[r][j] W[j][k] + b[k]) and the BN coefficients of column k. 16 x 16 output
// tiles on the matrix cores: wave w takes column tiles w, w + 8, w + 16; the batch operands of all four row tiles are read once.
// In three steps, so that a pass exposes ONE memory latency for all its small tensors instead of one per tile and table: l1_load
// requests every weight of the wave's tiles, bn_table builds the coefficient tables (its loads queue behind), l1_mma computes.
constexpr int L1_TILES = 3;  // column tiles per wave: 320 / 16 / 8 waves, rounded up
template <int K>
struct L1W {
    float w[L1_TILES][(K + 3) / 4], b[L1_TILES];
};
template <int K>
__device__ __forceinline__ void l1_load(L1W<K>& q, const float* __restrict__ W, const float* __restrict__ b, int H) {
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    const int ctiles = H >> 4;
#pragma unroll
    for (int i = 0; i < L1_TILES; ++i) {
        const int col = 16 * min(wave + NW * i, ctiles - 1) + lr;
#pragma unroll
        for (int st = 0; st < (K + 3) / 4; ++st) q.w[i][st] = W[min(4 * st + lg, K - 1) * H + col];  // (an index past K meets a zero batch operand)
        q.b[i] = b[col];
    }
}
__device__ __forceinline__ void bn_table(const float* __restrict__ g, const float* __restrict__ be, const float* __restrict__ mm,
                                         const float* __restrict__ mv, int H, float* inv, float* sh) {
    for (int k = tidx(); k < H; k += NT) {
        const float iv = (1.0f / sqrtf(mv[k] + BN_EPS)) * g[k];
        inv[k] = iv;
        sh[k] = be[k] - mm[k] * iv;
    }
}
template <int K>
__device__ __forceinline__ void l1_mma(const L1W<K>& q, const float* X, int H, float* out) {
    constexpr int ST = (K + 3) / 4;
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    const int ctiles = H >> 4;
    float xa[4][ST];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            const int j = 4 * st + lg;
            xa[m][st] = (K % 4 == 0 || j < K) ? X[(16 * m + lr) * K + min(j, K - 1)] : 0.f;
        }
#pragma unroll
    for (int i = 0; i < L1_TILES; ++i) {
        const int t = wave + NW * i;
        if (t < ctiles) {
            float* o = out + 16 * t + lr;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < ST; ++st) acc = MFMA16(xa[m][st], q.w[i][st], acc);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) o[(16 * m + 4 * lg + reg) * LDA] = fmaxf(acc[reg] + q.b[i], 0.f);
            }
        }
    }
}

// BN coefficient tables of the layer in front of the output layer
__device__ __forceinline__ void coefs_b(const float* __restrict__ g, const float* __restrict__ be, const float* __restrict__ mm,
                                        const float* __restrict__ mv, float* invB, float* shB, float* rsB, float* mmB) {
    for (int k = tidx(); k < H2; k += NT) {
        const float rs = 1.0f / sqrtf(mv[k] + BN_EPS);
        const float iv = rs * g[k];
        invB[k] = iv, shB[k] = be[k] - mm[k] * iv, rsB[k] = rs, mmB[k] = mm[k];
    }
}

// hidden layer forward: out[r][n] = relu(sum_k bn(X[r][k]) W[k][n] + b[n]), X [64][LDA] and out [64][LDB] in LDS, W global [K][160].
// Wave w: rows 32 (w & 1) .. + 32 (two row tiles against every weight operand) and one of four column groups, 160 = 48 + 48 + 32 + 32:
// waves 0..3 (the older wave of each SIMD) take the 48-column groups, waves 4..7 the 32-column ones -- 24 : 16 MFMAs per k-block, the
// younger wave's 16 fit into the older one's load / operand stage. Tile t of a wave holds columns c0 + NTL lr + t: one 12- or 8-byte
// load per weight row, one such LDS write per output row. The weights of a 16-deep k-block (four rows per lane) are requested three
// blocks before the MFMAs that use them; the LDS operand of a block is READ one block ahead and turned into the MFMA operand
// (x inv + sh) at the head of its own MFMA stage.
// Fully unrolled, two pinned stages per k-block: [weight loads three blocks ahead + the next block's LDS reads] | [MFMAs]. Without the
// sched_barriers hipcc sinks every load down to its first use (one load in flight: the ring prefetches nothing), and as a loop over
// groups of RING blocks it drains the ring on the back-edge (mlp.hip fast::gemm_fwd).
// (r04, the first plan: 16 rows x 80 columns per wave, five tiles. 44 k cycles per actor call for 25.6 k of MFMA, ~1.6 k per k-block
// in the steady state; a ring of 6 or 8 blocks changed nothing, 96 + 64 columns for the older / younger wave neither; with the weight
// loads ablated the loop ran at 1.283 k per k-block, the matrix pipe's rate -- the cost was every weight being fetched by the four
// waves that shared its columns. 32 rows per wave halve the loads per MFMA: 33 k cycles per call, and 13 registers fewer.)
#ifndef CEN_RING
#define CEN_RING 4
#endif
// FIRST / SNAP split the reduction for an input whose leading SNAP k-blocks do not change between two calls (the critic's state
// features in passes 1 and 2: same states, same weights): a call with SNAP > 0 stores its accumulators as they stand before block
// SNAP (row-major [64][160] at `snap`, global), a call with FIRST > 0 starts from them and runs blocks FIRST.. only -- the same
// additions in the same order as a full run, bit for bit; one sixth of the GEMM and of the W2 read.
template <int K, int NTL, int FIRST, int SNAP>
__device__ __forceinline__ void gemm_fwd_rows32(const float* X, const float* inv, const float* sh, const float* __restrict__ W,
                                                const float* __restrict__ b, float* out, float* __restrict__ snap) {
    static_assert(NTL == 3 || NTL == 2, "48- or 32-column groups");
    static_assert(FIRST < K / 16 && SNAP < K / 16 && !(FIRST && SNAP), "resume / snapshot points");
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    const int rh = wave & 1, c0 = (NTL == 3 ? 0 : 96) + 16 * NTL * ((wave >> 1) & 1);
    constexpr int NB = K / 16, RING = CEN_RING;
    f32x4 acc[2][NTL];
    float* sp = snap + (32 * rh + 4 * lg) * H2 + c0 + NTL * lr;  // the lane's accumulator elements: rows + 16 m + reg, columns + t
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (FIRST > 0) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) acc[m][t][reg] = sp[(16 * m + reg) * H2 + t];
            }
        }
    const float* wp = W + (long)(4 * lg) * H2 + c0 + NTL * lr;  // row 16 blk + 4 lg + jj
    float rq[RING][4][NTL];
#define CEN_FWD_ISSUE(blk, d)                                                                   \
    {                                                                                           \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                      \
            const float* p_ = wp + (long)(16 * (blk) + jj) * H2;                                \
            *(f32x2*)rq[d][jj] = *(const f32x2*)p_;                                             \
            if constexpr (NTL == 3) rq[d][jj][2] = p_[2];                                       \
        }                                                                                       \
    }
#pragma unroll
    for (int d = 0; d < RING - 1; ++d)
        if (FIRST + d < NB) CEN_FWD_ISSUE(FIRST + d, (FIRST + d) % RING);
    const float* xr = X + (32 * rh + lr) * LDA + 4 * lg;
    f32x4 rx[2][2], ri[2], rh_[2];
#define CEN_FWD_READ(blk)                                                                                                      \
    rx[(blk) & 1][0] = *(const f32x4*)(xr + 16 * (blk)), rx[(blk) & 1][1] = *(const f32x4*)(xr + 16 * LDA + 16 * (blk)),      \
    ri[(blk) & 1] = *(const f32x4*)(inv + 16 * (blk) + 4 * lg), rh_[(blk) & 1] = *(const f32x4*)(sh + 16 * (blk) + 4 * lg)
    CEN_FWD_READ(FIRST);
#pragma unroll
    for (int blk = FIRST; blk < NB; ++blk) {
        if constexpr (SNAP > 0) {
            if (blk == SNAP && snap) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int t = 0; t < NTL; ++t)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) sp[(16 * m + reg) * H2 + t] = acc[m][t][reg];
            }
        }
        if (blk + RING - 1 < NB) CEN_FWD_ISSUE(blk + RING - 1, (blk + RING - 1) % RING);
        if (blk + 1 < NB) CEN_FWD_READ(blk + 1);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 a0 = rx[blk & 1][0] * ri[blk & 1] + rh_[blk & 1], a1 = rx[blk & 1][1] * ri[blk & 1] + rh_[blk & 1];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                acc[0][t] = MFMA16(a0[jj], rq[blk % RING][jj][t], acc[0][t]);
                acc[1][t] = MFMA16(a1[jj], rq[blk % RING][jj][t], acc[1][t]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef CEN_FWD_READ
#undef CEN_FWD_ISSUE
    float bc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) bc[t] = b[c0 + NTL * lr + t];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float* o = out + (32 * rh + 16 * m + 4 * lg + reg) * LDB + c0 + NTL * lr;
#pragma unroll
            for (int t = 0; t < NTL; ++t) o[t] = fmaxf(acc[m][t][reg] + bc[t], 0.f);
        }
}
template <int K, int FIRST = 0, int SNAP = 0>
__device__ __forceinline__ void gemm_fwd(const float* X, const float* inv, const float* sh, const float* __restrict__ W,
                                         const float* __restrict__ b, float* out, float* __restrict__ snap = nullptr) {
    if ((tidx() >> 6) < 4) gemm_fwd_rows32<K, 3, FIRST, SNAP>(X, inv, sh, W, b, out, snap);
    else gemm_fwd_rows32<K, 2, FIRST, SNAP>(X, inv, sh, W, b, out, snap);
}

// narrow GEMM: out[r][a] = sum_k x(r, k) W[k wk + a wa] (+ bias[a]), a < A <= 16; x = X[r ldx + k] inv[k] + sh[k] (BN = false: x = X).
// Wave w owns row tile w & 3 (the A columns sit in one 16-wide MFMA tile whose unused columns are fed zeros); all weights of the lane
// are requested up front. SPLIT: the reduction is cut in two halves, waves 0..3 write theirs (+ bias) to out, waves 4..7 theirs to
// out2 -- the consumer adds the two; otherwise waves 4..7 have nothing to do.
template <int K, int A, bool BN, bool SPLIT>
__device__ __forceinline__ void narrow_gemm(const float* X, int ldx, const float* inv, const float* sh, const float* __restrict__ W, int wk,
                                            int wa, const float* __restrict__ bias, float* out, float* out2) {
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    const int rt = wave & 3, kh = wave >> 2;
    if (!SPLIT && kh) return;
    constexpr int NBT = K / 16, NB = SPLIT ? NBT / 2 : NBT;
    static_assert(!SPLIT || NBT % 2 == 0, "two equal halves");
    const int k0 = SPLIT ? kh * (K / 2) : 0;
    const float colmask = (lr < A) ? 1.f : 0.f;
    const int ac = min(lr, A - 1);
    float wv[NB][4];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) wv[blk][jj] = W[(k0 + 16 * blk + 4 * lg + jj) * wk + ac * wa] * colmask;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* xr = X + (rt * 16 + lr) * ldx + k0 + 4 * lg;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        f32x4 x = *(const f32x4*)(xr + 16 * blk);
        if (BN) x = x * *(const f32x4*)(inv + k0 + 16 * blk + 4 * lg) + *(const f32x4*)(sh + k0 + 16 * blk + 4 * lg);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc = MFMA16(x[jj], wv[blk][jj], acc);
    }
    if (lr < A) {
        const float bb = (bias && !kh) ? bias[lr] : 0.f;
        float* o = kh ? out2 : out;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) o[(rt * 16 + 4 * lg + reg) * A + lr] = acc[reg] + bb;
    }
}

// Output layer backward through the BN below it, in place: bufB[r][k] (p = relu(z2)) -> dz[r][k], on the matrix cores.
//   dy[r][k] = sum_a D[r][a] W3[k][a];  dW3[k][a] = sum_r y[r][k] D[r][a], y = bn(p);  db3[a] = sum_r D[r][a]
//   dgamma[k] = sum_r dy (p - mean) rs;  dbeta[k] = sum_r dy;  dz = [p > 0] dy inv;  db[k] = sum_r dz (the bias gradient of the
//   layer that produced z2: what gen::learn_kernel_g's separate column-sum pass computes)
// Wave w owns the 16-column tiles w, w + 8 for all 64 rows, so every column sum ends inside the wave. A lane holds its column's 16
// activations in the accumulator layout (rows 16 m + 4 lg + reg); the reduction index of dW3 = y^T D is permuted so that exactly
// these values are its A operand.
template <int A>
__device__ __forceinline__ void out_bwd(float* bufB, const float* invB, const float* shB, const float* rsB, const float* mmB, const float* D,
                                        const float* __restrict__ W3, float* __restrict__ gW3, float* __restrict__ gb3,
                                        float* __restrict__ gg, float* __restrict__ gbe, float* db, float* __restrict__ gdb) {
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    constexpr int AS = (A + 3) / 4;  // k-steps of the A-deep products
    for (int t = wave; t < H2 / 16; t += NW) {
        const int c = 16 * t + lr;
        float w3[AS];
#pragma unroll
        for (int s = 0; s < AS; ++s) w3[s] = (4 * s + lg < A) ? W3[c * A + min(4 * s + lg, A - 1)] : 0.f;
        const float iv = invB[c], sf = shB[c], rs = rsB[c], mean = mmB[c];
        float p[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) p[m][reg] = bufB[(16 * m + 4 * lg + reg) * LDB + c];
        f32x4 dy[4], dw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            dy[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < AS; ++s) {
                const float d = (4 * s + lg < A) ? D[(16 * m + lr) * A + min(4 * s + lg, A - 1)] : 0.f;
                dy[m] = MFMA16(d, w3[s], dy[m]);
            }
            if (gW3) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float d = (lr < A) ? D[(16 * m + 4 * lg + reg) * A + min(lr, A - 1)] : 0.f;
                    dw = MFMA16(fmaf(p[m][reg], iv, sf), d, dw);
                }
            }
        }
        float dgm = 0.f, dbt = 0.f, dbs = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float v = dy[m][reg], pp = p[m][reg];
                dgm = fmaf(v * (pp - mean), rs, dgm);
                dbt += v;
                const float dz = (pp > 0.f) ? v * iv : 0.f;
                dbs += dz;
                bufB[(16 * m + 4 * lg + reg) * LDB + c] = dz;
            }
        dgm += __shfl_xor(dgm, 16), dbt += __shfl_xor(dbt, 16), dbs += __shfl_xor(dbs, 16);
        dgm += __shfl_xor(dgm, 32), dbt += __shfl_xor(dbt, 32), dbs += __shfl_xor(dbs, 32);
        if (lg == 0) {
            db[c] = dbs;
            if (gW3) gg[c] = dgm, gbe[c] = dbt, gdb[c] = dbs;
        }
        if (gW3 && lr < A) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) gW3[(16 * t + 4 * lg + reg) * A + lr] = dw[reg];
        }
    }
    if (gb3 && tidx() < A) {
        float sum = 0.f;
        for (int r = 0; r < TILE; ++r) sum += D[r * A + tidx()];
        gb3[tidx()] = sum;
    }
}

// weight gradient of a hidden layer fed by a BN output (operands in LDS):
//   dW[k][n] = inv[k] sum_r P[r][k] DZ[r][n] + sh[k] db[n], k < K, n < 160 -> gW[k 160 + n]
// Items of 32 k-rows x 32 columns dealt round-robin over the eight waves; tile (ta, tb) of an item holds rows k0 + 2 i + ta
// (i = 4 lg + reg) and columns n0 + 2 lr + tb: both operands of a 4-row step are one 8-byte LDS read, an element pair of the result
// one 8-byte store.
__device__ __forceinline__ void gemm_dw(const float* P, const float* inv, const float* sh, int K, const float* DZ, const float* db,
                                        float* __restrict__ gW) {
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    constexpr int groups = H2 / 32;
    const int items = groups * (K >> 5);
    for (int item = wave; item < items; item += NW) {
        const int n0 = (item % groups) * 32, k0 = (item / groups) * 32;
        const f32x2 dbc = *(const f32x2*)(db + n0 + 2 * lr);
        f32x4 acc[2][2];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) acc[ta][0] = acc[ta][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* pp = P + lg * LDA + k0 + 2 * lr;
        const float* dp = DZ + lg * LDB + n0 + 2 * lr;
#pragma unroll 8
        for (int r = 0; r < TILE; r += 4) {
            const f32x2 pa = *(const f32x2*)(pp + r * LDA);
            const f32x2 dz = *(const f32x2*)(dp + r * LDB);
#pragma unroll
            for (int ta = 0; ta < 2; ++ta) {
                acc[ta][0] = MFMA16(pa[ta], dz[0], acc[ta][0]);
                acc[ta][1] = MFMA16(pa[ta], dz[1], acc[ta][1]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kb = k0 + 2 * (4 * lg + j);
            const f32x2 iv = *(const f32x2*)(inv + kb), sf = *(const f32x2*)(sh + kb);
#pragma unroll
            for (int ta = 0; ta < 2; ++ta) {
                f32x2 o;
                o[0] = fmaf(iv[ta], acc[ta][0][j], sf[ta] * dbc[0]);
                o[1] = fmaf(iv[ta], acc[ta][1][j], sf[ta] * dbc[1]);
                *(f32x2*)(gW + (long)(kb + ta) * H2 + n0 + 2 * lr) = o;
            }
        }
    }
}

// input gradient of a hidden layer + BN / relu backward of the layer below, in place:
//   dy[r][c] = sum_n DZ[r][n] W[c][n], c in [c_begin, c_end);  dgamma[c] = sum_r dy (p - mm) rs;  dbeta[c] = sum_r dy;
//   P[r][c] <- dy rs g [p > 0].  g / mm / mv / dg / dbe are indexed by c - c_begin; dg == nullptr skips the parameter gradients.
// 16-column tiles round-robin over the eight waves; a tile's W slice (16 x 160: ten 16-byte loads per lane) is requested one tile ahead.
__device__ __forceinline__ void gemm_dx_bn(const float* DZ, const float* __restrict__ W, int c_begin, int c_end, float* P,
                                           const float* __restrict__ g, const float* __restrict__ mm, const float* __restrict__ mv,
                                           float* __restrict__ dg, float* __restrict__ dbe) {
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    constexpr int NB = H2 / 16;
    f32x4 wc[NB], wn[NB];
    float bnc[3] = {0.f, 0.f, 1.f}, bnn[3] = {0.f, 0.f, 1.f};  // gamma, mean, var of column c0 + lr
    int c0 = c_begin + wave * 16;
    if (c0 < c_end) {
        const float* wrow = W + (long)(c0 + lr) * H2 + 4 * lg;
#pragma unroll
        for (int q = 0; q < NB; ++q) wc[q] = *(const f32x4*)(wrow + 16 * q);
        bnc[0] = g[c0 + lr - c_begin], bnc[1] = mm[c0 + lr - c_begin], bnc[2] = mv[c0 + lr - c_begin];
    }
    for (; c0 < c_end; c0 += NW * 16) {
        const int cn = c0 + NW * 16;
        if (cn < c_end) {
            const float* wrow = W + (long)(cn + lr) * H2 + 4 * lg;
#pragma unroll
            for (int q = 0; q < NB; ++q) wn[q] = *(const f32x4*)(wrow + 16 * q);
            bnn[0] = g[cn + lr - c_begin], bnn[1] = mm[cn + lr - c_begin], bnn[2] = mv[cn + lr - c_begin];
        }
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            f32x4 a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = *(const f32x4*)(DZ + (m * 16 + lr) * LDB + 16 * q + 4 * lg);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = MFMA16(a[m][jj], wc[q][jj], acc[m]);
        }
        const int c = c0 + lr;
        const float rs = 1.0f / sqrtf(bnc[2] + BN_EPS), gam = bnc[0], mean = bnc[1];
        float sg = 0.f, sb = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = m * 16 + lg * 4 + j;
                const float dy = acc[m][j], p = P[r * LDA + c];
                sg = fmaf(dy * (p - mean), rs, sg);
                sb += dy;
                P[r * LDA + c] = (p > 0.f) ? dy * (rs * gam) : 0.f;
            }
        sg += __shfl_xor(sg, 16), sb += __shfl_xor(sb, 16);
        sg += __shfl_xor(sg, 32), sb += __shfl_xor(sb, 32);
        if (dg && lg == 0) dg[c - c_begin] = sg, dbe[c - c_begin] = sb;
#pragma unroll
        for (int q = 0; q < NB; ++q) wc[q] = wn[q];
        bnc[0] = bnn[0], bnc[1] = bnn[1], bnc[2] = bnn[2];
    }
}

// first-layer gradients from dz[r][c0 .. c0 + H): dW[j][k] = sum_r X[r K + j] dz[r][k], db[k] = sum_r dz[r][k]. dW = X^T dz on the
// matrix cores: (16 inputs) x (16 columns) tiles over the waves, the 64 batch rows in 16 MFMA steps; db: one column per thread.
template <int K>
__device__ __forceinline__ void l1_grads(const float* X, const float* DZ, int c0, int H, float* __restrict__ gW, float* __restrict__ gb) {
    for (int k = tidx(); k < H; k += NT) {
        const float* dzk = DZ + c0 + k;
        float sb = 0.f;
#pragma unroll 16
        for (int r = 0; r < TILE; ++r) sb += dzk[r * LDA];
        gb[k] = sb;
    }
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    constexpr int jtiles = (K + 15) / 16;
    const int ctiles = H >> 4;
    for (int item = wave; item < jtiles * ctiles; item += NW) {
        const int jt = item % jtiles, t = item / jtiles;
        const int j = 16 * jt + lr;
        const float jm = (j < K) ? 1.f : 0.f;
        const float* xc = X + min(j, K - 1);
        const float* dc = DZ + c0 + 16 * t + lr;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < TILE / 4; ++st) {
            const int r = 4 * st + lg;
            acc = MFMA16(xc[r * K] * jm, dc[r * LDA], acc);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int jr = 16 * jt + 4 * lg + reg;
            if (jr < K) gW[jr * H + 16 * t + lr] = acc[reg];
        }
    }
}

__device__ __forceinline__ float block_sum(const float* v, int n, float* red) {  // sum of n LDS values, all threads get it
    float s = 0.f;
    for (int i = tidx(); i < n; i += NT) s += v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tidx() & 63) == 0) red[tidx() >> 6] = s;
    lds_barrier();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += red[w];
    lds_barrier();
    return t;
}

// one load per 128-byte line of [lo, hi): pulls a network's small tensors into L2 ahead of the passes that read them a few values at a time
__device__ __forceinline__ float warm(const float* __restrict__ base, int lo, int hi) {
    float t = 0.f;
    for (int i = lo + tidx() * 32; i < hi; i += NT * 32) t += base[i];
    return t;
}

// One agent (64-row batch) per workgroup; gradients go to the agent's row of the gradient slab (workers/trainer.py:472-508).
// STATS (the learn half of avd_learn_update_f32): the frozen BN statistics' soft update happens here too (agent/ddpgagent.py:44-53
// iterates .weights), once the target networks' forward passes have read the old ones.
template <int S, int A, bool STATS>
__global__ __launch_bounds__(NT) void learn_kernel_c(avd_mlp_layout L_arg, int set_mod, const float* __restrict__ theta,
                                                     const float* __restrict__ stats, float* __restrict__ theta_t,
                                                     float* __restrict__ stats_t, const float* __restrict__ s, const float* __restrict__ a,
                                                     const float* __restrict__ r, const float* __restrict__ s2, float gamma, float high,
                                                     float* __restrict__ grads, float* __restrict__ losses, float tau, float omt) {
    // the layout (43 offsets) is the first bytes of the kernarg segment: read through that pointer, re-derived inside each pass behind
    // an opaque zero -- as a by-value argument every field the kernel will ever need is loaded up front and kept alive (lean.hip)
    const avd_mlp_layout* const Lk = (const avd_mlp_layout*)__builtin_amdgcn_kernarg_segment_ptr();
    if (L_arg.theta_size != Lk->theta_size || L_arg.stats_size != Lk->stats_size) __builtin_trap();  // the layout IS argument 0
    extern __shared__ __attribute__((aligned(16))) float smem0[];
    typedef Lds<S, A> O;
    const int agent = blockIdx.x;
    {
        const avd_mlp_layout& L = *Lk;
        float* const smem = smem0;
        const int tid = tidx();
        float *sR = smem + O::sR, *sAct = smem + O::sAct;
        float* ga = grads + (long)agent * L.theta_size;
        float* gc = ga + L.actor_size;
        if (tid < TILE) sR[tid] = r[(long)agent * TILE + tid];
        for (int i = tid; i < TILE * A; i += NT) sAct[i] = a[(long)agent * TILE * A + i];
        if (tid == 0) {  // alignment padding of the gradient slab (only the A-wide biases can end off a 4-float boundary)
            for (int i = L.ab3 + A; i < L.actor_size; ++i) ga[i] = 0.f;
            for (int i = L.cb3 + A; i < L.theta_size - L.actor_size; ++i) gc[i] = 0.f;
        }
        const int set = set_mod > 0 ? agent % set_mod : agent;
        const Net net = {theta + (long)set * L.theta_size, stats + (long)set * L.stats_size};
        const Net tgt = {theta_t + (long)set * L.theta_size, stats_t + (long)set * L.stats_size};
        float t = 0.f;
        const int csz = L.theta_size - L.actor_size;
        t += warm(tgt.th, 0, L.aW2) + warm(tgt.th, L.ab2, L.actor_size);
        t += warm(tgt.th + L.actor_size, 0, L.cW2) + warm(tgt.th + L.actor_size, L.cb2, csz);
        t += warm(net.th, 0, L.aW2) + warm(net.th, L.ab2, L.actor_size);
        t += warm(net.th + L.actor_size, 0, L.cW2) + warm(net.th + L.actor_size, L.cb2, csz);
        t += warm(net.st, 0, L.stats_size) + warm(tgt.st, 0, L.stats_size);
        asm volatile("" ::"v"(t));  // keep the loads
    }
    PH_INIT();
    // pass 0: targets (y); 1: critic loss + gradient; 2: actor -> critic, gradient wrt the actions; 3: actor gradient
#pragma nounroll
    for (int it = 0; it < 4; ++it) {
        const avd_mlp_layout& L = *(const avd_mlp_layout*)((const char*)Lk + opaque_zero());
        float* const smem = smem0 + opaque_zero();
        float *bufA = smem + O::bufA, *bufB = smem + O::bufB, *invA = smem + O::invA, *shA = smem + O::shA, *invB = smem + O::invB,
              *shB = smem + O::shB, *rsB = smem + O::rsB, *mmB = smem + O::mmB, *db = smem + O::db, *sX = smem + O::sX, *sR = smem + O::sR,
              *sAct = smem + O::sAct, *sY = smem + O::sY, *sQ = smem + O::sQ, *sD = smem + O::sD, *sA1 = smem + O::sA1, *sT = smem + O::sT,
              *sDa = smem + O::sDa, *red = smem + O::red;
        const int tid = tidx();
        const int set = set_mod > 0 ? agent % set_mod : agent;
        const Net net = {theta + (long)set * L.theta_size, stats + (long)set * L.stats_size};
        const Net tgt = {theta_t + (long)set * L.theta_size, stats_t + (long)set * L.stats_size};
        float* ga = grads + (long)agent * L.theta_size;
        float* gc = ga + L.actor_size;
        constexpr float invn = 1.0f / (float)(TILE * A);
        const Net n = (it == 0) ? tgt : net;
        if (it < 2) {  // state batch of this pass: s2 for the targets, s afterwards
            lds_barrier();
            const float* src = (it == 0 ? s2 : s) + (long)agent * TILE * S;
            for (int i = tid; i < TILE * S; i += NT) sX[i] = src[i];
            lds_barrier();
        }
        PH(20);
        // Scratch in the agent's own gradient row: its ACTOR block is written by pass 3 only, so until then [0, 64 x 160) floats hold the
        // online actor's layer-2 activations of pass 2 (pass 3 needs them again: same weights, same states) and the next 64 x 160 the
        // critic's layer-2 accumulators after the state blocks (passes 1 and 2 run it on the same states with the same weights).
        float* const keep_p2 = ga;
        float* const keep_cs = ga + TILE * H2;
        if (it != 1) {  // ---- actor forward (agent/model.py:26-36)
            const float* th = n.th;
            L1W<S> qa;
            l1_load<S>(qa, th + L.aW1, th + L.ab1, H1);
            bn_table(th + L.ag1, th + L.abe1, n.st + L.amm1, n.st + L.amv1, H1, invA, shA);
            coefs_b(th + L.ag2, th + L.abe2, n.st + L.amm2, n.st + L.amv2, invB, shB, rsB, mmB);
            l1_mma<S>(qa, sX, H1, bufA);
            PH(21);
            if (it == 3) {
                // pass 2 has computed exactly this layer: its activations come back from the scratch (41 KB against 205 KB of W2 and
                // 3200 MFMAs), the tanh values are still in sT
                __syncthreads();  // (vmcnt(0) in every wave: pass 2's copy is complete, whichever wave wrote it)
                for (int i = tid; i < TILE * (H2 / 4); i += NT) {
                    const int r = i / (H2 / 4), c4 = i - r * (H2 / 4);
                    *(f32x4*)(bufB + r * LDB + 4 * c4) = *(const f32x4*)(keep_p2 + r * H2 + 4 * c4);
                }
                lds_barrier();
                PH(2);
            } else {
                lds_barrier();
                PH(1);
                gemm_fwd<H1>(bufA, invA, shA, th + L.aW2, th + L.ab2, bufB);
                lds_barrier();
                PH(2);
                narrow_gemm<H2, A, true, true>(bufB, LDB, invB, shB, th + L.aW3, A, 1, th + L.ab3, sQ, sD);  // (sD: free until this pass's loss)
                if (it == 2) {
                    for (int i = tid; i < TILE * (H2 / 4); i += NT) {
                        const int r = i / (H2 / 4), c4 = i - r * (H2 / 4);
                        *(f32x4*)(keep_p2 + r * H2 + 4 * c4) = *(const f32x4*)(bufB + r * LDB + 4 * c4);
                    }
                }
                lds_barrier();
                for (int i = tid; i < TILE * A; i += NT) {
                    const float t = tanhf(sQ[i] + sD[i]);
                    sT[i] = t, sA1[i] = t * high;
                }
                lds_barrier();
                PH(3);
            }
        }
        if (it != 3) {  // ---- critic forward (agent/model.py:63-83)
            const float* th = n.th + L.actor_size;
            const float* act = (it == 1) ? sAct : sA1;
            L1W<S> qs;
            L1W<A> qx;
            if (it != 2) l1_load<S>(qs, th + L.cWs, th + L.cbs, H1);
            l1_load<A>(qx, th + L.cWa, th + L.cba, HA);
            if (it != 2) bn_table(th + L.cgs, th + L.cbes, n.st + L.cmms, n.st + L.cmvs, H1, invA, shA);
            bn_table(th + L.cga, th + L.cbea, n.st + L.cmma, n.st + L.cmva, HA, invA + H1, shA + H1);
            coefs_b(th + L.cg3, th + L.cbe3, n.st + L.cmm3, n.st + L.cmv3, invB, shB, rsB, mmB);
            if (it != 2) l1_mma<S>(qs, sX, H1, bufA);  // (pass 2: the state blocks' sums come from pass 1's snapshot, nothing else reads them)
            PH(22);
            l1_mma<A>(qx, act, HA, bufA + H1);
            PH(23);
            lds_barrier();
            PH(4);
            if (it == 2) gemm_fwd<KC, H1 / 16, 0>(bufA, invA, shA, th + L.cW2, th + L.cb2, bufB, keep_cs);
            else gemm_fwd<KC, 0, H1 / 16>(bufA, invA, shA, th + L.cW2, th + L.cb2, bufB, it == 1 ? keep_cs : nullptr);
            lds_barrier();
            PH(5);
            narrow_gemm<H2, A, true, true>(bufB, LDB, invB, shB, th + L.cW3, A, 1, th + L.cb3, sQ, sD);
            lds_barrier();
            PH(6);
        }
        if (it == 0) {  // y = r + gamma Q'(s2, mu'(s2)), r broadcast over the A outputs, no done mask (trainer.py:494)
            for (int i = tid; i < TILE * A; i += NT) sY[i] = fmaf(gamma, sQ[i] + sD[i], sR[i / A]);
            if constexpr (STATS) {
#pragma clang fp contract(off)
                float* stt = stats_t + (long)set * L.stats_size;
                for (int i = tid; i < L.stats_size; i += NT) stt[i] = net.st[i] * tau + stt[i] * omt;
            }
            continue;
        }
        if (it == 1) {  // Lc = mean((y - q)^2) over B A (trainer.py:496)
            for (int i = tid; i < TILE * A; i += NT) {
                const float e = sY[i] - (sQ[i] + sD[i]);
                sD[i] = -2.0f * e * invn;
                sT[i] = e * e;
            }
            lds_barrier();
            const float lc = block_sum(sT, TILE * A, red) * invn;
            if (tid == 0 && losses) losses[(long)agent * 2 + 0] = lc;
        } else if (it == 2) {  // La = -mean(q1) (trainer.py:504)
            for (int i = tid; i < TILE * A; i += NT) sQ[i] += sD[i];  // (block_sum reads back what the same thread wrote)
            const float la = -block_sum(sQ, TILE * A, red) * invn;
            if (tid == 0 && losses) losses[(long)agent * 2 + 1] = la;
            for (int i = tid; i < TILE * A; i += NT) sD[i] = -invn;
        } else {  // through tanh(.) high
            for (int i = tid; i < TILE * A; i += NT) {
                const float t = sT[i];
                sD[i] = sDa[i] * high * (1.0f - t * t);
            }
        }
        lds_barrier();
        const bool crit = (it != 3), wg = (it != 2);
        const float* wth = crit ? net.th + L.actor_size : net.th;
        float* gout = crit ? gc : ga;
        out_bwd<A>(bufB, invB, shB, rsB, mmB, sD, wth + (crit ? L.cW3 : L.aW3), wg ? gout + (crit ? L.cW3 : L.aW3) : nullptr,
                   wg ? gout + (crit ? L.cb3 : L.ab3) : nullptr, gout + (crit ? L.cg3 : L.ag2), gout + (crit ? L.cbe3 : L.abe2), db,
                   gout + (crit ? L.cb2 : L.ab2));
        lds_barrier();
        PH(it == 1 ? 7 : (it == 2 ? 12 : 15));
        if (wg) {
            gemm_dw(bufA, invA, shA, crit ? KC : H1, bufB, db, gout + (crit ? L.cW2 : L.aW2));
            lds_barrier();
            PH(it == 1 ? 9 : 17);
        }
        const float* w2 = wth + (crit ? L.cW2 : L.aW2);
        if (it != 2) {
            if (crit)
                gemm_dx_bn(bufB, w2, 0, H1, bufA, wth + L.cgs, net.st + L.cmms, net.st + L.cmvs, gc + L.cgs, gc + L.cbes);
            else
                gemm_dx_bn(bufB, w2, 0, H1, bufA, wth + L.ag1, net.st + L.amm1, net.st + L.amv1, ga + L.ag1, ga + L.abe1);
        }
        if (crit) {
            const float* cth = net.th + L.actor_size;
            gemm_dx_bn(bufB, w2, H1, KC, bufA, cth + L.cga, net.st + L.cmma, net.st + L.cmva, wg ? gc + L.cga : nullptr,
                       wg ? gc + L.cbea : nullptr);
        }
        lds_barrier();
        PH(it == 1 ? 10 : (it == 2 ? 13 : 18));
        if (it == 1) {
            l1_grads<S>(sX, bufA, 0, H1, gc + L.cWs, gc + L.cbs);
            l1_grads<A>(sAct, bufA, H1, HA, gc + L.cWa, gc + L.cba);
        } else if (it == 2) {  // da[r][a] = sum_j dza[r][j] Wa[a][j]
            const float* cth = net.th + L.actor_size;
            narrow_gemm<HA, A, false, false>(bufA + H1, LDA, nullptr, nullptr, cth + L.cWa, 1, HA, nullptr, sDa, nullptr);
        } else {
            l1_grads<S>(sX, bufA, 0, H1, ga + L.aW1, ga + L.ab1);
        }
        lds_barrier();  // (the next pass's first layer overwrites bufA)
        PH(it == 1 ? 11 : (it == 2 ? 14 : 19));
    }
}

struct Span {  // agents [lo, lo + n) of the slabs
    int lo, n;
};
template <int S, int A, bool STATS>
static int launch_t(const avd_mlp_layout* lay, Span sp, int set_mod, const float* theta, const float* stats, float* theta_t,
                    float* stats_t, const float* s, const float* a, const float* r, const float* s2, float gamma, float high,
                    float* grads, float* losses, float tau, float omt, hipStream_t stream) {
    constexpr size_t lds = sizeof(float) * Lds<S, A>::total;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static std::atomic<bool> attr[64];  // (zero-initialised; a race sets the same attribute twice: harmless)
    if (dev < 0 || dev >= 64 || !attr[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)learn_kernel_c<S, A, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("cen_launch: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
            return AVD_E_LAUNCH;
        }
        if (dev >= 0 && dev < 64) attr[dev].store(true, std::memory_order_release);
    }
    // (a span is addressed by offsetting the per-agent arrays; shared sets -- set_mod > 0 -- only ever come with lo == 0)
    const long lo = sp.lo, ts = lay->theta_size, ss = lay->stats_size;
    const long wo = set_mod > 0 ? 0 : lo;
    hipLaunchKernelGGL((learn_kernel_c<S, A, STATS>), dim3(sp.n), dim3(NT), lds, stream, *lay, set_mod, theta + wo * ts, stats + wo * ss,
                       theta_t + wo * ts, stats_t + wo * ss, s + lo * TILE * S, a + lo * TILE * A, r + lo * TILE, s2 + lo * TILE * S, gamma,
                       high, grads + lo * ts, losses ? losses + lo * 2 : nullptr, tau, omt);
    return check_launch(STATS ? "avd_learn_update_f32 (centralized)" : "avd_learn_f32 (centralized)");
}

template <bool STATS>
static int launch_shape(const avd_mlp_layout* lay, Span sp, int set_mod, const float* theta, const float* stats, float* theta_t,
                        float* stats_t, const float* s, const float* a, const float* r, const float* s2, float gamma, float high,
                        float* grads, float* losses, float tau, float omt, hipStream_t stream) {
    if (lay->S == 20 && lay->A == 5)
        return launch_t<20, 5, STATS>(lay, sp, set_mod, theta, stats, theta_t, stats_t, s, a, r, s2, gamma, high, grads, losses, tau, omt, stream);
    if (lay->S == 12 && lay->A == 3)
        return launch_t<12, 3, STATS>(lay, sp, set_mod, theta, stats, theta_t, stats_t, s, a, r, s2, gamma, high, grads, losses, tau, omt, stream);
    set_error("cen_launch: shape S=%d A=%d H1=%d H2=%d Ha=%d is not one of the centralized instantiations", lay->S, lay->A, lay->H1,
              lay->H2, lay->Ha);
    return AVD_E_UNSUPPORTED;
}

// The side stream of the update passes and the events that order it against the caller's stream, per device (created on first use, kept
// for the life of the process).
constexpr int MAX_CHUNKS = 32;
struct Side {
    hipStream_t st = nullptr;
    hipEvent_t learned[MAX_CHUNKS] = {}, join = nullptr;
    bool ready = false;
    std::mutex enqueue;  // one caller at a time records / waits on the device's events (two host threads must not interleave them)
};
static Side* side_stream(hipStream_t of) {
    static Side sides[64];
    static std::mutex mu;
    int dev = 0;
    // the device the CALLER'S stream lives on (the null stream: the current device)
    if ((of ? hipStreamGetDevice(of, &dev) : hipGetDevice(&dev)) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    Side& sd = sides[dev];
    if (!sd.ready) {
        bool ok = hipStreamCreateWithFlags(&sd.st, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < MAX_CHUNKS && ok; ++i) ok = hipEventCreateWithFlags(&sd.learned[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) return nullptr;
        sd.ready = true;
    }
    return &sd;
}

}  // namespace cen

#ifdef AVD_PHASE_TIMING
}  // namespace avd
extern "C" __attribute__((visibility("default"))) int avd_debug_phase_cycles_cen(unsigned long long* h_out, int reset) {
    if (h_out) (void)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(avd::g_phase_cycles), sizeof(unsigned long long) * 32);
    if (reset) {
        unsigned long long z[32] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(avd::g_phase_cycles), z, sizeof(z));
    }
    return 0;
}
namespace avd {
#endif

bool cen_supports(const avd_mlp_layout* lay) {
    return lay->B == TILE && lay->H1 == cen::H1 && lay->H2 == cen::H2 && lay->Ha == cen::HA &&
           ((lay->S == 20 && lay->A == 5) || (lay->S == 12 && lay->A == 3)) &&
           lay->ab3 >= 2 * TILE * cen::H2;  // (the kernel parks 2 x 64 x 160 floats at the head of the gradient row's actor block)
}

int cen_launch(const avd_mlp_layout* lay, int n_agents, int set_mod, const float* theta, const float* stats, const float* theta_t,
               const float* stats_t, const float* s, const float* a, const float* r, const float* s2, float gamma, float high, float* grads,
               float* losses, void* stream) {
    return cen::launch_shape<false>(lay, cen::Span{0, n_agents}, set_mod, theta, stats, (float*)theta_t, (float*)stats_t, s, a, r, s2, gamma,
                                    high, grads, losses, 0.f, 0.f, (hipStream_t)stream);
}

// How cen_launch_update cuts n_agents models into chunks: one learn workgroup per CU and chunk, at most MAX_CHUNKS chunks (the chunk
// grows beyond the CU count instead), one update workgroup per CU (MI355X, 256 CUs, 4096 agents: 6.8 ms per step with 256 / 256, 7.3
// with chunks of 512, 11.4 with 128; 128 or 192 update workgroups: 8.1, 384 / 512: 6.6 against 6.3). Exported as
// avd_learn_update_plan so that a caller can describe the pipeline it measured.
void cen_update_plan(int n_agents, int* chunk_out, int* groups_out) {
    const int cus = fset::cu_count() > 0 ? fset::cu_count() : 256;
    int chunk = cus;
    if (const char* e = AVD_DIAG_ENV("CEN_CHUNK")) chunk = atoi(e) > 0 ? atoi(e) : n_agents;
    int groups = cus;  // update-pass workgroups (optim.hip adam_polyak_rows_kernel)
    if (const char* e = AVD_DIAG_ENV("CEN_GROUPS")) groups = atoi(e);
    if ((n_agents + chunk - 1) / chunk > cen::MAX_CHUNKS) chunk = (n_agents + cen::MAX_CHUNKS - 1) / cen::MAX_CHUNKS;
    *chunk_out = chunk, *groups_out = groups;
}

// avd_learn_update_f32 for the centralized shapes: learn + Adam + Polyak of every agent, theta -> theta_out.
// The update of an agent's 0.5 MB of weights moves 4 MB through one CU's memory pipeline -- applied where the gradients are
// produced (the general kernel's fused form; this kernel's first version: 283 k of 984 k cycles per agent at 12 B / clk / CU, with one
// or two waves per SIMD alike) it is bound by the requests one CU keeps in flight, while the memory system as a whole idles through
// the 70 % of the kernel that is matrix-core work. So the agents are cut into chunks: the learn kernels of the chunks run back to back
// in the caller's stream, each followed by an event; a side stream waits for chunk c's event and runs its Adam + Polyak pass
// (adam_polyak_ranges_kernel over the whole slab row: no LDS, few registers -- its workgroups co-reside on CUs whose LDS is held by
// learn workgroups) under chunk c + 1's MFMAs. The gradients take a round trip through HBM (+1 MB per agent) that the in-kernel form
// did not need; the side stream is joined into the caller's stream by an event (capturable in a hipGraph).
int cen_launch_update(const avd_mlp_layout* lay, int n_agents, const float* theta, const float* stats, float* theta_out, float* theta_t,
                      float* stats_t, float* m, float* v, const int32_t* step, const float* s, const float* a, const float* r,
                      const float* s2, float gamma, float high, float actor_lr, float critic_lr, double tau, float* grads, float* losses,
                      void* stream) {
    hipStream_t main = (hipStream_t)stream;
    const float tauf = (float)tau, omt = (float)(1.0 - tau);
    const long ts = lay->theta_size;
    // one learn workgroup per CU and chunk, one update workgroup per CU (MI355X, 256 CUs, 4096 agents: 6.8 ms per step with 256 / 256,
    // 7.3 with chunks of 512, 11.4 with 128; 128 or 192 update workgroups: 8.1, 384 / 512: 6.6 against 6.3)
    int chunk = 0, groups = 0;
    cen_update_plan(n_agents, &chunk, &groups);
    cen::Side* sd = n_agents > chunk ? cen::side_stream(main) : nullptr;
    std::unique_lock<std::mutex> lock;
    if (sd) lock = std::unique_lock<std::mutex>(sd->enqueue);
    bool forked = false;
    // every return after the first fork joins the side stream back into the caller's stream: in eager mode nothing the caller
    // enqueues next can race with update passes still queued there, and a hipGraph capture is not left with an unjoined fork
    auto join = [&]() -> bool {
        if (!forked) return true;
        forked = false;
        return hipEventRecord(sd->join, sd->st) == hipSuccess && hipStreamWaitEvent(main, sd->join, 0) == hipSuccess;
    };
    int c = 0;
    for (int lo = 0; lo < n_agents; lo += chunk, ++c) {
        const cen::Span sp = {lo, n_agents - lo < chunk ? n_agents - lo : chunk};
        int rc = cen::launch_shape<true>(lay, sp, 0, theta, stats, theta_t, stats_t, s, a, r, s2, gamma, high, grads, losses, tauf, omt, main);
        if (rc) {
            (void)join();
            return rc;
        }
        hipStream_t ust = main;
        if (sd) {
            if (hipEventRecord(sd->learned[c], main) != hipSuccess || hipStreamWaitEvent(sd->st, sd->learned[c], 0) != hipSuccess) {
                (void)join();
                return check_launch("avd_learn_update_f32 (centralized): fork");
            }
            forked = true;
            ust = sd->st;
        }
        const long o = (long)sp.lo * ts;
        rc = ::launch_adam_polyak_rows(lay, sp.n, groups, theta + o, theta_out + o, theta_t + o, m + o, v + o, grads + o, step + sp.lo, actor_lr,
                                       critic_lr, tau, ust);
        if (rc) {
            (void)join();
            return rc;
        }
    }
    if (!join()) return check_launch("avd_learn_update_f32 (centralized): join");
    return AVD_OK;
}

}  // namespace avd
