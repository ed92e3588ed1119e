// Centralized framework, second kernel design (r04): FOUR waves per workgroup and TWO workgroups per CU, without the
// [64][388] buffer of first-layer activations that ties cen::learn_kernel_c (cen.hip) to one workgroup per CU. Same four-pass plan,
// same results up to summation order; what is different:
//   * the first layer is regenerated on the matrix cores where it is consumed (S = 4 L inputs: 3 / 5 MFMAs per 16 x 16 block), in the
//     orientation the consumer needs -- features on the accumulator rows for the forward GEMM's A operand, batch rows there for the
//     weight-gradient GEMM's operand and for the input-gradient GEMM's epilogue (relu mask, d gamma, d beta);
//   * the first layer's own weight gradient dW1 = X^T dz1 is folded into that epilogue (the dz1 tile a lane has just formed is the B
//     operand: 16 / 32 MFMAs per 16 columns), so dz1 is never stored -- except the critic's 64 action columns in pass 2, which the
//     gradient w.r.t. the actions reads back from a [64][68] LDS tile;
//   * LDS: 80 KB per workgroup (layer-2 buffer, tables, batch), so two workgroups share a CU and one's memory phases (weights in, the
//     gradient rows out) run under the other's MFMAs.
// Gradients go to the model's row of the gradient slab (avd_learn_f32's contract); every sum has a fixed order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/avddpg_hip.h"
#include "learn_common.h"

namespace avd {
namespace fset {
int cu_count();  // fset.hip: CUs of the current device (cached per device ordinal)
}
namespace cen2 {

constexpr int NT = 256, NW = NT / 64;
constexpr int H1 = 320, HA = 64, H2 = 160, KC = H1 + HA, NBS = H1 / 16;
constexpr int LDB = ld_of(H2), LDX = HA + 4;
static_assert(LDB == 164, "row stride");

__device__ __forceinline__ int tidx() { return (int)threadIdx.x + opaque_zero(); }

// One token per CU for the fused update's memory phase. Two workgroups share a CU and run the same program on equal work: left alone they
// fall into step -- both in their MFMA phases at half the pipe each, then both streaming Adam operands at half the CU's memory rate each
// (s_memtime stamps: 0.81 M + 0.88 M cycles per model and workgroup, i.e. nothing overlaps). A workgroup takes its CU's token before
// the weight-gradient GEMM whose epilogue streams, and gives it back behind it: the other one waits ONCE, and from then on one streams
// while the other computes. The token is a hint, not a lock the results depend on: a bounded spin, and a stale value costs time only.
__device__ unsigned g_cu_token[4096];
__device__ __forceinline__ unsigned cu_slot() {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    return (((xcc & 15) * 8 + se) * 2 + sh) * 16 + cu;  // < 4096
}
__device__ __forceinline__ void token_take(unsigned slot) {  // (one thread; the workgroup waits at the barrier behind it)
    for (int i = 0; i < 4000; ++i) {  // bounded: ~4000 x 8 k clocks
        if (atomicCAS(&g_cu_token[slot], 0u, 1u) == 0u) return;
        __builtin_amdgcn_s_sleep(127);
    }
}
__device__ __forceinline__ void token_give(unsigned slot) { atomicExch(&g_cu_token[slot], 0u); }

template <int S, int A>
struct Lds {
    static constexpr int bufB = 0, bufX = bufB + TILE * LDB, b1A = bufX + TILE * LDX, invA = b1A + KC, shA = invA + KC, invB = shA + KC,
                         shB = invB + H2, rsB = shB + H2, mmB = rsB + H2, db = mmB + H2, sX = db + H2, sR = sX + TILE * S, sAct = sR + TILE,
                         sY = sAct + TILE * A, sQ = sY + TILE * A, sD = sQ + TILE * A, sA1 = sD + TILE * A, sT = sA1 + TILE * A,
                         sDa = sT + TILE * A, red = sDa + TILE * A, total = red + NW;
    static_assert(sizeof(float) * total <= 80 * 1024, "two workgroups per CU");
};

// tables of a first layer's columns [col0, col0 + H): bias, BN scale and shift
__device__ __forceinline__ void l1_tables(const float* __restrict__ b, const float* __restrict__ g, const float* __restrict__ be,
                                          const float* __restrict__ mm, const float* __restrict__ mv, int H, float* b1, float* inv, float* sh) {
    for (int k = tidx(); k < H; k += NT) {
        const float iv = (1.0f / sqrtf(mv[k] + BN_EPS)) * g[k];
        b1[k] = b[k];
        inv[k] = iv;
        sh[k] = be[k] - mm[k] * iv;
    }
}
__device__ __forceinline__ void coefs_b(const float* __restrict__ g, const float* __restrict__ be, const float* __restrict__ mm,
                                        const float* __restrict__ mv, float* invB, float* shB, float* rsB, float* mmB) {
    for (int k = tidx(); k < H2; k += NT) {
        const float rs = 1.0f / sqrtf(mv[k] + BN_EPS);
        const float iv = rs * g[k];
        invB[k] = iv, shB[k] = be[k] - mm[k] * iv, rsB[k] = rs, mmB[k] = mm[k];
    }
}

// the lane's batch operand of a first-layer MFMA: X[(row0 + lr) K + 4 st + lg] (either operand side: the row index is the lane's lr)
template <int K>
__device__ __forceinline__ void load_xop(float (&x)[(K + 3) / 4], const float* X, int row0, int lr, int lg) {
#pragma unroll
    for (int st = 0; st < (K + 3) / 4; ++st) {
        const int j = 4 * st + lg;
        x[st] = (K % 4 == 0 || j < K) ? X[(row0 + lr) * K + min(j, K - 1)] : 0.f;
    }
}
// the lane's weight operand: W[(4 st + lg) H + col] (an index past K meets a zero batch operand)
template <int K>
__device__ __forceinline__ void load_wop(float (&w)[(K + 3) / 4], const float* __restrict__ W, int H, int col, int lg) {
#pragma unroll
    for (int st = 0; st < (K + 3) / 4; ++st) w[st] = W[min(4 * st + lg, K - 1) * H + col];
}

// Hidden layer forward with the first layer generated on the fly: out[r][n] = relu(sum_k y[r][k] W2[k][n] + b2[n]),
// y[r][k] = bn(relu(sum_j X[r][j] W1[j][k] + b1[k])). Wave w: rows 32 (w & 1) .. + 32, columns 80 (w >> 1) .. + 80 (tiles 0..3:
// columns 4 lr + t of the first 64, tile 4: 64 + lr). Per 16-deep k-block the block's first-layer tile is formed FEATURE-major
// (A' = W1 column operand, B' = batch operand: the accumulator holds features 4 lg + reg of row lr = the GEMM's A operand in its
// permuted reduction order), one block ahead of the MFMAs that consume it; W2 rows three blocks ahead in a register ring.
// FIRST / SNAP as in cen.hip: the critic's state blocks are summed once for passes 1 and 2.
template <int S, int A, bool CRITIC, int FIRST, int SNAP>
__device__ __forceinline__ void gemm_fwd(const float* sX, const float* act, const float* b1A, const float* invA, const float* shA,
                                         const float* __restrict__ W1s, const float* __restrict__ W1a, const float* __restrict__ W2,
                                         const float* __restrict__ b2, float* out, float* __restrict__ snap) {
    #ifndef C2_RING
#define C2_RING 4
#endif
    constexpr int NB = (CRITIC ? KC : H1) / 16, RING = C2_RING, STS = (S + 3) / 4, STA = (A + 3) / 4;
    static_assert(FIRST < NB && SNAP < NB && !(FIRST && SNAP), "resume / snapshot points");
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    const int rh = wave & 1, ch = wave >> 1;
    float xs[2][STS], xa[2][STA];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        load_xop<S>(xs[m], sX, 32 * rh + 16 * m, lr, lg);
        if constexpr (CRITIC) load_xop<A>(xa[m], act, 32 * rh + 16 * m, lr, lg);
    }
    f32x4 acc[2][5];
    float* sp = snap + (32 * rh + 4 * lg) * H2 + 80 * ch;  // element (m, reg, t): + (16 m + reg) H2 + (t < 4 ? 4 lr + t : 64 + lr)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (FIRST > 0) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) acc[m][t][reg] = sp[(16 * m + reg) * H2 + (t < 4 ? 4 * lr + t : 64 + lr)];
            }
        }
    const float* wp = W2 + (long)(4 * lg) * H2 + 80 * ch;  // row 16 blk + 4 lg + jj
    f32x4 rq[RING][4];
    float rs[RING][4];
    float w1[3][STS];  // first-layer weight operands of blocks blk, blk + 1, blk + 2 (index blk % 3)
#define C2_ISSUE(blk, d)                                              \
    {                                                                 \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {            \
            const float* p_ = wp + (long)(16 * (blk) + jj) * H2;      \
            rq[d][jj] = *(const f32x4*)(p_ + 4 * lr);                 \
            rs[d][jj] = p_[64 + lr];                                  \
        }                                                             \
    }
#define C2_W1(blk)                                                                                  \
    {                                                                                               \
        if ((blk) < NBS) load_wop<S>(w1[(blk) % 3], W1s, H1, 16 * (blk) + lr, lg);                  \
        else {                                                                                      \
            float t_[STA];                                                                          \
            load_wop<A>(t_, W1a, HA, 16 * ((blk) - NBS) + lr, lg);                                  \
            _Pragma("unroll") for (int st = 0; st < STA; ++st) w1[(blk) % 3][st] = t_[st];          \
        }                                                                                           \
    }
    // the A operands (x inv + sh of the relu'd first layer) of block blk for the wave's two row tiles
#define C2_GEN(blk, ao)                                                                                                           \
    {                                                                                                                             \
        const f32x4 tb_ = *(const f32x4*)(b1A + 16 * (blk) + 4 * lg), ti_ = *(const f32x4*)(invA + 16 * (blk) + 4 * lg),          \
                    ts_ = *(const f32x4*)(shA + 16 * (blk) + 4 * lg);                                                             \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                                           \
            f32x4 d_ = {0.f, 0.f, 0.f, 0.f};                                                                                      \
            if ((blk) < NBS) {                                                                                                    \
                _Pragma("unroll") for (int st = 0; st < STS; ++st) d_ = MFMA16(w1[(blk) % 3][st], xs[m][st], d_);                 \
            } else {                                                                                                              \
                _Pragma("unroll") for (int st = 0; st < STA; ++st) d_ = MFMA16(w1[(blk) % 3][st], xa[m][st], d_);                 \
            }                                                                                                                     \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) ao[m][e] = fmaf(fmaxf(d_[e] + tb_[e], 0.f), ti_[e], ts_[e]);            \
        }                                                                                                                         \
    }
#pragma unroll
    for (int d = 0; d < RING - 1; ++d)
        if (FIRST + d < NB) C2_ISSUE(FIRST + d, (FIRST + d) % RING);
    C2_W1(FIRST);
    if (FIRST + 1 < NB) C2_W1(FIRST + 1);
    f32x4 acur[2], anext[2];
    C2_GEN(FIRST, acur);
#pragma unroll
    for (int blk = FIRST; blk < NB; ++blk) {
        if constexpr (SNAP > 0) {
            if (blk == SNAP && snap) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int t = 0; t < 5; ++t)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) sp[(16 * m + reg) * H2 + (t < 4 ? 4 * lr + t : 64 + lr)] = acc[m][t][reg];
            }
        }
        if (blk + RING - 1 < NB) C2_ISSUE(blk + RING - 1, (blk + RING - 1) % RING);
        if (blk + 2 < NB) C2_W1(blk + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (blk + 1 < NB) C2_GEN(blk + 1, anext);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[m][t] = MFMA16(acur[m][jj], rq[blk % RING][jj][t], acc[m][t]);
                acc[m][4] = MFMA16(acur[m][jj], rs[blk % RING][jj], acc[m][4]);
            }
        __builtin_amdgcn_sched_barrier(0);
        acur[0] = anext[0], acur[1] = anext[1];
    }
#undef C2_GEN
#undef C2_W1
#undef C2_ISSUE
    const f32x4 bq = *(const f32x4*)(b2 + 80 * ch + 4 * lr);
    const float bs = b2[80 * ch + 64 + lr];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float* o = out + (32 * rh + 16 * m + 4 * lg + reg) * LDB + 80 * ch;
            f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = fmaxf(acc[m][t][reg] + bq[t], 0.f);
            *(f32x4*)(o + 4 * lr) = v;
            o[64 + lr] = fmaxf(acc[m][4][reg] + bs, 0.f);
        }
}

// narrow GEMM (cen.hip): out[r][a] = sum_k x(r, k) W[k wk + a wa] (+ bias[a]); wave w owns row tile w
template <int K, int A, bool BN>
__device__ __forceinline__ void narrow_gemm(const float* X, int ldx, const float* inv, const float* sh, const float* __restrict__ W, int wk,
                                            int wa, const float* __restrict__ bias, float* out) {
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    constexpr int NB = K / 16;
    const float colmask = (lr < A) ? 1.f : 0.f;
    const int ac = min(lr, A - 1);
    float wv[NB][4];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) wv[blk][jj] = W[(16 * blk + 4 * lg + jj) * wk + ac * wa] * colmask;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* xr = X + (wave * 16 + lr) * ldx + 4 * lg;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        f32x4 x = *(const f32x4*)(xr + 16 * blk);
        if (BN) x = x * *(const f32x4*)(inv + 16 * blk + 4 * lg) + *(const f32x4*)(sh + 16 * blk + 4 * lg);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc = MFMA16(x[jj], wv[blk][jj], acc);
    }
    if (lr < A) {
        const float bb = bias ? bias[lr] : 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) out[(wave * 16 + 4 * lg + reg) * A + lr] = acc[reg] + bb;
    }
}

// output layer backward through the BN below it on the matrix cores, in place (cen.hip out_bwd; four waves: column tiles w, w + 4, w + 8)
template <int A>
__device__ __forceinline__ void out_bwd(float* bufB, const float* invB, const float* shB, const float* rsB, const float* mmB, const float* D,
                                        const float* __restrict__ W3, float* __restrict__ gW3, float* __restrict__ gb3,
                                        float* __restrict__ gg, float* __restrict__ gbe, float* db, float* __restrict__ gdb) {
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    constexpr int AS = (A + 3) / 4;
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

// Weight gradient of the hidden layer: dW[k][n] = inv[k] sum_r P[r][k] DZ[r][n] + sh[k] db[n] -> gW[k 160 + n], P = the relu'd first
// layer regenerated ROW-major (A' = batch operand, B' = W1 column operand: the accumulator holds rows 4 lg + reg of feature lr = the
// A operand of P^T DZ in its permuted reduction order). Items = (16-feature tile, column group), dealt round-robin over the four
// waves; NC = 4: the two 64-column groups (a lane holds columns 64 g + 4 lr + t: 16-byte LDS reads and global accesses, a wave
// instruction covers 256 contiguous bytes of four rows), NC = 2: the last 32 columns (128 + 2 lr + t).
// Sink = AdamSink: gW is the tensor's position in the OUTPUT weight slab and every element is updated where it is produced (Adam +
// Polyak, optim.hip's arithmetic). ALL operands of an item (w, w_target, m, v of its four feature rows) are requested one whole item
// ahead -- before the previous item's MFMA loop -- in two register sets used alternately: a wave keeps 16 KB in flight behind its
// MFMAs (with the operands of a row requested a row ahead the update ran at 4 B / clk and workgroup, the latency of every row exposed).
__device__ __forceinline__ const float* sink_wo(const StoreSink&) { return nullptr; }
__device__ __forceinline__ const float* sink_wo(const AdamSink& k) { return k.wo; }
template <int K, int NC, class Sink>
__device__ __forceinline__ void gemm_dw(const float* X, const float* __restrict__ W1, int H, int col0, int ntiles, const float* b1A,
                                        const float* invA, const float* shA, const float* DZ, const float* db, float* __restrict__ gW,
                                        Sink sink) {
    constexpr bool kFused = !std::is_same<Sink, StoreSink>::value;
    constexpr int ST = (K + 3) / 4, NG = NC == 4 ? 2 : 1, CB = NC == 4 ? 0 : 128, GW = 16 * NC;
    static_assert(NC == 4 || NC == 2, "column groups of 64 or 32");
    typedef float vecc __attribute__((ext_vector_type(NC)));
    typedef typename std::conditional<NC == 4, AdamSink::Quad4, AdamSink::Quad>::type QuadT;
    typedef typename std::conditional<kFused, QuadT, int>::type QT;
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    const int items = ntiles * NG;
    float xr[4][ST];
#pragma unroll
    for (int m = 0; m < 4; ++m) load_xop<K>(xr[m], X, 16 * m, lr, lg);
    auto issue = [&](int item, QT(&qq)[4], float(&w1)[ST]) {  // the Adam operands and the first-layer weight operand of an item
        const int kt = item / NG, g = item - kt * NG;
        load_wop<K>(w1, W1, H, 16 * kt + lr, lg);
        if constexpr (kFused) {
            const long base = (gW - sink_wo(sink)) + (long)(col0 + 16 * kt + 4 * lg) * H2 + CB + GW * g + NC * lr;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if constexpr (NC == 4) sink.load4(qq[reg], base + (long)reg * H2);
                else sink.load2(qq[reg], base + (long)reg * H2);
            }
        }
    };
    auto body = [&](int item, QT(&qc)[4], float(&w1c)[ST], QT(&qn)[4], float(&w1n)[ST]) {
        if (item + NW < items) issue(item + NW, qn, w1n);
        const int kt = item / NG, g = item - kt * NG, cb = CB + GW * g + NC * lr;
        const float bq = b1A[col0 + 16 * kt + lr];
        f32x4 acc[NC];
#pragma unroll
        for (int t = 0; t < NC; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < ST; ++st) d = MFMA16(xr[m][st], w1c[st], d);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float pv = fmaxf(d[reg] + bq, 0.f);
                const vecc z = *(const vecc*)(DZ + (16 * m + 4 * lg + reg) * LDB + cb);
#pragma unroll
                for (int t = 0; t < NC; ++t) acc[t] = MFMA16(pv, z[t], acc[t]);
            }
            __builtin_amdgcn_sched_barrier(0);  // (or every LDS read of the item is hoisted to its head)
        }
        const f32x4 iv = *(const f32x4*)(invA + col0 + 16 * kt + 4 * lg), sf = *(const f32x4*)(shA + col0 + 16 * kt + 4 * lg);
        const vecc dbv = *(const vecc*)(db + cb);
        const long base = (kFused ? (gW - sink_wo(sink)) : 0) + (long)(col0 + 16 * kt + 4 * lg) * H2 + cb;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float o[NC];
#pragma unroll
            for (int t = 0; t < NC; ++t) o[t] = fmaf(iv[reg], acc[t][reg], sf[reg] * dbv[t]);
            if constexpr (kFused) {
                if constexpr (NC == 4) sink.update4(qc[reg], base + (long)reg * H2, o);
                else sink.update2(qc[reg], base + (long)reg * H2, o);
            } else {
                vecc v;
#pragma unroll
                for (int t = 0; t < NC; ++t) v[t] = o[t];
                *(vecc*)(gW + base + (long)reg * H2) = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    QT q0[4], q1[4];
    float w10[ST], w11[ST];
    if (wave < items) issue(wave, q0, w10);
    for (int item = wave; item < items; item += 2 * NW) {
        body(item, q0, w10, q1, w11);
        if (item + NW < items) body(item + NW, q1, w11, q0, w10);
    }
}

// Input gradient of the hidden layer + BN / relu backward of the first layer + the first layer's own gradients, for the columns
// [c_begin, c_end) of one first-layer branch (K inputs X, weights W1 [K][H], H = c_end - c_begin columns at table offset c_begin):
//   dy[r][c] = sum_n DZ[r][n] W2[c][n];  p regenerated row-major;  dgamma, dbeta;  dz1 = [p > 0] dy rs g;
//   dW1[j][c] = sum_r X[r][j] dz1[r][c] (folded: the lane's dz1 values are the B operand),  db1[c] = sum_r dz1[r][c].
// PARAMS = false (pass 2: only the gradient w.r.t. the actions is wanted): no parameter gradients; dz1 goes to the LDS tile dzx.
template <int K, bool PARAMS>
__device__ __forceinline__ void gemm_dx(const float* X, const float* __restrict__ W1, const float* b1A, const float* DZ,
                                        const float* __restrict__ W2, int c_begin, int c_end, const float* __restrict__ g,
                                        const float* __restrict__ mm, const float* __restrict__ mv, float* __restrict__ dg,
                                        float* __restrict__ dbe, float* __restrict__ gW1, float* __restrict__ gb1, float* dzx) {
    constexpr int ST = (K + 3) / 4, JT = (K + 15) / 16, NB = H2 / 16;
    const int wave = tidx() >> 6, lane = tidx() & 63, lr = lane & 15, lg = lane >> 4;
    const int H = c_end - c_begin;
    float xr[4][ST];
#pragma unroll
    for (int m = 0; m < 4; ++m) load_xop<K>(xr[m], X, 16 * m, lr, lg);
    f32x4 wc[NB], wn[NB];
    float bnc[3] = {0.f, 0.f, 1.f}, bnn[3] = {0.f, 0.f, 1.f};
    float w1c[ST], w1n[ST];
    int c0 = c_begin + wave * 16;
    if (c0 < c_end) {
        const float* wrow = W2 + (long)(c0 + lr) * H2 + 4 * lg;
#pragma unroll
        for (int q = 0; q < NB; ++q) wc[q] = *(const f32x4*)(wrow + 16 * q);
        bnc[0] = g[c0 + lr - c_begin], bnc[1] = mm[c0 + lr - c_begin], bnc[2] = mv[c0 + lr - c_begin];
        load_wop<K>(w1c, W1, H, c0 - c_begin + lr, lg);
    }
    for (; c0 < c_end; c0 += NW * 16) {
        const int cn = c0 + NW * 16;
        if (cn < c_end) {
            const float* wrow = W2 + (long)(cn + lr) * H2 + 4 * lg;
#pragma unroll
            for (int q = 0; q < NB; ++q) wn[q] = *(const f32x4*)(wrow + 16 * q);
            bnn[0] = g[cn + lr - c_begin], bnn[1] = mm[cn + lr - c_begin], bnn[2] = mv[cn + lr - c_begin];
            load_wop<K>(w1n, W1, H, cn - c_begin + lr, lg);
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
            if (q & 1) __builtin_amdgcn_sched_barrier(0);  // (two blocks of LDS operands in flight, not ten)
        }
        const int c = c0 + lr;
        const float rs = 1.0f / sqrtf(bnc[2] + BN_EPS), gam = bnc[0], mean = bnc[1], bq = b1A[c];
        float sg = 0.f, sb = 0.f, sz = 0.f;
        f32x4 wacc[JT];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) wacc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < ST; ++st) d = MFMA16(xr[m][st], w1c[st], d);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float p = fmaxf(d[reg] + bq, 0.f), dy = acc[m][reg];
                sg = fmaf(dy * (p - mean), rs, sg);
                sb += dy;
                const float dz1 = (p > 0.f) ? dy * (rs * gam) : 0.f;
                if constexpr (PARAMS) {
                    sz += dz1;
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {  // fold operand X[(16 m + 4 lg + reg) K + 16 jt + lr], read where it is used (LDS)
                        const int j = 16 * jt + lr;
                        const float xv = (j < K) ? X[(16 * m + 4 * lg + reg) * K + min(j, K - 1)] : 0.f;
                        wacc[jt] = MFMA16(xv, dz1, wacc[jt]);
                    }
                } else {
                    dzx[(16 * m + 4 * lg + reg) * LDX + c - c_begin] = dz1;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PARAMS) {
            sg += __shfl_xor(sg, 16), sb += __shfl_xor(sb, 16), sz += __shfl_xor(sz, 16);
            sg += __shfl_xor(sg, 32), sb += __shfl_xor(sb, 32), sz += __shfl_xor(sz, 32);
            if (lg == 0) dg[c - c_begin] = sg, dbe[c - c_begin] = sb, gb1[c - c_begin] = sz;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int j = 16 * jt + 4 * lg + reg;
                    if (j < K) gW1[j * H + c - c_begin] = wacc[jt][reg];
                }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) wc[q] = wn[q];
        bnc[0] = bnn[0], bnc[1] = bnn[1], bnc[2] = bnn[2];
#pragma unroll
        for (int st = 0; st < ST; ++st) w1c[st] = w1n[st];
    }
}

__device__ __forceinline__ float block_sum(const float* v, int n, float* red) {
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
__device__ __forceinline__ float warm(const float* __restrict__ base, int lo, int hi) {
    float t = 0.f;
    for (int i = lo + tidx() * 32; i < hi; i += NT * 32) t += base[i];
    return t;
}

// FUSED (avd_learn_update_f32): Adam + Polyak of the two W2 matrices in the epilogue of the weight-gradient GEMM that produces their
// gradient (theta -> theta_out ping-pong); the small tensors go through the gradient slab and adam_polyak_ranges_kernel; the frozen BN
// statistics' soft update happens here too -- gen::learn_kernel_g's contract (workers/trainer.py:472-508, agent/ddpgagent.py:44-53).
template <int S, int A, bool FUSED>
__global__ __launch_bounds__(NT, 2) void learn_kernel_c2(avd_mlp_layout L_arg, int set_mod, const float* __restrict__ theta,
                                                         const float* __restrict__ stats, float* __restrict__ theta_t,
                                                         float* __restrict__ stats_t, const float* __restrict__ s,
                                                         const float* __restrict__ a, const float* __restrict__ r,
                                                         const float* __restrict__ s2, float gamma, float high, float* __restrict__ grads,
                                                         float* __restrict__ losses, UpdArgs upd, int stagger_first, int stagger_sleeps) {
    const avd_mlp_layout* const Lk = (const avd_mlp_layout*)__builtin_amdgcn_kernarg_segment_ptr();
    if (L_arg.theta_size != Lk->theta_size || L_arg.stats_size != Lk->stats_size) __builtin_trap();  // the layout IS argument 0
    extern __shared__ __attribute__((aligned(16))) float smem0[];
    typedef Lds<S, A> O;
    const int agent = blockIdx.x;
    // Two workgroups share a CU and run the same program on equal work: started together they stay in phase -- both in their MFMA
    // phases, then both in their memory phases -- and nothing overlaps. The workgroups that fill the CUs' second slots in the first
    // round (blockIdx in [stagger_first, 2 stagger_first)) therefore start half a model late; every later workgroup inherits the
    // offset of the slot it takes over.
    if (blockIdx.x >= stagger_first && blockIdx.x < 2 * stagger_first)
        for (int i = 0; i < stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    const bool use_token = stagger_first < 0;  // (experiment switch: negative = token instead of stagger)
    const unsigned slot = cu_slot();
    {
        const avd_mlp_layout& L = *Lk;
        float* const smem = smem0;
        const int tid = tidx();
        float *sR = smem + O::sR, *sAct = smem + O::sAct;
        float* ga = grads + (long)agent * L.theta_size;
        float* gc = ga + L.actor_size;
        if (tid < TILE) sR[tid] = r[(long)agent * TILE + tid];
        for (int i = tid; i < TILE * A; i += NT) sAct[i] = a[(long)agent * TILE * A + i];
        if (tid == 0) {
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
        asm volatile("" ::"v"(t));
    }
    float alpha_a = 0.f, alpha_c = 0.f;  // Adam step sizes of this model's update (optim.hip adam_polyak_kernel's arithmetic)
    if constexpr (FUSED) {
        const int t = upd.step[agent];
        const float b1p = (float)pow((double)0.9f, (double)t), b2p = (float)pow((double)0.999f, (double)t);
        const float root = sqrtf(1.0f - b2p);
        alpha_a = (upd.actor_lr * root) / (1.0f - b1p), alpha_c = (upd.critic_lr * root) / (1.0f - b1p);
    }
    PH_INIT();
#pragma nounroll
    for (int it = 0; it < 4; ++it) {
        const avd_mlp_layout& L = *(const avd_mlp_layout*)((const char*)Lk + opaque_zero());
        float* const smem = smem0 + opaque_zero();
        float *bufB = smem + O::bufB, *bufX = smem + O::bufX, *b1A = smem + O::b1A, *invA = smem + O::invA, *shA = smem + O::shA,
              *invB = smem + O::invB, *shB = smem + O::shB, *rsB = smem + O::rsB, *mmB = smem + O::mmB, *db = smem + O::db,
              *sX = smem + O::sX, *sR = smem + O::sR, *sAct = smem + O::sAct, *sY = smem + O::sY, *sQ = smem + O::sQ, *sD = smem + O::sD,
              *sA1 = smem + O::sA1, *sT = smem + O::sT, *sDa = smem + O::sDa, *red = smem + O::red;
        const int tid = tidx();
        const int set = set_mod > 0 ? agent % set_mod : agent;
        const Net net = {theta + (long)set * L.theta_size, stats + (long)set * L.stats_size};
        const Net tgt = {theta_t + (long)set * L.theta_size, stats_t + (long)set * L.stats_size};
        float* ga = grads + (long)agent * L.theta_size;
        float* gc = ga + L.actor_size;
        constexpr float invn = 1.0f / (float)(TILE * A);
        typedef typename std::conditional<FUSED, AdamSink, StoreSink>::type BulkSink;
        BulkSink bulk;
        float* gw2 = ga;  // where gemm_dw "stores": the gradient slab, or the model's slab of theta_out
        if constexpr (FUSED) {
            const long o = (long)agent * L.theta_size;
            gw2 = upd.theta_out + o;
            bulk.wo = gw2, bulk.wi = net.th, bulk.wt = theta_t + o, bulk.m = upd.m + o, bulk.v = upd.v + o;
            bulk.alpha_a = alpha_a, bulk.alpha_c = alpha_c, bulk.tau = upd.tau, bulk.omt = upd.omt, bulk.actor_size = L.actor_size;
        }
        const Net n = (it == 0) ? tgt : net;
        if (it < 2) {
            lds_barrier();
            const float* src = (it == 0 ? s2 : s) + (long)agent * TILE * S;
            for (int i = tid; i < TILE * S; i += NT) sX[i] = src[i];
            lds_barrier();
        }
        PH(20);
        float* const keep_p2 = ga;             // (cen.hip: scratch in the model's own gradient row, whose actor block only pass 3 writes)
        float* const keep_cs = ga + TILE * H2;
        if (it != 1) {  // ---- actor forward (agent/model.py:26-36)
            const float* th = n.th;
            l1_tables(th + L.ab1, th + L.ag1, th + L.abe1, n.st + L.amm1, n.st + L.amv1, H1, b1A, invA, shA);
            coefs_b(th + L.ag2, th + L.abe2, n.st + L.amm2, n.st + L.amv2, invB, shB, rsB, mmB);
            if (it == 3) {
                __syncthreads();  // (tables visible; vmcnt(0) in every wave: pass 2's copy of the activations is complete)
                for (int i = tid; i < TILE * (H2 / 4); i += NT) {
                    const int rr = i / (H2 / 4), c4 = i - rr * (H2 / 4);
                    *(f32x4*)(bufB + rr * LDB + 4 * c4) = *(const f32x4*)(keep_p2 + rr * H2 + 4 * c4);
                }
                lds_barrier();
            } else {
                lds_barrier();
                PH(1);
                gemm_fwd<S, A, false, 0, 0>(sX, nullptr, b1A, invA, shA, th + L.aW1, nullptr, th + L.aW2, th + L.ab2, bufB, nullptr);
                lds_barrier();
                PH(2);
                narrow_gemm<H2, A, true>(bufB, LDB, invB, shB, th + L.aW3, A, 1, th + L.ab3, sQ);
                if (it == 2) {
                    for (int i = tid; i < TILE * (H2 / 4); i += NT) {
                        const int rr = i / (H2 / 4), c4 = i - rr * (H2 / 4);
                        *(f32x4*)(keep_p2 + rr * H2 + 4 * c4) = *(const f32x4*)(bufB + rr * LDB + 4 * c4);
                    }
                }
                lds_barrier();
                for (int i = tid; i < TILE * A; i += NT) {
                    const float t = tanhf(sQ[i]);
                    sT[i] = t, sA1[i] = t * high;
                }
                lds_barrier();
            }
        }
        if (it != 3) {  // ---- critic forward (agent/model.py:63-83)
            const float* th = n.th + L.actor_size;
            const float* act = (it == 1) ? sAct : sA1;
            if (it != 2) l1_tables(th + L.cbs, th + L.cgs, th + L.cbes, n.st + L.cmms, n.st + L.cmvs, H1, b1A, invA, shA);
            l1_tables(th + L.cba, th + L.cga, th + L.cbea, n.st + L.cmma, n.st + L.cmva, HA, b1A + H1, invA + H1, shA + H1);
            coefs_b(th + L.cg3, th + L.cbe3, n.st + L.cmm3, n.st + L.cmv3, invB, shB, rsB, mmB);
            lds_barrier();
            PH(3);
            if (it == 2)
                gemm_fwd<S, A, true, NBS, 0>(sX, act, b1A, invA, shA, th + L.cWs, th + L.cWa, th + L.cW2, th + L.cb2, bufB, keep_cs);
            else
                gemm_fwd<S, A, true, 0, NBS>(sX, act, b1A, invA, shA, th + L.cWs, th + L.cWa, th + L.cW2, th + L.cb2, bufB,
                                             it == 1 ? keep_cs : nullptr);
            lds_barrier();
            PH(5);
            narrow_gemm<H2, A, true>(bufB, LDB, invB, shB, th + L.cW3, A, 1, th + L.cb3, sQ);
            lds_barrier();
        }
        if (it == 0) {
            for (int i = tid; i < TILE * A; i += NT) sY[i] = fmaf(gamma, sQ[i], sR[i / A]);
            if (FUSED || upd.omt != 0.f) {  // (gradients-out launches of the chunked update pass tau / 1 - tau as well)
#pragma clang fp contract(off)
                float* stt = stats_t + (long)set * L.stats_size;
                for (int i = tid; i < L.stats_size; i += NT) stt[i] = net.st[i] * upd.tau + stt[i] * upd.omt;
            }
            continue;
        }
        if (it == 1) {
            for (int i = tid; i < TILE * A; i += NT) {
                const float e = sY[i] - sQ[i];
                sD[i] = -2.0f * e * invn;
                sT[i] = e * e;
            }
            lds_barrier();
            const float lc = block_sum(sT, TILE * A, red) * invn;
            if (tid == 0 && losses) losses[(long)agent * 2 + 0] = lc;
        } else if (it == 2) {
            const float la = -block_sum(sQ, TILE * A, red) * invn;
            if (tid == 0 && losses) losses[(long)agent * 2 + 1] = la;
            for (int i = tid; i < TILE * A; i += NT) sD[i] = -invn;
        } else {
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
        PH(7);
        const float* w2 = wth + (crit ? L.cW2 : L.aW2);
        if (FUSED && wg && use_token) {
            if (tid == 0) token_take(slot);
            lds_barrier();
        }
        if (it == 1) {
            const float* cth = wth;
            gemm_dw<S, 4>(sX, cth + L.cWs, H1, 0, H1 / 16, b1A, invA, shA, bufB, db, gw2 + L.actor_size + L.cW2, bulk);
            gemm_dw<S, 2>(sX, cth + L.cWs, H1, 0, H1 / 16, b1A, invA, shA, bufB, db, gw2 + L.actor_size + L.cW2, bulk);
            gemm_dw<A, 4>(sAct, cth + L.cWa, HA, H1, HA / 16, b1A, invA, shA, bufB, db, gw2 + L.actor_size + L.cW2, bulk);
            gemm_dw<A, 2>(sAct, cth + L.cWa, HA, H1, HA / 16, b1A, invA, shA, bufB, db, gw2 + L.actor_size + L.cW2, bulk);
            if (FUSED && use_token) {
                lds_barrier();
                if (tid == 0) token_give(slot);
            }
            PH(9);
            gemm_dx<S, true>(sX, cth + L.cWs, b1A, bufB, w2, 0, H1, cth + L.cgs, net.st + L.cmms, net.st + L.cmvs, gc + L.cgs, gc + L.cbes,
                             gc + L.cWs, gc + L.cbs, nullptr);
            gemm_dx<A, true>(sAct, cth + L.cWa, b1A, bufB, w2, H1, KC, cth + L.cga, net.st + L.cmma, net.st + L.cmva, gc + L.cga,
                             gc + L.cbea, gc + L.cWa, gc + L.cba, nullptr);
        } else if (it == 2) {  // only the gradient w.r.t. the actions: da[r][a] = sum_j dza[r][j] Wa[a][j]
            const float* cth = wth;
            gemm_dx<A, false>(sA1, cth + L.cWa, b1A, bufB, w2, H1, KC, cth + L.cga, net.st + L.cmma, net.st + L.cmva, nullptr, nullptr,
                              nullptr, nullptr, bufX);
            lds_barrier();
            narrow_gemm<HA, A, false>(bufX, LDX, nullptr, nullptr, cth + L.cWa, 1, HA, nullptr, sDa);
        } else {
            gemm_dw<S, 4>(sX, wth + L.aW1, H1, 0, H1 / 16, b1A, invA, shA, bufB, db, gw2 + L.aW2, bulk);
            gemm_dw<S, 2>(sX, wth + L.aW1, H1, 0, H1 / 16, b1A, invA, shA, bufB, db, gw2 + L.aW2, bulk);
            if (FUSED && use_token) {
                lds_barrier();
                if (tid == 0) token_give(slot);
            }
            PH(17);
            gemm_dx<S, true>(sX, wth + L.aW1, b1A, bufB, w2, 0, H1, wth + L.ag1, net.st + L.amm1, net.st + L.amv1, ga + L.ag1, ga + L.abe1,
                             ga + L.aW1, ga + L.ab1, nullptr);
        }
        lds_barrier();  // (the next pass rewrites the tables and bufB)
        PH(10);
    }
}

template <int S, int A, bool FUSED>
static int launch_t(const avd_mlp_layout* lay, int n, int set_mod, const float* theta, const float* stats, float* theta_t, float* stats_t,
                    const float* s, const float* a, const float* r, const float* s2, float gamma, float high, float* grads, float* losses,
                    const UpdArgs& upd, hipStream_t stream) {
    int stagger_first = 0, stagger_sleeps = 0;
    if (FUSED && n > 2 * fset::cu_count()) {
        stagger_first = fset::cu_count();
        stagger_sleeps = 48;  // x 127 x 64 clocks = ~0.39 M cycles, half of the ~0.8 M a model takes with two workgroups per CU
        if (const char* e = AVD_DIAG_ENV("CEN2_SLEEPS")) stagger_sleeps = atoi(e);
        if (AVD_DIAG_ENV("CEN2_TOKEN")) stagger_first = -1, stagger_sleeps = 0;
    }
    constexpr size_t lds = sizeof(float) * Lds<S, A>::total;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr[64] = {};
    if (dev >= 0 && dev < 64 && !attr[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)learn_kernel_c2<S, A, FUSED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("cen2_launch: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
            return AVD_E_LAUNCH;
        }
        attr[dev] = true;
        if (AVD_DIAG_ENV("CEN2_OCC")) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)learn_kernel_c2<S, A, FUSED>, NT, lds);
            fprintf(stderr, "cen2: %d workgroups per CU (LDS %zu B)\n", nb, lds);
        }
    }
    hipLaunchKernelGGL((learn_kernel_c2<S, A, FUSED>), dim3(n), dim3(NT), lds, stream, *lay, set_mod, theta, stats, theta_t, stats_t, s, a, r,
                       s2, gamma, high, grads, losses, upd, stagger_first, stagger_sleeps);
    return check_launch(FUSED ? "avd_learn_update_f32 (centralized)" : "avd_learn_f32 (centralized)");
}

}  // namespace cen2

#ifdef AVD_PHASE_TIMING
}  // namespace avd
extern "C" int avd_debug_phase_cycles_cen2(unsigned long long* h_out, int reset) {
    if (h_out) (void)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(avd::g_phase_cycles), sizeof(unsigned long long) * 32);
    if (reset) {
        unsigned long long z[32] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(avd::g_phase_cycles), z, sizeof(z));
    }
    return 0;
}
namespace avd {
#endif

int cen2_launch(const avd_mlp_layout* lay, bool fused, int n_agents, int set_mod, const float* theta, const float* stats, float* theta_t,
                float* stats_t, const float* s, const float* a, const float* r, const float* s2, float gamma, float high, float* grads,
                float* losses, const UpdArgs& upd, void* stream) {
#define C2_GO(S_, A_)                                                                                                               \
    return fused ? cen2::launch_t<S_, A_, true>(lay, n_agents, set_mod, theta, stats, theta_t, stats_t, s, a, r, s2, gamma, high, grads, \
                                                losses, upd, (hipStream_t)stream)                                                  \
                 : cen2::launch_t<S_, A_, false>(lay, n_agents, set_mod, theta, stats, theta_t, stats_t, s, a, r, s2, gamma, high, grads, \
                                                 losses, upd, (hipStream_t)stream)
    if (lay->S == 20 && lay->A == 5) C2_GO(20, 5);
    if (lay->S == 12 && lay->A == 3) C2_GO(12, 3);
#undef C2_GO
    set_error("cen2_launch: shape S=%d A=%d is not one of the centralized instantiations", lay->S, lay->A);
    return AVD_E_UNSUPPORTED;
}

}  // namespace avd
