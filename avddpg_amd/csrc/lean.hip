// learn_kernel_l -- Trainer.learn (workers/trainer.py:472-508), optionally fused with Adam x2 + update_target
// (trainer.py:348-356, agent/ddpgagent.py:31-55), for one agent's batch of 64 rows per 256-thread workgroup, built so
// that TWO workgroups share a CU (<= 80 KB of LDS, <= 256 registers per lane).
//
// Why: the fused form of learn_kernel_t (mlp.hip) alternates a matrix-core phase (~130 us per agent) with an HBM
// stream phase (Adam/Polyak over 2.3 MB per agent, ~85 us); with one workgroup per CU nothing overlaps the two
// (tools/probes/overlap.hip: 202 us per tile alone, 138 us per tile with a second workgroup on the CU). What kept
// learn_kernel_t at one workgroup per CU is bufA, the [64][304] f32 image of the first-layer activations (79 KB).
// Here those activations are never stored: the first layers have K = S <= 4 (state) or 1 (action) inputs, so
// relu(x W1 + b1) costs S FMAs + add + max per element and is recomputed, in the lane that needs it, directly as
// the MFMA operand of
//   * the second-layer forward GEMMs   (A operand, row m*16+lr, features 16 blk + 4 lg + jj),
//   * the weight-gradient GEMMs        (A operand, row 4 it + lg, features of the lane's output rows),
//   * the BN/ReLU backward in the input-gradient epilogues (element (row, column) of the accumulator),
// and the input gradient is reduced into the first-layer parameter gradients in that same epilogue instead of being
// written back. What stays in LDS: the two [64][132] second-layer buffers, the batch (states padded to 4 floats),
// the BN coefficient tables. The pass structure (targets / critic / actor-through-critic / actor, critic state blocks
// of pass 1 resumed in pass 2, actor activations of pass 2 kept for pass 3) is that of learn_kernel_t.
//
// Weight-gradient tiles are laid out so that a lane owns 4 CONSECUTIVE columns of 8 rows: the fused Adam/Polyak
// epilogue moves w, w_target, m, v with 16-byte accesses (learn_kernel_t: 8-byte).
#include "learn_common.h"

namespace avd {
namespace lean {

constexpr int FT = 256;  // 4 waves per workgroup; two workgroups per CU = 2 waves per SIMD
constexpr int NW = FT / 64;
constexpr int R = 3;     // W2-operand register ring depth (blocks of 16 k)

// Lane index that LLVM cannot treat as invariant of the pass loop: without this, loop-invariant code motion hoists every
// per-lane address of the loop body (hundreds of 32-bit offsets) above the loop and spills them (learn_common.h opaque_zero).
__device__ __forceinline__ int tid_here() { return (int)threadIdx.x + opaque_zero(); }

struct Lds {
    float *bufB, *bufC;          // [64][LDB] second-layer activations / gradients
    float *xS, *xS2;             // [64][4] states, next states (columns >= S are zero)
    float *invA, *shA;           // [H1 + HA] BN coefficients of the first-layer features
    float *invB, *shB, *w3B, *rsB, *mmB, *db;  // [H2]
    float *scr;                  // [3 * H2] reduction scratch (also the per-tile action-gradient partials)
    float *sAct, *sR, *sY, *sQ, *sD, *sA1, *sT, *sDa;  // [64]
    float *red;                  // [8]
};
__host__ __device__ constexpr int lds_floats(int KC, int H2) {
    return 2 * TILE * ld_of(H2) + 2 * TILE * 4 + 2 * KC + 6 * H2 + 3 * H2 + 8 * TILE + 8;
}
__device__ __forceinline__ Lds carve(float* p, int KC, int H2) {
    Lds l;
    const int ldB = ld_of(H2);
    l.bufB = p, p += TILE * ldB;
    l.bufC = p, p += TILE * ldB;
    l.xS = p, p += TILE * 4;
    l.xS2 = p, p += TILE * 4;
    l.invA = p, p += KC;
    l.shA = p, p += KC;
    l.invB = p, p += H2;
    l.shB = p, p += H2;
    l.w3B = p, p += H2;
    l.rsB = p, p += H2;
    l.mmB = p, p += H2;
    l.db = p, p += H2;
    l.scr = p, p += 3 * H2;
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

// A first layer (Dense + BN): where its parameters, BN statistics and gradients live. Three uniform base pointers plus
// 32-bit offsets: the kernel keeps a handful of 64-bit bases in SGPRs and every per-lane address is base + u32.
struct L1Set {
    const float* th;  // weight slab of the network (actor block or critic block)
    const float* st;  // statistics slab
    float* gr;        // gradient slab of the same block (same offsets as th)
    unsigned oW1, ob1, og, obe;  // offsets into th / gr: W1 [Kin][H], b1, BN gamma, beta
    unsigned omm, omv;           // offsets into st: moving mean, moving variance
    unsigned H;
};

template <int W>
__device__ __forceinline__ void ldv(float (&d)[W], const float* p) {
    if constexpr (W == 4) {
        const f32x4 v = *(const f32x4*)p;
        d[0] = v[0], d[1] = v[1], d[2] = v[2], d[3] = v[3];
    } else if constexpr (W == 2) {
        const f32x2 v = *(const f32x2*)p;
        d[0] = v[0], d[1] = v[1];
    } else {
        d[0] = p[0];
    }
}

// first-layer weights of W consecutive features: w[j][e] = W1[j][k + e], b[e] = b1[k + e]
template <int KIN, int W>
struct FeatW {
    float w[KIN][W], b[W];
};
template <int KIN, int W>
__device__ __forceinline__ void featw_load(FeatW<KIN, W>& f, const L1Set& s, unsigned k) {
#pragma unroll
    for (int j = 0; j < KIN; ++j) ldv<W>(f.w[j], (s.th + (s.oW1 + j * s.H)) + k);  // uniform pointer + u32 lane offset
    ldv<W>(f.b, (s.th + s.ob1) + k);
}
// relu(sum_j x[j] W1[j][k+e] + b1[k+e]): products summed in input order, bias last (Dense = matmul, then bias_add)
template <int KIN, int W>
__device__ __forceinline__ float feat(const FeatW<KIN, W>& f, const float (&x)[KIN], int e) {
    float t = x[0] * f.w[0][e];
#pragma unroll
    for (int j = 1; j < KIN; ++j) t = fmaf(x[j], f.w[j][e], t);
    return fmaxf(t + f.b[e], 0.f);
}
template <int KIN>
__device__ __forceinline__ void ldx(float (&x)[KIN], const float* X, int r) {  // X: [64][4] (KIN > 1) or [64] (KIN == 1)
    if constexpr (KIN == 1) {
        x[0] = X[r];
    } else {
        const f32x4 v = *(const f32x4*)(X + 4 * r);
#pragma unroll
        for (int j = 0; j < KIN; ++j) x[j] = v[j];
    }
}

// BN coefficient tables, in two steps so that the parameter loads can be issued BEFORE a forward GEMM's operand prefetch
// (in-order vmcnt: requested after it, the table phase would wait for the prefetch's HBM round trip) and consumed after it.
struct BnRaw {
    float g, be, mm, mv;
};
__device__ __forceinline__ BnRaw l1_raw(const L1Set& s, unsigned k, unsigned H) {
    const unsigned kk = k < H ? k : H - 1;  // threads past the layer load a valid column and store nothing
    return {(s.th + s.og)[kk], (s.th + s.obe)[kk], (s.st + s.omm)[kk], (s.st + s.omv)[kk]};
}
__device__ __forceinline__ void l1_coefs(const BnRaw& r, float* inv, float* sh, unsigned k, unsigned H) {
    if (k < H) {
        const float iv = (1.0f / sqrtf(r.mv + BN_EPS)) * r.g;
        inv[k] = iv;
        sh[k] = r.be - r.mm * iv;
    }
}

// ------------------------------------------------------------------------------------------
// out[r][n] = relu(sum_k bn(feature(r, k)) * W2[k][n] + b2[n]), feature = first layer(s) recomputed on the fly.
// Features [0, H1) come from `st` (inputs X4[r][0..S)), features [H1, H1 + HA) from `ac` (input act[r]) when CRITIC.
// Wave w owns columns [32 w, 32 w + 32): MFMA tile t holds columns base + 2 lr + t; reduction index of the 4 MFMAs of
// a 16-deep block: k = 16 blk + 4 lg + jj. BN is folded as in learn_kernel_t: p @ (inv (.) W2) + sh . W2.
// FIRST / SNAP: see fast::gemm_fwd (critic state blocks of pass 1 snapshotted, pass 2 resumes with the action blocks).
// ------------------------------------------------------------------------------------------
// Operands a forward GEMM needs before its first MFMA, requested one phase early (before the BN coefficient tables are
// built and the barrier that publishes them): the first R-1 W2 blocks, the first-layer weights of the first two blocks,
// the bias. Nothing here depends on LDS, so the cold HBM/L2 misses overlap the table phase.
template <int S>
struct FwdPre {
    float ring[R][4][2];
    FeatW<S, 4> fs[2];
    FeatW<1, 4> fa[2];
    float bc0, bc1;
};
template <int S, int H1, int HA, int H2, bool CRITIC, int FIRST>
__device__ __forceinline__ void fwd_prefetch(FwdPre<S>& p, const L1Set& st, const L1Set& ac, const float* __restrict__ W2,
                                             const float* __restrict__ b2) {
    constexpr int NSB = H1 / 16, NBLK = CRITIC ? (H1 + HA) / 16 : H1 / 16, N = H2;
    const int tx = tid_here(), wave = tx >> 6, lane = tx & 63, lr = lane & 15, lg = lane >> 4;
    const unsigned col = wave * 32 + 2 * lr, wl = (4 * lg) * N + col;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int blk = FIRST + d;
        if (blk < NBLK) {
            if (blk < NSB)
                featw_load<S, 4>(p.fs[blk & 1], st, 16 * blk + 4 * lg);
            else
                featw_load<1, 4>(p.fa[blk & 1], ac, 16 * (blk - NSB) + 4 * lg);
        }
    }
#pragma unroll
    for (int d = 0; d < R - 1; ++d)
        if (FIRST + d < NBLK)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) ldv<2>(p.ring[(FIRST + d) % R][jj], (W2 + ((FIRST + d) * 16 + jj) * N) + wl);
    p.bc0 = b2[col], p.bc1 = (b2 + 1)[col];
}

template <int S, int H1, int HA, int H2, bool CRITIC, int FIRST, int SNAP>
__device__ __forceinline__ void gemm_fwd(FwdPre<S>& pre, const float* X4, const float* act, const L1Set& st,
                                         const L1Set& ac, const float* inv, const float* sh,
                                         const float* __restrict__ W2, float* out, float* snap, float (&cs_snap)[2]) {
    constexpr int NSB = H1 / 16, NBLK = CRITIC ? (H1 + HA) / 16 : H1 / 16, N = H2, LDO = ld_of(H2);
    static_assert(N == 32 * NW, "two 16-column tiles per wave");
    const int tx = tid_here(), wave = tx >> 6, lane = tx & 63, lr = lane & 15, lg = lane >> 4;
    const int col = wave * 32 + 2 * lr;
    f32x4 acc[4][2];
    float cs[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        cs[t] = (FIRST > 0) ? cs_snap[t] : 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (FIRST > 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 v = *(const f32x2*)(snap + (m * 16 + lg * 4 + j) * LDO + col);
                acc[m][0][j] = v[0], acc[m][1][j] = v[1];
            }
    }
    float xr[4][S], ar[4][1];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (FIRST < NSB) ldx<S>(xr[m], X4, m * 16 + lr);
        if (CRITIC) ldx<1>(ar[m], act, m * 16 + lr);
    }
    const unsigned wl = (4 * lg) * N + col;  // lane part of every W2 address: the block / row parts are uniform
    float(&ring)[R][4][2] = pre.ring;
    auto load_blk = [&](float(&dst)[4][2], int blk) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ldv<2>(dst[jj], (W2 + (blk * 16 + jj) * N) + wl);
    };
    FeatW<S, 4>(&fs)[2] = pre.fs;
    FeatW<1, 4>(&fa)[2] = pre.fa;
    auto load_feat = [&](int blk) {
        if (blk < NSB)
            featw_load<S, 4>(fs[blk & 1], st, 16 * blk + 4 * lg);
        else
            featw_load<1, 4>(fa[blk & 1], ac, 16 * (blk - NSB) + 4 * lg);
    };
    // a[m][jj]: this lane's A operands of the current block. The values of group jj are dead once its 8 MFMAs have been
    // issued, so the next block's group jj is computed into the same registers right behind them (one buffer, not two).
    float a[4][4];
    auto feat_of = [&](int blk, int m, int jj) {
        return (blk < NSB) ? feat<S, 4>(fs[blk & 1], xr[m], jj) : feat<1, 4>(fa[blk & 1], ar[m], jj);
    };
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) a[m][jj] = feat_of(FIRST, m, jj);
    // BN coefficients of the block's 4 features per lane: read one block ahead (the per-block scheduling fence below would
    // otherwise leave every block waiting on its own LDS round trip)
    f32x4 ivs[2], sfs[2];
    ivs[FIRST & 1] = *(const f32x4*)(inv + 16 * FIRST + 4 * lg);
    sfs[FIRST & 1] = *(const f32x4*)(sh + 16 * FIRST + 4 * lg);
#pragma unroll
    for (int blk = FIRST; blk < NBLK; ++blk) {
        if (SNAP > 0 && blk == SNAP && snap) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x2 v;
                    v[0] = acc[m][0][j], v[1] = acc[m][1][j];
                    *(f32x2*)(snap + (m * 16 + lg * 4 + j) * LDO + col) = v;
                }
            cs_snap[0] = cs[0], cs_snap[1] = cs[1];
        }
        // Issue order matters: vmcnt retires in order, so a wait for the first-layer weights (L2 hits, needed next block)
        // also waits for every load issued BEFORE them. They go first, the W2 block (HBM, needed R-1 blocks from now)
        // after them; the fence keeps the compiler from sinking either below the MFMAs.
        if (blk + 2 < NBLK) load_feat(blk + 2);  // into the buffer block blk's operands came from
        if (blk + R - 1 < NBLK) load_blk(ring[(blk + R - 1) % R], blk + R - 1);
        if (blk + 1 < NBLK) {
            ivs[(blk + 1) & 1] = *(const f32x4*)(inv + 16 * (blk + 1) + 4 * lg);
            sfs[(blk + 1) & 1] = *(const f32x4*)(sh + 16 * (blk + 1) + 4 * lg);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 iv = ivs[blk & 1], sf = sfs[blk & 1];
        float bs[4][2];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bs[jj][t] = ring[blk % R][jj][t] * iv[jj];
                cs[t] = fmaf(ring[blk % R][jj][t], sf[jj], cs[t]);
            }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[m][t] = MFMA16(a[m][jj], bs[jj][t], acc[m][t]);
            if (blk + 1 < NBLK)  // next block's group jj (feature weights loaded one block earlier)
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m][jj] = feat_of(blk + 1, m, jj);
        }
        // pin the shift sums here: left alone, LLVM sinks the whole cs chain (only read after the last block) to the end
        // of the GEMM and keeps -- i.e. spills -- every block's W2 operands and coefficients until then
        asm volatile("" : "+v"(cs[0]), "+v"(cs[1]));
        __builtin_amdgcn_sched_barrier(0);       // nothing moves across blocks: bounds the live ranges
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        cs[t] += __shfl_xor(cs[t], 16);
        cs[t] += __shfl_xor(cs[t], 32);
    }
    cs[0] += pre.bc0, cs[1] += pre.bc1;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 o;
            o[0] = fmaxf(acc[m][0][j] + cs[0], 0.f);
            o[1] = fmaxf(acc[m][1][j] + cs[1], 0.f);
            *(f32x2*)(out + (m * 16 + lg * 4 + j) * LDO + col) = o;
        }
}

// ------------------------------------------------------------------------------------------
// One 16-column tile of the input gradient of a second layer, taken all the way to the first-layer parameter gradients:
//   dy[r][c]   = sum_n DZ[r][n] * W2[c0g + lr][n]                  (c = this lane's column, r = 16 m + 4 lg + reg)
//   dgamma[c]  = sum_r dy (p - mean) rs;   dbeta[c] = sum_r dy      p = relu(x[r] . W1[:, c] + b1[c]) recomputed
//   dz1[r][c]  = dy * rs * gamma * (p > 0)
//   dW1[j][c]  = sum_r x[r][j] dz1[r][c];  db1[c] = sum_r dz1[r][c]
//   WANT_DA: dap[r] = sum_c dz1[r][c] * W1[0][c]  (KIN == 1: gradient w.r.t. the action input), this tile's share
// cl = column index inside the layer `s` (c0g = cl + the layer's first row in W2).
// ------------------------------------------------------------------------------------------
// dx_load requests the tile's operands (the 16 W2 rows, the columns' first-layer weights and BN parameters), dx_run does the
// rest: callers put other work between the two.
template <int KIN, int N>
struct DxOps {
    f32x4 wc[N / 16];
    FeatW<KIN, 1> fw;
    float gam, mean, var;
};
template <int KIN, int N>
__device__ __forceinline__ void dx_load(DxOps<KIN, N>& o, const float* __restrict__ W2row /* W2 + c0g * N */, const L1Set& s,
                                        int cl) {
    const int lane = tid_here() & 63, lr = lane & 15, lg = lane >> 4;
    const unsigned wrow = lr * N + 4 * lg, c = cl + lr;
#pragma unroll
    for (int q = 0; q < N / 16; ++q) o.wc[q] = *(const f32x4*)((W2row + 16 * q) + wrow);
    featw_load<KIN, 1>(o.fw, s, c);
    o.gam = (s.th + s.og)[c], o.mean = (s.st + s.omm)[c], o.var = (s.st + s.omv)[c];
}
template <int KIN, int N, int LDZ, bool WANT_DA, class Sink>
__device__ __forceinline__ void dx_run(const DxOps<KIN, N>& o, const float* DZ, const float* Xin, const L1Set& s, int cl,
                                       bool write_grads, Sink sink, float* dap) {
    constexpr int NB = N / 16;
    const int lane = tid_here() & 63, lr = lane & 15, lg = lane >> 4;
    const unsigned c = cl + lr;
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a[2][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) a[0][m] = *(const f32x4*)(DZ + (m * 16 + lr) * LDZ + 4 * lg);
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        if (q + 1 < NB)
#pragma unroll
            for (int m = 0; m < 4; ++m) a[(q + 1) & 1][m] = *(const f32x4*)(DZ + (m * 16 + lr) * LDZ + 16 * (q + 1) + 4 * lg);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = MFMA16(a[q & 1][m][jj], o.wc[q][jj], acc[m]);
    }
    const float rs = 1.0f / sqrtf(o.var + BN_EPS), mean = o.mean;
    const float rg = rs * o.gam;
    float sg = 0.f, sb = 0.f, ab = 0.f, aw[KIN];
#pragma unroll
    for (int j = 0; j < KIN; ++j) aw[j] = 0.f;
    float da[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = m * 16 + lg * 4 + j;
            float x[KIN];
            ldx<KIN>(x, Xin, r);
            const float p = feat<KIN, 1>(o.fw, x, 0);
            const float dy = acc[m][j];
            sg = fmaf(dy * (p - mean), rs, sg);
            sb += dy;
            const float dz1 = (p > 0.f) ? dy * rg : 0.f;
            ab += dz1;
#pragma unroll
            for (int i = 0; i < KIN; ++i) aw[i] = fmaf(x[i], dz1, aw[i]);
            if (WANT_DA) da[m][j] = dz1 * o.fw.w[0][0];
        }
    if (write_grads) {
        sg += __shfl_xor(sg, 16), sg += __shfl_xor(sg, 32);
        sb += __shfl_xor(sb, 16), sb += __shfl_xor(sb, 32);
        ab += __shfl_xor(ab, 16), ab += __shfl_xor(ab, 32);
#pragma unroll
        for (int i = 0; i < KIN; ++i) aw[i] += __shfl_xor(aw[i], 16), aw[i] += __shfl_xor(aw[i], 32);
        if (lg == 0) {
            sink.put((s.gr + s.og) + c, sg);
            sink.put((s.gr + s.obe) + c, sb);
            sink.put((s.gr + s.ob1) + c, ab);
#pragma unroll
            for (int i = 0; i < KIN; ++i) sink.put((s.gr + (s.oW1 + i * s.H)) + c, aw[i]);
        }
    }
    if constexpr (WANT_DA) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = da[m][j];
                v += __shfl_xor(v, 1), v += __shfl_xor(v, 2), v += __shfl_xor(v, 4), v += __shfl_xor(v, 8);
                if (lr == 0) dap[m * 16 + lg * 4 + j] = v;
            }
    }
}

__device__ __forceinline__ const float* sink_base(const StoreSink&, const float* gW) { return gW; }
__device__ __forceinline__ const float* sink_base(const AdamSink& s, const float*) { return s.wo; }
__device__ __forceinline__ float alpha_of(const StoreSink&, bool) { return 0.f; }
__device__ __forceinline__ float alpha_of(const AdamSink& s, bool critic) { return critic ? s.alpha_c : s.alpha_a; }

// ------------------------------------------------------------------------------------------
// Weight gradient of a second layer (+ fused Adam/Polyak) and its input gradient, interleaved per block of 64 features:
//   dW2[k][n] = inv[k] * sum_r feature(r, k) * DZ[r][n] + sh[k] * db[n]
// Wave w: column half ch = w & 1 (MFMA tile t holds columns 64 ch + 4 lr + t), row half rh = w >> 1 (tile ta holds the
// features k0 + 32 rh + 2 i + ta, i = MFMA row). A lane ends up with rows k0 + 32 rh + 8 lg + {0..7}, 4 consecutive
// columns each: 8 float4 pieces per array for the fused update.
// Order inside a block, chosen for the in-order vmcnt counter (a wait for a load also waits for every older load AND
// store):  [first-layer weights of the NEXT block | Adam operands of this block]  MFMA loop  [operands of the block's
// input-gradient tile (columns k0 + 16 w .., dx_load): W2 rows the update is pulling through L2 right now]
// update + stores  |  input-gradient tile (dx_run).  The MFMA loop thus never waits behind the operand stream, and the
// next block's loop starts on weights that arrived a block ago, underneath this block's stores.
// ------------------------------------------------------------------------------------------
template <int S, int H1, int HA, int H2, bool CRITIC, class Sink, class SmallSink>
__device__ __forceinline__ void gemm_dw_dx(const float* X4, const float* act, const L1Set& st, const L1Set& ac,
                                           const float* inv, const float* sh, const float* DZ, const float* db,
                                           float* __restrict__ gW, Sink sink, const float* __restrict__ W2,
                                           SmallSink small) {
    constexpr int N = H2, LDZ = ld_of(H2), K = CRITIC ? H1 + HA : H1;
    constexpr bool kFused = !std::is_same<Sink, StoreSink>::value;
    static_assert(N == 128 && H1 % 64 == 0 && HA <= 64 && HA % 16 == 0, "tile plan");
    const int tx = tid_here(), wave = tx >> 6, lane = tx & 63, lr = lane & 15, lg = lane >> 4;
    const int ch = wave & 1, rh = wave >> 1;
    const int col = 64 * ch + 4 * lr;
    const f32x4 dbc = *(const f32x4*)(db + col);
    const float* dp = DZ + lg * LDZ + col;
    const long tens = gW - sink_base(sink, gW);  // position of this W2 tensor in the slab (uniform)
    const float alpha = alpha_of(sink, CRITIC);
    typedef typename std::conditional<kFused, AdamSink::Quad4, int>::type QuadT;

    // A lane's 8 rows are kl + 8 lg + {0..7}: one u32 lane offset `lo`, the row inside the group is an immediate.
    auto load_quads = [&](QuadT(&q)[8], unsigned lo) {
        if constexpr (kFused) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                AdamSink::load4p(q[j], (sink.wi + tens + j * N) + lo, (sink.wt + tens + j * N) + lo,
                                 (sink.m + tens + j * N) + lo, (sink.v + tens + j * N) + lo);
        }
    };
    auto mfma_loop = [&](f32x4(&acc)[2][4], const auto& fw, const float* Xin, auto kin_tag) {
        constexpr int KIN = decltype(kin_tag)::value;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[ta][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float x[2][KIN];
        f32x4 dz[2];
        ldx<KIN>(x[0], Xin, lg);
        dz[0] = *(const f32x4*)dp;
#pragma unroll
        for (int it = 0; it < TILE / 4; ++it) {
            if (it + 1 < TILE / 4) {
                ldx<KIN>(x[(it + 1) & 1], Xin, 4 * (it + 1) + lg);
                dz[(it + 1) & 1] = *(const f32x4*)(dp + 4 * (it + 1) * LDZ);
            }
            float pa[2];
            pa[0] = feat<KIN, 2>(fw, x[it & 1], 0), pa[1] = feat<KIN, 2>(fw, x[it & 1], 1);
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[ta][t] = MFMA16(pa[ta], dz[it & 1][t], acc[ta][t]);
        }
    };
    // rows [R0, R1) of the lane's group of 8 (row rj = 2 j + ta holds feature kl + 8 lg + rj)
    auto epilogue = [&](const f32x4(&acc)[2][4], const QuadT(&q)[8], int kl, unsigned lo, auto r0_tag, auto r1_tag) {
        constexpr int R0 = decltype(r0_tag)::value, R1 = decltype(r1_tag)::value;
        const f32x4 iv0 = *(const f32x4*)(inv + kl + 8 * lg), iv1 = *(const f32x4*)(inv + kl + 8 * lg + 4);
        const f32x4 sf0 = *(const f32x4*)(sh + kl + 8 * lg), sf1 = *(const f32x4*)(sh + kl + 8 * lg + 4);
#pragma unroll
        for (int rj = R0; rj < R1; ++rj) {
            const int j = rj >> 1, ta = rj & 1;
            const float iv = rj < 4 ? iv0[rj & 3] : iv1[rj & 3], sf = rj < 4 ? sf0[rj & 3] : sf1[rj & 3];
            float o[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = fmaf(iv, acc[ta][t][j], sf * dbc[t]);
            if constexpr (kFused) {
                sink.update4p(q[rj], alpha, (sink.wo + tens + rj * N) + lo, (sink.wt + tens + rj * N) + lo,
                              (sink.m + tens + rj * N) + lo, (sink.v + tens + rj * N) + lo, o);
            } else {
                f32x4 ov;
                ov[0] = o[0], ov[1] = o[1], ov[2] = o[2], ov[3] = o[3];
                *(f32x4*)((gW + rj * N) + lo) = ov;
            }
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 8> I8;

    FeatW<S, 2> fw;
    featw_load<S, 2>(fw, st, 32 * rh + 2 * lr);
#pragma nounroll
    for (int k0 = 0; k0 < H1; k0 += 64) {  // blocks of state features
        const int kl = k0 + 32 * rh;       // first feature row of this wave's half block
        const unsigned lo = (unsigned)((kl + 8 * lg) * N + col);
        // the input-gradient tile's W2 rows are requested together with the Adam operands, which cover the same rows: the
        // two requests meet on the same in-flight L2 lines (with 64 workgroups per XCD streaming, a line fetched now is
        // gone from the 4 MB L2 a few microseconds later: requested after the update, these rows came from HBM again)
        DxOps<S, N> dxo;
        dx_load<S, N>(dxo, W2 + (k0 + 16 * wave) * N, st, k0 + 16 * wave);
        QuadT q[8];
        load_quads(q, lo);
        f32x4 acc[2][4];
        mfma_loop(acc, fw, X4, std::integral_constant<int, S>());
        epilogue(acc, q, kl, lo, I0(), I8());
        // next block's first-layer weights (the last block reloads its own): behind this block's stores in vmcnt order, with
        // the input-gradient tile between them and their first use
        featw_load<S, 2>(fw, st, min(kl + 64, H1 - 64 + 32 * rh) + 2 * lr);
        dx_run<S, N, LDZ, false>(dxo, DZ, X4, st, k0 + 16 * wave, true, small, nullptr);
    }
    if constexpr (CRITIC) {  // the critic's action features (their own first layer): rows H1 .. H1 + HA of W2
        const int kl = H1 + 32 * rh;
        // rows of whole lanes (lg >= (HA - 32) / 8 in the upper half) lie past K: those lanes work on the rows of the lower
        // lanes again (valid memory) and store nothing
        const bool lane_ok = kl + 8 * lg < K;
        const int lge = lane_ok ? lg : lg - 2;
        const unsigned lo = (unsigned)((kl + 8 * lge) * N + col);
        FeatW<1, 2> fwa;
        featw_load<1, 2>(fwa, ac, min(32 * rh + 2 * lr, HA - 2));  // rows past HA: clamped, never stored
        QuadT q[8];
        load_quads(q, lo);
        f32x4 acc[2][4];
        mfma_loop(acc, fwa, act, std::integral_constant<int, 1>());
        const bool have = H1 + 16 * wave < K;
        if (lane_ok) epilogue(acc, q, kl, lo, I0(), I8());
        if (have) {
            DxOps<1, N> dxo;
            dx_load<1, N>(dxo, W2 + (H1 + 16 * wave) * N, ac, 16 * wave);
            dx_run<1, N, LDZ, false>(dxo, DZ, act, ac, 16 * wave, true, small, nullptr);
        }
    }
}

// ------------------------------------------------------------------------------------------
// width-1 output layer backward through the BN below it (learn_common.h out_layer_backward with a 3*K scratch:
// the two row halves meet through LDS, same summation order). Ends with a barrier.
// ------------------------------------------------------------------------------------------
template <int K, class Sink>
__device__ __forceinline__ void out_backward(const float* P, int ldp, const float* inv, const float* sh, const float* d,
                                             const float* w3, const float* rsl, const float* mml, float* DZ, int ldz,
                                             float* scr, float* __restrict__ gW3, float* __restrict__ gg,
                                             float* __restrict__ gbe, Sink sink) {
    static_assert(2 * K == FT, "two row halves");
    const int tx = tid_here(), part = tx / K, k = tx - part * K;
    float dw = 0.f, dgm = 0.f, dbt = 0.f;
    const float wk = w3[k], iv = inv[k], s = sh[k], rs = rsl[k], mean = mml[k];
    for (int rb = part * 32; rb < (part + 1) * 32; rb += 8) {
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
    if (gW3) {
        if (part == 1) scr[k] = dw, scr[K + k] = dgm, scr[2 * K + k] = dbt;
        lds_barrier();
        if (part == 0) {
            sink.put(gW3 + k, (0.f + dw) + scr[k]);
            sink.put(gg + k, (0.f + dgm) + scr[K + k]);
            sink.put(gbe + k, (0.f + dbt) + scr[2 * K + k]);
        }
    }
    lds_barrier();
}

struct L2Raw {
    float g, be, mm, mv, w3;
};
__device__ __forceinline__ L2Raw l2_raw(const float* __restrict__ g, const float* __restrict__ be,
                                        const float* __restrict__ mm, const float* __restrict__ mv,
                                        const float* __restrict__ w3, unsigned H2, unsigned k) {
    const unsigned kk = k < H2 ? k : H2 - 1;
    return {g[kk], be[kk], mm[kk], mv[kk], w3[kk]};
}
__device__ __forceinline__ void l2_coefs(const L2Raw& r, Lds& l, unsigned H2, unsigned k) {
    if (k < H2) {
        const float rs = 1.0f / sqrtf(r.mv + BN_EPS);
        const float iv = rs * r.g;
        l.invB[k] = iv, l.shB[k] = r.be - r.mm * iv, l.w3B[k] = r.w3, l.rsB[k] = rs, l.mmB[k] = r.mm;
    }
}

template <int S, int H1, int H2, int HA, bool FUSED>
__global__ __launch_bounds__(FT, 2) void learn_kernel_l(avd_mlp_layout L_arg, int set_mod, const float* __restrict__ theta,
                                                         const float* __restrict__ stats, float* __restrict__ theta_t,
                                                         float* __restrict__ stats_t, const float* __restrict__ s,
                                                         const float* __restrict__ a, const float* __restrict__ r,
                                                         const float* __restrict__ s2, float gamma, float high,
                                                         float* __restrict__ grads, float* __restrict__ losses,
                                                         UpdArgs upd) {
    static_assert(H1 <= FT && HA <= FT && 2 * H2 == FT && S <= 4, "widths");
    constexpr int KC = H1 + HA, LDB = ld_of(H2);
    // The layout (43 offsets) is the kernel's FIRST argument, i.e. the first bytes of the kernarg segment. It is read
    // through that pointer, re-derived inside each pass behind an opaque zero: as a plain by-value argument LLVM loads
    // every field it will ever need up front and keeps ~40 SGPRs alive -- and spilled into VGPR lanes -- all kernel long.
    const avd_mlp_layout* const Lk = (const avd_mlp_layout*)__builtin_amdgcn_kernarg_segment_ptr();
    if (L_arg.theta_size != Lk->theta_size || L_arg.stats_size != Lk->stats_size) __builtin_trap();  // the layout IS argument 0
    const avd_mlp_layout& L = *Lk;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Lds l = carve(smem, KC, H2);
    const int agent = blockIdx.x;
    const int set = set_mod > 0 ? agent % set_mod : agent;
    const Net net = {theta + (long)set * L.theta_size, stats + (long)set * L.stats_size};
    const Net tgt = {theta_t + (long)set * L.theta_size, stats_t + (long)set * L.stats_size};
    float* g = grads + (long)agent * L.theta_size;
    float* ga = g;
    float* gc = g + L.actor_size;
    const int tid = threadIdx.x;
    const StoreSink sink;
    typedef typename std::conditional<FUSED, AdamSink, StoreSink>::type BulkSink;
    BulkSink bulk;
    float* gw2 = g;
    if constexpr (FUSED) {
        const long o = (long)agent * L.theta_size;
        gw2 = upd.theta_out + o;
        bulk.wo = gw2, bulk.wi = net.th, bulk.wt = theta_t + o, bulk.m = upd.m + o, bulk.v = upd.v + o;
        bulk.alpha_a = bulk.alpha_c = 0.f;  // Adam step sizes: computed where the update runs (passes 1 and 3)
        bulk.tau = upd.tau, bulk.omt = upd.omt, bulk.actor_size = L.actor_size;
    }
    constexpr float invn = 1.0f / (float)TILE;
    constexpr int LPR = FT / 64;

    for (int i = tid; i < TILE * 4; i += FT) {
        const int rr = i >> 2, j = i & 3;
        l.xS[i] = (j < S) ? s[(long)agent * TILE * S + rr * S + j] : 0.f;
        l.xS2[i] = (j < S) ? s2[(long)agent * TILE * S + rr * S + j] : 0.f;
    }
    if (tid < TILE) {
        l.sAct[tid] = a[(long)agent * TILE + tid];
        l.sR[tid] = r[(long)agent * TILE + tid];
    }
    if (tid == 0) {  // alignment padding of the gradient slab
        for (int i = L.ab3 + 1; i < L.actor_size; ++i) ga[i] = 0.f;
        for (int i = L.cb3 + 1; i < L.theta_size - L.actor_size; ++i) gc[i] = 0.f;
    }
    PH_INIT();
    PH_CLK_INIT();
    float cs_snap[2] = {0.f, 0.f};  // lane-local shift sums of the critic's state blocks, pass 1 -> pass 2

    // pass 0: targets (y); pass 1: critic loss + gradient; pass 2: actor -> critic, gradient wrt the action;
    // pass 3: actor gradient (second-layer activations of pass 2 kept in bufB)      (workers/trainer.py:492-506)
#pragma nounroll
    for (int it = 0; it < 4; ++it) {
        const int tid = tid_here();
        const avd_mlp_layout& L = *(const avd_mlp_layout*)((const char*)Lk + opaque_zero());  // re-read per pass, see above
        const Net n = (it == 0) ? tgt : net;
        const float* X = (it == 0) ? l.xS2 : l.xS;
        const float* ath = n.th;
        const float* cth = n.th + L.actor_size;
        // first layers of this pass's networks (gradient destinations only matter in passes 1 and 3)
        const L1Set aL1 = {ath, n.st, ga, (unsigned)L.aW1, (unsigned)L.ab1, (unsigned)L.ag1, (unsigned)L.abe1,
                           (unsigned)L.amm1, (unsigned)L.amv1, H1};
        const L1Set cS1 = {cth, n.st, gc, (unsigned)L.cWs, (unsigned)L.cbs, (unsigned)L.cgs, (unsigned)L.cbes,
                           (unsigned)L.cmms, (unsigned)L.cmvs, H1};
        const L1Set cA1 = {cth, n.st, gc, (unsigned)L.cWa, (unsigned)L.cba, (unsigned)L.cga, (unsigned)L.cbea,
                           (unsigned)L.cmma, (unsigned)L.cmva, HA};
        float* const aP2 = l.bufB;
        float* const cP2 = (it == 2) ? l.bufC : l.bufB;
        float* const bP2 = (it == 2) ? l.bufC : l.bufB;
        float* const bDZ = (it == 1) ? l.bufB : l.bufC;
        // Each forward is [request the GEMM's first operands | build the BN coefficient tables | barrier | GEMM]; the
        // sequence is written out per variant so that the prefetched registers never meet at a control-flow join
        // (joined, the register allocator spills them).
        if (it == 3) {  // ---- pass 3: only the second layer's coefficient tables of the actor are rebuilt
            l2_coefs(l2_raw(ath + L.ag2, ath + L.abe2, n.st + L.amm2, n.st + L.amv2, ath + L.aW3, H2, tid), l, H2, tid);
            lds_barrier();
            PH(1);
        } else if (it != 1) {  // ---- actor (agent/model.py:26-36)
            const BnRaw r1 = l1_raw(aL1, tid, H1);
            const L2Raw r2 = l2_raw(ath + L.ag2, ath + L.abe2, n.st + L.amm2, n.st + L.amv2, ath + L.aW3, H2, tid);
            FwdPre<S> fp;
            fwd_prefetch<S, H1, HA, H2, false, 0>(fp, aL1, aL1, ath + L.aW2, ath + L.ab2);
            l1_coefs(r1, l.invA, l.shA, tid, H1);
            l2_coefs(r2, l, H2, tid);
            lds_barrier();
            PH(1);
            gemm_fwd<S, H1, HA, H2, false, 0, 0>(fp, X, nullptr, aL1, aL1, l.invA, l.shA, ath + L.aW2, aP2, nullptr, cs_snap);
            lds_barrier();
            PH(2);
            const float z = out_layer_row(aP2, LDB, l.invB, l.shB, l.w3B, ath[L.ab3], H2);
            if (tid % LPR == 0) {
                const float t = tanhf(z);
                l.sT[tid / LPR] = t;
                l.sA1[tid / LPR] = t * high;
            }
            lds_barrier();
            PH(3);
        }
        if (it != 3) {  // ---- critic (agent/model.py:63-83)
            const float* act = (it == 1) ? l.sAct : l.sA1;
            if (it == 2) {  // resumes from pass 1's state-block sums; the ACTOR's state coefficients stay in invA/shA (pass 3)
                const BnRaw ra = l1_raw(cA1, tid, HA);
                const L2Raw r2 = l2_raw(cth + L.cg3, cth + L.cbe3, n.st + L.cmm3, n.st + L.cmv3, cth + L.cW3, H2, tid);
                FwdPre<S> fp;
                fwd_prefetch<S, H1, HA, H2, true, H1 / 16>(fp, cS1, cA1, cth + L.cW2, cth + L.cb2);
                l1_coefs(ra, l.invA + H1, l.shA + H1, tid, HA);
                l2_coefs(r2, l, H2, tid);
                lds_barrier();
                PH(4);
                gemm_fwd<S, H1, HA, H2, true, H1 / 16, 0>(fp, X, act, cS1, cA1, l.invA, l.shA, cth + L.cW2, cP2, l.bufC,
                                                          cs_snap);
            } else {
                const BnRaw rs1 = l1_raw(cS1, tid, H1), ra = l1_raw(cA1, tid, HA);
                const L2Raw r2 = l2_raw(cth + L.cg3, cth + L.cbe3, n.st + L.cmm3, n.st + L.cmv3, cth + L.cW3, H2, tid);
                FwdPre<S> fp;
                fwd_prefetch<S, H1, HA, H2, true, 0>(fp, cS1, cA1, cth + L.cW2, cth + L.cb2);
                l1_coefs(rs1, l.invA, l.shA, tid, H1);
                l1_coefs(ra, l.invA + H1, l.shA + H1, tid, HA);
                l2_coefs(r2, l, H2, tid);
                lds_barrier();
                PH(4);
                gemm_fwd<S, H1, HA, H2, true, 0, H1 / 16>(fp, X, act, cS1, cA1, l.invA, l.shA, cth + L.cW2, cP2,
                                                          it == 1 ? l.bufC : nullptr, cs_snap);
            }
            lds_barrier();
            PH(5);
            const float q = out_layer_row(cP2, LDB, l.invB, l.shB, l.w3B, cth[L.cb3], H2);
            if (tid % LPR == 0) l.sQ[tid / LPR] = q;
            lds_barrier();
            PH(6);
        }
        if (it == 0) {  // TD target, no done mask (trainer.py:494)
            if (tid < TILE) l.sY[tid] = fmaf(gamma, l.sQ[tid], l.sR[tid]);
            if constexpr (FUSED) {  // the frozen BN statistics take part in the soft update too (ddpgagent.py:44-53)
#pragma clang fp contract(off)
                float* stt = stats_t + (long)set * L.stats_size;
                for (int i = tid; i < L.stats_size; i += FT) stt[i] = net.st[i] * upd.tau + stt[i] * upd.omt;
            }
            lds_barrier();
            continue;
        }
        // ---- d(loss)/d(output-layer input) for this pass
        if (it == 1) {
            if (tid < TILE) {
                const float e = l.sY[tid] - l.sQ[tid];
                l.sD[tid] = -2.0f * e * invn;
                l.sT[tid] = e * e;
            }
            lds_barrier();
            const float lc = block_sum64(l.sT, l.red) * invn;
            const float db3 = block_sum64(l.sD, l.red);
            if (tid == 0) {
                sink.put(gc + L.cb3, db3);
                if (losses) losses[(long)agent * 2 + 0] = lc;
            }
        } else if (it == 2) {
            const float la = -block_sum64(l.sQ, l.red) * invn;
            if (tid == 0 && losses) losses[(long)agent * 2 + 1] = la;
            if (tid < TILE) l.sD[tid] = -invn;
            lds_barrier();
        } else {
            if (tid < TILE) {
                const float t = l.sT[tid];
                l.sD[tid] = l.sDa[tid] * high * (1.0f - t * t);
            }
            lds_barrier();
            const float db3 = block_sum64(l.sD, l.red);
            if (tid == 0) sink.put(ga + L.ab3, db3);
        }
        const bool crit = (it != 3), wg = (it != 2);
        float* gout = crit ? gc : ga;
        out_backward<H2>(bP2, LDB, l.invB, l.shB, l.sD, l.w3B, l.rsB, l.mmB, bDZ, LDB, l.scr,
                         wg ? gout + (crit ? L.cW3 : L.aW3) : nullptr, gout + (crit ? L.cg3 : L.ag2),
                         gout + (crit ? L.cbe3 : L.abe2), sink);
        PH(it == 1 ? 7 : (it == 2 ? 12 : 15));
        if (wg) {
            col_sums(bDZ, LDB, H2, l.db, gout + (crit ? L.cb2 : L.ab2), sink);
            if constexpr (FUSED) {  // lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t), as adam_polyak_kernel (optim.hip) computes it
                const int t = upd.step[agent];
                const float b1p = (float)pow((double)0.9f, (double)t), b2p = (float)pow((double)0.999f, (double)t);
                const float root = sqrtf(1.0f - b2p);
                // wave-uniform values computed on the vector unit: moved to SGPRs so that they cost no vector register for
                // the rest of the kernel
                bulk.alpha_a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
                    __builtin_bit_cast(int, (upd.actor_lr * root) / (1.0f - b1p))));
                bulk.alpha_c = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
                    __builtin_bit_cast(int, (upd.critic_lr * root) / (1.0f - b1p))));
            }
            lds_barrier();
            PH(it == 1 ? 8 : 16);
            if (crit)
                gemm_dw_dx<S, H1, HA, H2, true>(X, l.sAct, cS1, cA1, l.invA, l.shA, bDZ, l.db,
                                                gw2 + L.actor_size + L.cW2, bulk, cth + L.cW2, sink);
            else
                gemm_dw_dx<S, H1, HA, H2, false>(X, nullptr, aL1, aL1, l.invA, l.shA, bDZ, l.db, gw2 + L.aW2, bulk,
                                                 ath + L.aW2, sink);
            lds_barrier();
            PH(it == 1 ? 9 : 17);
        } else {  // pass 2: gradient w.r.t. the action only -- the critic's action-feature columns, three 16-column tiles
            const int wave = tid >> 6;
            if (wave < HA / 16) {
                DxOps<1, H2> dxo;
                dx_load<1, H2>(dxo, cth + L.cW2 + (H1 + 16 * wave) * H2, cA1, 16 * wave);
                dx_run<1, H2, LDB, true>(dxo, bDZ, l.sA1, cA1, 16 * wave, false, sink, l.scr + wave * TILE);
            }
            lds_barrier();
            PH(13);
            if (tid < TILE) {
                float v = l.scr[tid];
#pragma unroll
                for (int w = 1; w < HA / 16; ++w) v += l.scr[w * TILE + tid];
                l.sDa[tid] = v;
            }
            lds_barrier();
            PH(14);
        }
    }
    if constexpr (FUSED) {
        // Adam + Polyak of the small tensors (biases, BN gamma/beta, first and output layers: everything outside the two W2
        // matrices, ~6 % of the parameters), whose gradients the passes above left in this agent's gradient slab. Same
        // arithmetic, element for element, as adam_polyak_ranges_kernel (optim.hip), which learn_kernel_t launches for this.
#pragma clang fp contract(off)
        __syncthreads();  // every wave's gradient stores are visible to the workgroup
        const int tid = tid_here();
        const int T = L.theta_size, a0 = L.aW2, a1 = L.aW2 + H1 * H2, c0 = L.actor_size + L.cW2, c1 = c0 + KC * H2;
        const int n0 = a0 / 4, n1 = (c0 - a1) / 4, n2 = (T - c1) / 4;
        const f32x4* g4 = (const f32x4*)g;
        const f32x4* wi4 = (const f32x4*)bulk.wi;
        f32x4* wo4 = (f32x4*)bulk.wo;
        f32x4* wt4 = (f32x4*)bulk.wt;
        f32x4* m4 = (f32x4*)bulk.m;
        f32x4* v4 = (f32x4*)bulk.v;
        for (int j = tid; j < n0 + n1 + n2; j += FT) {
            const unsigned i = j < n0 ? j : (j < n0 + n1 ? a1 / 4 + (j - n0) : c1 / 4 + (j - n0 - n1));
            const float alpha = ((int)i * 4 < L.actor_size) ? bulk.alpha_a : bulk.alpha_c;
            const f32x4 gg = g4[i];
            f32x4 w = wi4[i], wt = wt4[i], mm = m4[i], vv = v4[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mm[k] = mm[k] + (gg[k] - mm[k]) * (1.0f - 0.9f);
                vv[k] = vv[k] + (gg[k] * gg[k] - vv[k]) * (1.0f - 0.999f);
                w[k] = w[k] - (mm[k] * alpha) / (sqrtf(vv[k]) + 1e-7f);
                wt[k] = w[k] * bulk.tau + wt[k] * bulk.omt;
            }
            wo4[i] = w, wt4[i] = wt, m4[i] = mm, v4[i] = vv;
        }
        if (upd.act_out) {
            // The agent's NEXT action, actor(next state) with the weights this workgroup has just written (still in L2; the
            // separate launch re-reads 143 KB per agent from HBM): the building blocks and the summation order of
            // mlp_rows_kernel, so the value is the same bit for bit. bufB / bufC are free by now.
            __syncthreads();
            float* h1 = l.bufB;           // H1
            float* h2 = h1 + H1;          // H2
            float* part = h2 + H2;        // 256
            float* xin = part + NTHREADS; // S
            const float* tho = bulk.wo;
            if (threadIdx.x < S) xin[threadIdx.x] = upd.act_x[(long)agent * upd.act_x_stride + threadIdx.x];
            __syncthreads();
            gemv_relu(xin, S, tho + L.aW1, tho + L.ab1, H1, part, h1);
            bn_apply(h1, H1, tho + L.ag1, tho + L.abe1, net.st + L.amm1, net.st + L.amv1);
            __syncthreads();
            gemv_relu(h1, H1, tho + L.aW2, tho + L.ab2, H2, part, h2);
            bn_apply(h2, H2, tho + L.ag2, tho + L.abe2, net.st + L.amm2, net.st + L.amv2);
            __syncthreads();
            const float z = block_dot(h2, tho + L.aW3, 1, H2, part) + tho[L.ab3];
            if (threadIdx.x == 0) upd.act_out[agent] = tanhf(z) * high;
        }
    }
    PH_CLK_END();
}

template <int S, bool FUSED>
static int launch(const avd_mlp_layout* lay, int n_agents, int set_mod, const float* theta, const float* stats,
                  float* theta_t, float* stats_t, const float* s, const float* a, const float* r, const float* s2,
                  float gamma, float high, float* grads, float* losses, UpdArgs upd, void* stream) {
    constexpr int H1 = 256, H2 = 128, HA = 48;
    size_t lds = sizeof(float) * lds_floats(H1 + HA, H2);
    if (const char* kb = AVD_DIAG_ENV("LEAN_LDS_KB")) {  // diagnostics: force 1 workgroup per CU; never below what the kernel needs
        const size_t want = (size_t)atoi(kb) * 1024;
        if (want > lds && want <= 160 * 1024) lds = want;
    }
    static_assert(sizeof(float) * lds_floats(H1 + HA, H2) <= 80 * 1024, "two workgroups per CU");
    hipError_t e = hipFuncSetAttribute((const void*)learn_kernel_l<S, H1, H2, HA, FUSED>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        set_error("lean_launch: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        return AVD_E_LAUNCH;
    }
    hipLaunchKernelGGL((learn_kernel_l<S, H1, H2, HA, FUSED>), dim3(n_agents), dim3(FT), lds, (hipStream_t)stream, *lay,
                       set_mod, theta, stats, theta_t, stats_t, s, a, r, s2, gamma, high, grads, losses, upd);
    return check_launch(FUSED ? "avd_learn_update_f32 (lean)" : "avd_learn_f32 (lean)");
}

}  // namespace lean

#ifdef AVD_PHASE_TIMING
}  // namespace avd
extern "C" __attribute__((visibility("default"))) int avd_debug_phase_cycles_lean(unsigned long long* h_out, int reset) {
    if (h_out) (void)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(avd::g_phase_cycles), sizeof(unsigned long long) * 32);
    if (reset) {
        unsigned long long z[32] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(avd::g_phase_cycles), z, sizeof(z));
    }
    return 0;
}
namespace avd {
#endif

int lean_launch(const avd_mlp_layout* lay, bool fused, int n_agents, int set_mod, const float* theta, const float* stats,
                float* theta_t, float* stats_t, const float* s, const float* a, const float* r, const float* s2,
                float gamma, float high, float* grads, float* losses, UpdArgs upd, void* stream) {
    if (!(lay->A == 1 && lay->H1 == 256 && lay->H2 == 128 && lay->Ha == 48 && (lay->S == 3 || lay->S == 4) &&
          lay->B == TILE)) {
        set_error("lean_launch: built for the reference widths 256/128/48, A=1, B=64, S in {3,4}");
        return AVD_E_UNSUPPORTED;
    }
#define AVD_LEAN(SS, FF)                                                                                              \
    return lean::launch<SS, FF>(lay, n_agents, set_mod, theta, stats, theta_t, stats_t, s, a, r, s2, gamma, high, grads, \
                                losses, upd, stream)
    if (lay->S == 4) {
        if (fused) AVD_LEAN(4, true);
        AVD_LEAN(4, false);
    }
    if (fused) AVD_LEAN(3, true);
    AVD_LEAN(3, false);
#undef AVD_LEAN
}

}  // namespace avd
