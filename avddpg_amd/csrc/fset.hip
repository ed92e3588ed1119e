// Fused shared-weight-set learner ("fset"): Trainer.learn (workers/trainer.py:472-508) + the federated mean over the
// platoons (src/server/federated.py:47-63, 99-118; workers/trainer.py:400-431) for agents that SHARE their networks
// (interfrl with every step federated: the P copies of vehicle m's networks stay identical, trainer.py:121-128), at the
// reference widths 256 / 128 / 48 (src/config.py:112-117), as a chain of PERSISTENT kernels: a workgroup (8 waves, one per
// CU) is bound to ONE weight set for its whole life, keeps that set's weights on chip and streams the agents' 64-row
// batches through them.
//
// Why: the layer-wise GEMM chain of wide.hip moves every activation of the P x 64 rows of a set through HBM (8.4 ms at
// 4096 x 5 agents, 4 % of the bf16 matrix peak), and the per-agent f32 kernel (lean.hip) is bound by the f32 MFMA rate
// (8.2 ms). Here
//   * weights never move after the first tile: the BN-folded second-layer weights (+ the folded bias as one more feature)
//     sit in LDS (head_kernel, 70 / 82 KB) or in the waves' registers as MFMA fragments (dx_kernel, dxa_kernel);
//   * the first layers (K = S or 1 inputs) are evaluated ON the matrix cores with split operands
//     [x_hi | x_lo | x_hi | 1 | 1] . [w_hi | w_hi | w_lo | b_hi | b_lo] (one v_mfma_f32_32x32x16_bf16 per 32 x 32 tile,
//     products exact, 2^-16 relative), in whichever orientation the consumer needs, and never touch memory: as the B operand
//     of the second-layer MFMA after one exchange between the two lanes of a row (head_kernel), or directly as the A operand
//     of a product that sums over the tile's rows (an accumulator tile IS that operand: dw_kernel, dx_kernel);
//   * gradients never leave the registers per tile: W2 gradients (80 accumulator registers per lane), first-layer and BN
//     gradients and the output-layer sums are accumulated over all the tiles a workgroup sees and written ONCE, as one
//     partial per workgroup / wave; finalize_* sums the partials of a set in a fixed order (deterministic) and applies the
//     BN folds.
// Between kernels only per-row scalars (target action, TD target, mu, dmu) and the bf16 second-layer gradient dZ2
// (16 KB per agent) go through HBM.
//
// Passes (one launch each; "head" = first layer + second-layer GEMM + output layer [+ its backward]):
//   0 prep (bf16 weight images, folded vectors), pack (split input fragments of s and s')
//   1 head  target actor(s')                -> a'                   5 dw   critic: G += P1^T . dZ2c (+ column sums = db2)
//   2 head  target critic(s', a')           -> y = r + gamma q      6 dx   critic state tiles, dxa critic action tiles:
//   3 head  actor(s)                        -> mu                          first-layer / BN gradient sums
//   4 head  critic(s, a), seed 2(q-y)/N     -> dZ2c, T1, loss       7 head critic(s, mu), seed -1/N -> dmu (the product with
//   8 head  actor(s), seed dmu*high*(1-t^2) -> dZ2a, T1                    the action rows of W2 taken from registers), loss
//   9 dw actor     10 dx actor     11 finalize_small, finalize_w2
// GEMM operands are bf16 (8 significant bits), accumulation / parameters / gradients f32: the same numerics class as
// avd_learn_shared_bf16 (wide.hip), against which, against the exact f32 per-agent kernel + federated mean and against the
// float64 oracle it is tested (tests/test_gpu_fset.py).
#include "fset_common.h"

namespace avd {
namespace fset {

// ---- first layer on the matrix cores -------------------------------------------------------------------------------
// z1[row][f] = sum_k x[row][k] W1[k][f] + b1[f] as ONE 32x32x16 bf16 MFMA with both operands split into bf16 pairs:
//   k slot      0..3        4..7        8..11       12     13     14 15
//   input  x:   x_hi        x_lo        x_hi        1      1      0  0        (lane half 0 holds slots 0..7, half 1 8..15)
//   weight w:   w_hi        w_hi        w_lo        b_hi   b_lo   0  0
// Every product of two bf16 numbers is exact in f32; what is dropped is x_lo*w_lo and the third bf16 of each split:
// 2^-16 relative. The same fragment serves as the A or as the B operand (row / column = lane & 31 either way).
__device__ __forceinline__ bf16x8 make_xf(float x0, float x1, float x2, float x3, int h) {
    const bf16 a0 = (bf16)x0, a1 = (bf16)x1, a2 = (bf16)x2, a3 = (bf16)x3;
    const bf16 l0 = (bf16)(x0 - (float)a0), l1 = (bf16)(x1 - (float)a1), l2 = (bf16)(x2 - (float)a2), l3 = (bf16)(x3 - (float)a3);
    const bf16 one = (bf16)1.f, zero = (bf16)0.f;
    bf16x8 v;
    v[0] = a0, v[1] = a1, v[2] = a2, v[3] = a3;
    v[4] = h ? one : l0, v[5] = h ? one : l1, v[6] = h ? zero : l2, v[7] = h ? zero : l3;
    return v;
}
__device__ __forceinline__ bf16x8 make_wf(float w0, float w1, float w2, float w3, float b, int h) {
    const bf16 a0 = (bf16)w0, a1 = (bf16)w1, a2 = (bf16)w2, a3 = (bf16)w3, bh = (bf16)b;
    const bf16 l0 = (bf16)(w0 - (float)a0), l1 = (bf16)(w1 - (float)a1), l2 = (bf16)(w2 - (float)a2), l3 = (bf16)(w3 - (float)a3);
    const bf16 bl = (bf16)(b - (float)bh), zero = (bf16)0.f;
    bf16x8 v;
    v[0] = h ? l0 : a0, v[1] = h ? l1 : a1, v[2] = h ? l2 : a2, v[3] = h ? l3 : a3;
    v[4] = h ? bh : a0, v[5] = h ? bl : a1, v[6] = h ? zero : a2, v[7] = h ? zero : a3;
    return v;
}

// one network's operands (device pointers; `th` already points at the net's block of the slab of set 0)
struct NetP {
    const float* th;  // theta or theta_t (+ actor_size for a critic); set stride th_stride
    const float* st;  // stats or stats_t; set stride st_stride
    long th_stride, st_stride;
    int oW1, ob1;     // first (state) layer weights [S][H1] / bias, offsets into th
    int oWa, oba;     // critic action layer [1][HA] / bias (critic only)
    int oga, omva;    // critic action-branch BN gamma (th) / moving variance (st): dx_kernel's action-gradient mode
    const bf16* W2T;  // [sets][H2][KW]  bf16(inv1[f] * W2[f][n]) transposed, features K, K + 1 = b2' (folded bias) as a bf16 pair: head_kernel's LDS image
    const bf16* W2R;  // [sets][KP][H2]  bf16(W2[f][n]), rows >= K zero: dx_kernel's resident operand (online nets only)
    const float* vec; // [sets][VEC]: b2'[128] = b2 + sh1 . W2, c3[128] = inv2 * w3, d3 = b3 + sh2 . w3
};

// feature tile `ft` (32 features) of a net's first layer as a weight fragment; tiles >= 8 are the critic's action layer
template <int S, class NET>
__device__ __forceinline__ bf16x8 layer1_wf(const NetP& n, const float* th, int ft, int r, int h, bool with_one = false) {
    if (32 * ft + r >= NET::K) {  // past the real features: zeros, or (dw_kernel) the constant-one feature K = relu(0 x + 1)
        return make_wf(0.f, 0.f, 0.f, 0.f, (with_one && 32 * ft + r == NET::K) ? 1.f : 0.f, h);
    }
    if (NET::critic && ft >= 8) {
        const int f = 32 * (ft - 8) + r;
        return make_wf(th[n.oWa + f], 0.f, 0.f, 0.f, th[n.oba + f], h);
    }
    const int f = 32 * ft + r;
    const float* W = th + n.oW1;
    return make_wf(W[f], W[H1 + f], W[2 * H1 + f], S > 3 ? W[3 * H1 + f] : 0.f, th[n.ob1 + f], h);
}

template <int S>
__device__ __forceinline__ void load_x(const float* xa, int row, float (&x)[4]) {
    if (S == 4) {
        const float4 v = *(const float4*)(xa + 4 * row);
        x[0] = v.x, x[1] = v.y, x[2] = v.z, x[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = k < S ? xa[S * row + k] : 0.f;
    }
}

// The first-layer input fragments of every batch row, built ONCE per learn call (every pass of the chain and each of the 8
// waves of a workgroup would otherwise redo the split): out[row][h] = make_xf(x[row], h), 32 bytes per row.
// `extra` (a or r, one float per row; lane half 0 checks it) and x are also tested for non-finite values: this file is built
// with -fno-honor-nans and relu() turns a NaN into 0, so a non-finite input would otherwise vanish instead of propagating
// like it does in the f32 engines: *bad is set and finalize writes NaN gradients.
template <int S>
__global__ __launch_bounds__(256) void pack_x_kernel(const float* x, const float* extra, long rows, bf16* out, int* bad) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * rows) return;
    float v[4];
    load_x<S>(x, (int)(i >> 1), v);
    bool nf = not_finite(v[0]) || not_finite(v[1]) || not_finite(v[2]) || not_finite(v[3]);
    if (!(i & 1)) nf = nf || not_finite(extra[i >> 1]);
    if (nf) atomicOr(bad, 1);
    *(bf16x8*)(out + 8 * i) = make_xf(v[0], v[1], v[2], v[3], (int)(i & 1));
}

// ---- operand preparation: one block per (output column n, net, set) ---------------------------------------------------
struct PrepArgs {
    avd_mlp_layout L;
    const float *theta, *stats, *theta_t, *stats_t;
    bf16* W2T[4];  // net 0 actor, 1 critic, 2 target actor, 3 target critic
    bf16* W2R[4];  // online nets only
    float* vec[4];
    int* bad;      // set when a weight / BN statistic is not finite
};
__global__ __launch_bounds__(320) void prep_kernel(const PrepArgs a) {
    __shared__ float red[320];
    const int n = blockIdx.x, net = blockIdx.y, set = blockIdx.z, f = threadIdx.x;
    const bool critic = net & 1, target = net >= 2;
    const avd_mlp_layout& L = a.L;
    const float* th = (target ? a.theta_t : a.theta) + (long)set * L.theta_size + (critic ? L.actor_size : 0);
    const float* st = (target ? a.stats_t : a.stats) + (long)set * L.stats_size;
    const int K = critic ? Critic::K : Actor::K, KP = critic ? Critic::KP : Actor::KP, KW = critic ? Critic::KW : Actor::KW;
    const int oW2 = critic ? L.cW2 : L.aW2, ob2 = critic ? L.cb2 : L.ab2, oW3 = critic ? L.cW3 : L.aW3, ob3 = critic ? L.cb3 : L.ab3;
    float shw = 0.f;
    {
        float w = 0.f, inv = 0.f;
        if (f < K) {
            int og, obe, omm, omv, ff = f;
            if (!critic) og = L.ag1, obe = L.abe1, omm = L.amm1, omv = L.amv1;
            else if (f < H1) og = L.cgs, obe = L.cbes, omm = L.cmms, omv = L.cmvs;
            else og = L.cga, obe = L.cbea, omm = L.cmma, omv = L.cmva, ff = f - H1;
            inv = (1.0f / sqrtf(st[omv + ff] + BN_EPS)) * th[og + ff];
            const float sh = th[obe + ff] - st[omm + ff] * inv;
            w = th[oW2 + (long)f * H2 + n];
            shw = sh * w;
            if (not_finite(w) || not_finite(inv) || not_finite(sh)) atomicOr(a.bad, 1);
        }
        if (f < KW && f != K && f != K + 1) a.W2T[net][((long)set * H2 + n) * KW + f] = (bf16)(inv * w);  // (K, K + 1: the bias, below)
        if (!target && f < KP) a.W2R[net][((long)set * KP + f) * H2 + n] = (bf16)w;
    }
    red[f] = shw;
    __syncthreads();
    if (f < 64) red[f] += red[f + 256];  // 320 = 256 + 64
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (f < o) red[f] += red[f + o];
        __syncthreads();
    }
    float* vec = a.vec[net] + (long)set * VEC;
    const int og2 = critic ? L.cg3 : L.ag2, obe2 = critic ? L.cbe3 : L.abe2, omm2 = critic ? L.cmm3 : L.amm2, omv2 = critic ? L.cmv3 : L.amv2;
    if (f == 0) {
        vec[n] = th[ob2 + n] + red[0];
        if (not_finite(vec[n]) || not_finite(th[oW3 + n]) || not_finite(th[ob3])) atomicOr(a.bad, 1);
        // the folded bias b2' rides in the weight image as features K and K + 1 = its bf16 pair (hi, lo), met by two
        // constant-one activations: 2^-17 relative (one bf16 alone would put 2^-9 of the bias into every pre-activation)
        const bf16 bh = (bf16)vec[n];
        a.W2T[net][((long)set * H2 + n) * KW + K] = bh;
        a.W2T[net][((long)set * H2 + n) * KW + K + 1] = (bf16)(vec[n] - (float)bh);
        const float inv2 = (1.0f / sqrtf(st[omv2 + n] + BN_EPS)) * th[og2 + n];
        vec[H2 + n] = inv2 * th[oW3 + n];
    }
    if (n == 0) {  // d3 = b3 + sum_n sh2[n] w3[n]
        __syncthreads();
        float v = 0.f;
        if (f < H2) {
            const float inv2 = (1.0f / sqrtf(st[omv2 + f] + BN_EPS)) * th[og2 + f];
            v = (th[obe2 + f] - st[omm2 + f] * inv2) * th[oW3 + f];
        }
        red[f] = v;
        __syncthreads();
        for (int o = 64; o > 0; o >>= 1) {
            if (f < o) red[f] += red[f + o];
            __syncthreads();
        }
        if (f == 0) vec[2 * H2] = th[ob3] + red[0];
    }
}

// ---- head: first layer -> second-layer GEMM -> output layer [-> its backward] ------------------------------------------
enum HeadMode { OUT_TANH = 0, OUT_TD = 1, HEAD_CRITIC = 2, HEAD_CONST = 3, HEAD_ACTOR = 4 };
struct HeadArgs {
    NetP net;
    int n_agents, n_sets;
    const bf16* xf;    // [n_agents][64][2][8] packed first-layer input fragments of the states (pack_x_kernel)
    const float* act;  // [n_agents][64] the critic's action input (a, a' or mu)
    const float* r;    // OUT_TD: rewards [n_agents][64]
    const float* yin;  // HEAD_CRITIC: TD targets; HEAD_ACTOR: dmu
    const float* aw;   // per-agent factor on the loss seeds (weighted federated mean) or NULL
    float* out;        // OUT_*: per-row result
    bf16* dz;          // HEAD_CRITIC / HEAD_ACTOR: dZ2 [n_agents][64][128]
    float* dmu;        // HEAD_CONST: dLa/dmu per row [n_agents][64] (the action gradient, through the critic's action branch)
    float* part;       // HEAD_*: [grid][8 waves][128] sums T1 = sum_rows seed * p2 per output column
    float* part_s;     // HEAD_*: [grid][8 waves][2] sums of the seeds and of the loss terms
    float gamma, high, inv_n;
};

// The set's BN-folded second-layer weights sit in LDS for the workgroup's whole life (bf16, [column n][feature f], plus the
// folded bias b2' as feature K: 70 / 82 KB); after the one barrier behind that load the 8 waves never synchronise again:
// wave w owns rows [32 (w & 1), +32) of every 4th tile of the workgroup. Per feature tile it evaluates the first layer for
// ITS rows on the matrix cores (lane = batch row, registers = features), trades feature groups between the two lanes of a
// row (v_permlane32_swap) so that each holds 8 consecutive features -- which IS the B operand of the second-layer MFMA,
// no LDS round trip for activations -- and multiplies it into all four 32-column tiles (A = weight fragments from LDS, one
// ds_read_b128 per MFMA). Result: lane = batch row, 64 registers = its second-layer outputs, so the output layer is a
// per-lane dot product (+ one exchange with the partner lane) and dZ2 leaves in 16-byte row-major pieces.
// Waves per workgroup: the modes without the 64 T1 accumulators fit 168 registers -> three waves per SIMD instead of two
// (more issue-stall and LDS-latency cover); part / part_s are indexed [grid][8 waves] and only exist for the 8-wave modes.
__host__ __device__ constexpr int head_waves(int mode) { return (mode == OUT_TANH || mode == OUT_TD) ? 12 : 8; }

template <int S, class NET, int MODE>
__global__ __launch_bounds__(64 * head_waves(MODE)) void head_kernel(const HeadArgs p) {
    constexpr int NW = head_waves(MODE), NTH = 64 * NW;
    constexpr int K = NET::K, KW = NET::KW, NKS = K / 16, NFT = NET::NFT, LD = KW + 8;  // LD/2 = 4 (mod 8) dwords: conflict-free b128
    __shared__ __attribute__((aligned(16))) bf16 wimg[H2 * LD];
    __shared__ __attribute__((aligned(16))) float c3s[H2];
    __shared__ __attribute__((aligned(16))) bf16x8 wfs[NFT * 64];  // first-layer weight fragments [tile][lane] (registers are short)
    constexpr bool AG = (MODE == HEAD_CONST);  // action-gradient epilogue (pass 7; the critic)
    constexpr int LDA = H2 + 8;
    __shared__ __attribute__((aligned(16))) bf16 w2a[AG ? 64 * LDA : 8];  // raw W2 rows of the 48 action features (+ 16 zero rows)
    __shared__ __attribute__((aligned(16))) float ctab[AG ? 64 : 1][4];    // per action feature: wa, ba, inv_a * wa
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5, q = w >> 1, rh = w & 1;
    const int set = blockIdx.x % p.n_sets, j0 = blockIdx.x / p.n_sets, J = gridDim.x / p.n_sets, P = p.n_agents / p.n_sets;
    const float* th = p.net.th + (long)set * p.net.th_stride;
    {
        const bf16* src = p.net.W2T + (long)set * H2 * KW;
        for (int i = tid; i < H2 * (KW / 8); i += NTH) {
            const int n = i / (KW / 8), c = i - n * (KW / 8);
            *(uint4*)(wimg + n * LD + 8 * c) = *(const uint4*)(src + (long)n * KW + 8 * c);
        }
    }
    const float* vec = p.net.vec + (long)set * VEC;
    if (tid < H2) c3s[tid] = vec[H2 + tid];
    const float d3 = vec[2 * H2];
    if (AG) {
        const bf16* src = p.net.W2R + ((long)set * NET::KP + H1) * H2;
        for (int i = tid; i < 64 * (H2 / 8); i += NTH) {
            const int f = i / (H2 / 8), c = i - f * (H2 / 8);
            *(uint4*)(w2a + f * LDA + 8 * c) = *(const uint4*)(src + (long)f * H2 + 8 * c);
        }
        if (tid < 64) {
            const bool ok = tid < HA;
            const float* st = p.net.st + (long)set * p.net.st_stride;
            const float wa = ok ? th[p.net.oWa + tid] : 0.f, ba = ok ? th[p.net.oba + tid] : 0.f;
            const float ia = ok ? (1.0f / sqrtf(st[p.net.omva + tid] + BN_EPS)) * th[p.net.oga + tid] : 0.f;
            ctab[tid][0] = wa, ctab[tid][1] = ba, ctab[tid][2] = ia * wa, ctab[tid][3] = 0.f;
        }
    }
    for (int ft = w; ft < NFT; ft += NW) wfs[ft * 64 + lane] = layer1_wf<S, NET>(p.net, th, ft, r, h);
    bf16x8 onef;  // activation fragment of the bias step: features K, K + 1 = 1 (the bias's bf16 pair), K + 2.. = 0
#pragma unroll
    for (int j = 0; j < 8; ++j) onef[j] = (bf16)((j < 2 && h == 0) ? 1.f : 0.f);
    const f32x16 zero16 = {};
    float T1[4][16], Dacc = 0.f, Lacc = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) T1[t][i] = 0.f;
    __syncthreads();

    const int ntile = j0 < P ? (P - j0 + J - 1) / J : 0;  // tiles of this workgroup: j0, j0 + J, ..
    const int row = 32 * rh + r;
    bf16x8 nx = {};
    float na = 0.f, ny = 0.f, nw = 1.f;
    auto fetch_in = [&](int k) {  // one unit (32 rows of a tile) ahead
        const int agent = (j0 + k * J) * p.n_sets + set;
        const long ri = (long)agent * TILE + row;
        nx = ((const bf16x8*)p.xf)[2 * ri + h];
        if (NET::critic) na = p.act[ri];
        if (MODE == OUT_TD) ny = p.r[ri];
        if (MODE == HEAD_CRITIC || MODE == HEAD_ACTOR) ny = p.yin[ri];
        if (MODE >= HEAD_CRITIC && p.aw) nw = p.aw[agent];
    };
    if (q < ntile) fetch_in(q);
    const bf16* wrow = wimg + r * LD + 8 * h;  // + 32 nt LD + 16 ks
    for (int k = q; k < ntile; k += NW / 2) {
        const int agent = (j0 + k * J) * p.n_sets + set;
        const long ri = (long)agent * TILE + row;
        const bf16x8 xs = nx, xa = make_xf(na, 0.f, 0.f, 0.f, h);
        const float ty = ny, tw = nw, ta = na;
        if (k + NW / 2 < ntile) fetch_in(k + NW / 2);
        f32x16 acc[4] = {zero16, zero16, zero16, zero16};
        // relu + bf16 of a first-layer tile [feature][row] (row on the lane): 4 groups of 4 consecutive features
        auto pack = [&](const f32x16& p1, unsigned (&pk)[4][2]) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                pk[g][0] = relu_bf16x2(p1[4 * g], p1[4 * g + 1]);
                pk[g][1] = relu_bf16x2(p1[4 * g + 2], p1[4 * g + 3]);
            }
        };
        // Software pipeline over the feature tiles: [first-layer MFMA of tile ft+1] [8 second-layer MFMAs of tile ft]
        // [relu/bf16 of tile ft+1]. The VALU part then runs in the shadow of MFMAs already issued (this wave's and its SIMD
        // partner's); issued where its result is consumed, the first-layer MFMA queues behind the partner's eight and the
        // two waves of a SIMD serialise (measured: 31 of 101 us).
        unsigned pk[4][2];
        pack(mfma(wfs[lane], xs, zero16), pk);
        bf16x8 wfq = wfs[64 + lane], wfq2 = wfq;  // first-layer weights of tile ft + 1, read a tile ahead as well
        // weight fragments of a feature tile (2 k-steps x 4 column tiles), read one tile ahead: a ds_read_b128 issued right in
        // front of the MFMA that consumes it exposes the LDS latency four times per tile
        // (the modes that carry the 64 T1 accumulators have no registers for all 8 fragments of the next tile: they read its
        //  first k-step ahead and the second k-step of the current tile in front of the first k-step's four MFMAs)
        constexpr bool HALF = (MODE == HEAD_CRITIC || MODE == HEAD_ACTOR);
        bf16x8 wcur[2][4], wnxt[HALF ? 1 : 2][4];
        auto read_w1 = [&](int ft, int gg, bf16x8 (&dst)[4]) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (2 * ft + gg < NKS) dst[t] = *(const bf16x8*)(wrow + 32 * t * LD + 16 * (2 * ft + gg));
        };
        auto read_w = [&](int ft, bf16x8 (&dst)[HALF ? 1 : 2][4]) {
            read_w1(ft, 0, dst[0]);
            if (!HALF) read_w1(ft, 1, dst[HALF ? 0 : 1]);
        };
        read_w1(0, 0, wcur[0]);
        if (!HALF) read_w1(0, 1, wcur[1]);
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            f32x16 p1n = zero16;
            if (ft + 1 < NFT) {
                p1n = mfma(wfq, (NET::critic && ft + 1 >= 8) ? xa : xs, zero16);
                read_w(ft + 1, wnxt);
                if (ft + 2 < NFT) wfq2 = wfs[(ft + 2) * 64 + lane];
            }
            if (HALF) read_w1(ft, 1, wcur[1]);
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int ks = 2 * ft + gg;
                if (ks >= NKS) continue;  // (the critic's last tile holds only 16 features)
                // the lane holds features 8g + 4h + j; after the swap h = 0 holds 16gg + [0, 8), h = 1 holds 16gg + [8, 16)
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gg][0], pk[2 * gg + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gg][1], pk[2 * gg + 1][1], false, false);
                uint4 o;
                o.x = s0[0], o.y = s1[0], o.z = s0[1], o.w = s1[1];
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, o);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = mfma(wcur[gg][t], bfrag, acc[t]);
            }
            if (ft + 1 < NFT) {
                pack(p1n, pk);
#pragma unroll
                for (int gg = 0; gg < (HALF ? 1 : 2); ++gg)
#pragma unroll
                    for (int t = 0; t < 4; ++t) wcur[gg][t] = wnxt[gg][t];
                wfq = wfq2;
            }
            // Issue order inside the tile: an MFMA holds the SIMD's issue port for 8 of its 32 cycles, so the next tile's
            // relu/bf16 conversions (24 VALU) and weight reads go INTO the gaps between this tile's 8 second-layer MFMAs -- a
            // wave is in-order: VALU placed behind the MFMAs waits for all of them to be issued (measured: VALU-active 28 % +
            // issue-stalled 44 % of the wave's cycles, the matrix pipe 48 % busy).
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // first-layer MFMA of the next tile
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // the four lane-half swaps
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            // (tiles stay in program order: the scheduler otherwise hoists every tile's first-layer MFMA and weight reads to
            //  the top and spills hundreds of registers)
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma(*(const bf16x8*)(wrow + 32 * t * LD + 16 * NKS), onef, acc[t]);  // + b2'
        float zp = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 c = *(const float4*)(c3s + 32 * t + 8 * g + 4 * h);
                const float cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[t][4 * g + j] = relu(acc[t][4 * g + j]);
                    zp = fmaf(acc[t][4 * g + j], cc[j], zp);
                }
            }
        zp += __shfl_xor(zp, 32);
        const float z = d3 + zp;
        if (MODE == OUT_TANH) {
            const float o = tanhf(z) * p.high;
            if (h == 0) p.out[ri] = o;
        } else if (MODE == OUT_TD) {
            if (h == 0) p.out[ri] = ty + p.gamma * z;
        } else {
            float g3, loss;
            if (MODE == HEAD_CRITIC) {
                const float diff = z - ty;
                g3 = 2.f * diff * p.inv_n * tw, loss = diff * diff;
            } else if (MODE == HEAD_CONST) {
                g3 = -p.inv_n * tw, loss = z;
            } else {
                const float t = tanhf(z);
                g3 = ty * p.high * (1.f - t * t), loss = 0.f;
            }
            bf16* dst = p.dz + ri * H2 + 8 * h;
            f32x16 dct[2] = {zero16, zero16};  // AG: dC^T [action feature][row], row on the lane
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                unsigned pk[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 c = *(const float4*)(c3s + 32 * t + 8 * g + 4 * h);
                    const float cc[4] = {c.x, c.y, c.z, c.w};
                    bf16x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = 4 * g + j;
                        if (MODE != HEAD_CONST) T1[t][i] = fmaf(g3, acc[t][i], T1[t][i]);
                        v[j] = (bf16)(acc[t][i] > 0.f ? g3 * cc[j] : 0.f);
                    }
                    const uint2 u = __builtin_bit_cast(uint2, v);
                    pk[g][0] = u.x, pk[g][1] = u.y;
                }
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {  // 16-byte row-major pieces: the row's two lanes cover 32 contiguous bytes
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gg][0], pk[2 * gg + 1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gg][1], pk[2 * gg + 1][1], false, false);
                    uint4 o;
                    o.x = s0[0], o.y = s1[0], o.z = s0[1], o.w = s1[1];
                    if (AG) {
                        // dZ2' never leaves the registers: these 8 consecutive columns of the lane's row ARE the B operand
                        // (k-step 2t + gg of the 128-column reduction) of dC^T = W2[action rows] . dZ2'^T
                        const bf16x8 bfrag = __builtin_bit_cast(bf16x8, o);
#pragma unroll
                        for (int fl = 0; fl < 2; ++fl)
                            dct[fl] = mfma(*(const bf16x8*)(w2a + (32 * fl + r) * LDA + 16 * (2 * t + gg) + 8 * h), bfrag, dct[fl]);
                    } else {
                        *(uint4*)(dst + 32 * t + 16 * gg) = o;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // one column tile at a time (register pressure)
            }
            if (AG) {
                // dmu[row] = sum_f dC[row][f] * inv_a[f] * (pa[f] > 0) * wa[f], pa = relu(mu wa + ba): BN / ReLU backward of the
                // action layer and its 1-wide input; the lane holds 16 + 16 features of its row, its partner the others
                float sum = 0.f;
#pragma unroll
                for (int fl = 0; fl < 2; ++fl)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float4 c = *(const float4*)ctab[32 * fl + acc_row(i, h)];
                        const float pa = fmaf(ta, c.x, c.y);
                        sum += pa > 0.f ? dct[fl][i] * c.z : 0.f;
                    }
                sum += __shfl_xor(sum, 32);
                if (h == 0) p.dmu[ri] = sum;
            }
            if (h == 0) Dacc += g3, Lacc += loss;
        }
    }
    if (MODE >= HEAD_CRITIC) {
        // one partial per wave: sums over the 32 row lanes of each half (fixed shuffle tree), one writer per half
        if (MODE != HEAD_CONST) {
            float* pt = p.part + ((long)blockIdx.x * 8 + w) * H2;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = T1[t][i];
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o);
                    if (r == 0) pt[32 * t + acc_row(i, h)] = v;
                }
        }
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            Dacc += __shfl_xor(Dacc, o);
            Lacc += __shfl_xor(Lacc, o);
        }
        if (lane == 0) {
            p.part_s[((long)blockIdx.x * 8 + w) * 2] = Dacc;
            p.part_s[((long)blockIdx.x * 8 + w) * 2 + 1] = Lacc;
        }
    }
}

// ---- dw: G[f][n] += sum_rows P1[row][f] dZ2[row][n] over all tiles of the workgroup -------------------------------------
struct DwArgs {
    NetP net;
    int n_agents, n_sets;
    const bf16* xf;  // packed state fragments (pack_x_kernel)
    const float* act;
    const bf16* dz;
    float* partG;  // [grid][KG][128] (row K: the constant-one feature = column sums of dZ2)
};
// Wave w = (fg, wn): feature tiles fg*NFT/2 .. of column quarter wn. The first-layer tile is computed with the batch rows as
// the M index (result: feature on the lane, rows in the registers) and used straight as the A operand of G = P1^T . dZ2
// (k order permuted, tools/probes/mfma_layout.hip); the dZ2 tile goes through LDS (row stride 320 B: conflict-free for
// ds_read_b64_tr_b16, which delivers it reduction-contiguous in exactly that order). One barrier per tile.
template <int S, class NET>
__global__ __launch_bounds__(NT) void dw_kernel(const DwArgs p) {
    constexpr int KG = NET::KG, NGT = NET::NGT, NFW = (NGT + 1) / 2, LDZ = 160;
    __shared__ __attribute__((aligned(16))) bf16 dzimg[2][TILE * LDZ];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5, wn = w & 3, fg = w >> 2;
    const int set = blockIdx.x % p.n_sets, j0 = blockIdx.x / p.n_sets, J = gridDim.x / p.n_sets, P = p.n_agents / p.n_sets;
    const float* th = p.net.th + (long)set * p.net.th_stride;
    bf16x8 wf[NFW];
    f32x16 G[NFW];
    const f32x16 zero16 = {};
#pragma unroll
    for (int i = 0; i < NFW; ++i) wf[i] = layer1_wf<S, NET>(p.net, th, fg * NFW + i, r, h, true), G[i] = zero16;
    const bool act_tiles = NET::critic && fg == 1;  // feature tiles 8, 9 (i = 3, 4) take the action as their input

    const int srow = tid >> 3, sch = tid & 7;  // staging: 64 rows x 8 chunks of 32 bytes
    uint4 d0 = {}, d1 = {};
    auto fetch = [&](int agent) {
        const uint4* src = (const uint4*)(p.dz + ((long)agent * TILE + srow) * H2 + 16 * sch);
        d0 = src[0], d1 = src[1];
    };
    auto stage = [&](int buf) {
        uint4* dst = (uint4*)(dzimg[buf] + srow * LDZ + 16 * sch);
        dst[0] = d0, dst[1] = d1;
    };
    bf16x8 nx0 = {}, nx1 = {};  // the next tile's inputs, requested a tile ahead like its dZ2
    float na0 = 0.f, na1 = 0.f;
    auto fetch_x = [&](int agent) {
        const bf16x8* xa = (const bf16x8*)p.xf + (long)agent * TILE * 2;
        nx0 = xa[2 * r + h], nx1 = xa[2 * (32 + r) + h];
        if (act_tiles) na0 = p.act[(long)agent * TILE + r], na1 = p.act[(long)agent * TILE + 32 + r];
    };
    if (j0 < P) fetch(j0 * p.n_sets + set), fetch_x(j0 * p.n_sets + set), stage(0);
    __syncthreads();
    const int g4 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    int buf = 0;
    for (int pi = j0; pi < P; pi += J, buf ^= 1) {
        const bool more = pi + J < P;
        const bf16x8 xs0 = nx0, xs1 = nx1;
        const bf16x8 xa0 = make_xf(na0, 0.f, 0.f, 0.f, h), xa1 = make_xf(na1, 0.f, 0.f, 0.f, h);
        if (more) fetch((pi + J) * p.n_sets + set), fetch_x((pi + J) * p.n_sets + set);
        bf16x8 bfr[2][2];
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int R0 = 32 * rh + 16 * s + 8 * hf + 4 * (g4 >> 1);
                    const bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(dzimg[buf] + (R0 + q) * LDZ + 32 * wn + 16 * (g4 & 1) + 4 * pp));
#pragma unroll
                    for (int j = 0; j < 4; ++j) bfr[rh][s][4 * hf + j] = t[j];
                }
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            if (fg * NFW + i >= NGT) continue;
            const bool at = act_tiles && i >= 3;
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const bf16x8 xf = rh ? (at ? xa1 : xs1) : (at ? xa0 : xs0);
                const f32x16 p1 = mfma(xf, wf[i], zero16);  // [row][feature]: feature on the lane
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    uint4 xu;
                    xu.x = relu_bf16x2(p1[8 * s], p1[8 * s + 1]), xu.y = relu_bf16x2(p1[8 * s + 2], p1[8 * s + 3]);
                    xu.z = relu_bf16x2(p1[8 * s + 4], p1[8 * s + 5]), xu.w = relu_bf16x2(p1[8 * s + 6], p1[8 * s + 7]);
                    G[i] = mfma(__builtin_bit_cast(bf16x8, xu), bfr[rh][s], G[i]);
                }
            }
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
        if (fg * NFW + i >= NGT) continue;
        float* dst = p.partG + ((long)blockIdx.x * KG + 32 * (fg * NFW + i)) * H2 + 32 * wn + r;
#pragma unroll
        for (int k = 0; k < 16; ++k) dst[(long)acc_row(k, h) * H2] = G[i][k];
    }
}

// ---- dx: dC = dZ2 . W2^T, BN/ReLU backward of the first layer and its parameter sums; or the action gradient -----------
struct DxArgs {
    NetP net;
    int n_agents, n_sets;
    const bf16* xf;  // dx_kernel: packed state fragments (pack_x_kernel)
    const float* x;  // dxa_kernel: the per-row action input [n_agents][64]
    const bf16* dz;
    float* partU;    // [grid][2 rh][2 h][KP][2]   sum dC, sum dC * p1 per feature
    float* partV;    // [grid][2 rh][KP][16]       sum_rows (dC * mask) * [x_hi | x_lo | 1] per feature
};
// Wave w = (rh, fg): row half rh of feature tiles fg + 4 i (the 8 state tiles; the critic's action tiles: dxa_kernel).
// Resident: the raw W2 rows of its tiles as B fragments (reduction over the 128 columns). dC comes out [row][feature] (feature
// on the lane), like the recomputed first layer: the per-feature sums over rows are per-lane sums over the registers, and
// the masked gradient tile is the A operand of V = (dC*mask)^T . [x_hi | x_lo | 1], which yields dW1 and db1 (k order
// permuted as in dw_kernel; the [k][row] image of the inputs is the wave's own LDS area).
template <int S, class NET>
__global__ __launch_bounds__(NT) void dx_kernel(const DxArgs p) {
    constexpr int KP = NET::KP, NF = 2, LDZ = 136;  // 272-byte rows: conflict-free b128 row reads
    __shared__ __attribute__((aligned(16))) bf16 dzimg[2][TILE * LDZ];
    __shared__ __attribute__((aligned(16))) bf16 xt[8][2][32 * 32];  // per wave, per buffer: [k column][row of its half]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5, rh = w & 1, fg = w >> 1;
    const int set = blockIdx.x % p.n_sets, j0 = blockIdx.x / p.n_sets, J = gridDim.x / p.n_sets, P = p.n_agents / p.n_sets;
    const float* th = p.net.th + (long)set * p.net.th_stride;
    const f32x16 zero16 = {};
    bf16x8 w2[NF][8], wf[NF];
    f32x16 V[NF];
    float U0[NF], U1[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int ft = fg + 4 * i;
        const bf16* src = p.net.W2R + ((long)set * KP + 32 * ft + r) * H2 + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) w2[i][s] = *(const bf16x8*)(src + 16 * s);
        wf[i] = layer1_wf<S, NET>(p.net, th, ft, r, h);
        V[i] = zero16, U0[i] = 0.f, U1[i] = 0.f;
    }
    for (int i = lane; i < 2 * 32 * 32; i += 64) {  // columns 9.. stay zero, column 8 is the ones column (bias)
        const int k = (i >> 5) & 31;
        xt[w][0][i] = (bf16)(k == 8 ? 1.f : 0.f);
    }
    const int srow = tid >> 3, sch = tid & 7;
    uint4 d0 = {}, d1 = {};
    auto fetch = [&](int agent) {
        const uint4* src = (const uint4*)(p.dz + ((long)agent * TILE + srow) * H2 + 16 * sch);
        d0 = src[0], d1 = src[1];
    };
    auto stage = [&](int buf) {
        uint4* dst = (uint4*)(dzimg[buf] + srow * LDZ + 16 * sch);
        dst[0] = d0, dst[1] = d1;
    };
    bf16x8 xfn = {};  // the row's input fragment [x_hi | x_lo] (h = 0) / [x_hi | 1 1 0 0] (h = 1)
    auto fetch_x = [&](int agent) { xfn = ((const bf16x8*)p.xf)[((long)agent * TILE + 32 * rh + r) * 2 + h]; };
    auto stage_x = [&](int buf) {  // [k][row] image of [x_hi | x_lo] of the wave's 32 rows: the h = 0 fragment, transposed
        if (h == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) xt[w][buf][k * 32 + r] = xfn[k];
        }
    };
    __syncthreads();  // the zero / ones fill above before the first stage_x
    if (j0 < P) fetch(j0 * p.n_sets + set), fetch_x(j0 * p.n_sets + set), stage(0), stage_x(0);
    __syncthreads();
    int buf = 0;
    for (int pi = j0; pi < P; pi += J, buf ^= 1) {
        const bool more = pi + J < P;
        const bf16x8 xf = xfn;
        if (more) fetch((pi + J) * p.n_sets + set), fetch_x((pi + J) * p.n_sets + set);
        bf16x8 dzf[8];
        const bf16* arow = dzimg[buf] + (32 * rh + r) * LDZ + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) dzf[s] = *(const bf16x8*)(arow + 16 * s);
        bf16x8 xb[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x4 lo = *(const bf16x4*)(&xt[w][buf][r * 32 + 16 * s + 4 * h]);
            const bf16x4 hi = *(const bf16x4*)(&xt[w][buf][r * 32 + 16 * s + 8 + 4 * h]);
#pragma unroll
            for (int j = 0; j < 4; ++j) xb[s][j] = lo[j], xb[s][4 + j] = hi[j];
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            f32x16 dc = zero16;  // [row][feature]: feature on the lane
#pragma unroll
            for (int s = 0; s < 8; ++s) dc = mfma(dzf[s], w2[i][s], dc);
            const f32x16 p1 = mfma(xf, wf[i], zero16);
            float u0 = 0.f, u1 = 0.f;
            bf16x8 va[2];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float pr = relu(p1[k]);
                u0 += dc[k];
                u1 = fmaf(dc[k], pr, u1);
                va[k >> 3][k & 7] = (bf16)(pr > 0.f ? dc[k] : 0.f);
            }
            U0[i] += u0, U1[i] += u1;
            V[i] = mfma(va[0], xb[0], V[i]);
            V[i] = mfma(va[1], xb[1], V[i]);
        }
        if (more) stage(buf ^ 1), stage_x(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int ft = fg + 4 * i;
        float* pu = p.partU + ((((long)blockIdx.x * 2 + rh) * 2 + h) * KP + 32 * ft + r) * 2;
        pu[0] = U0[i], pu[1] = U1[i];
        if (r < 16) {
            float* pv = p.partV + (((long)blockIdx.x * 2 + rh) * KP + 32 * ft) * 16 + r;
#pragma unroll
            for (int k = 0; k < 16; ++k) pv[(long)acc_row(k, h) * 16] = V[i][k];
        }
    }
}

// ---- dxa: the critic's ACTION feature tiles (48 features = tiles 8, 9) of dC = dZ2 . W2^T and their parameter sums --------
// A sixth of dx_kernel's work per tile, bound by reading dZ2 (16 KB per agent) once more. Every wave is on its own: wave
// w = (rh, ft, par) takes row half rh of feature tile 8 + ft of every second tile (parity par) of the workgroup and reads its
// 32 dZ2 rows straight from global memory as A fragments, one tile ahead; the two parities are combined once, at the end.
// (The action gradient of pass 7 needs the same product on dZ2': head_kernel<HEAD_CONST> takes it from registers.)
template <int S>
__global__ __launch_bounds__(NT) void dxa_kernel(const DxArgs p) {
    typedef Critic NET;
    constexpr int KP = NET::KP;
    __shared__ __attribute__((aligned(16))) bf16 xt[8][2][32 * 32];  // per wave, per buffer: [k column][row of its half]
    __shared__ float comb[4][64][2 + 16];  // parity-1 waves' U0, U1, V -> their parity-0 partners
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5, rh = w & 1, ftl = (w >> 1) & 1, par = w >> 2;
    const int set = blockIdx.x % p.n_sets, j0 = blockIdx.x / p.n_sets, J = gridDim.x / p.n_sets, P = p.n_agents / p.n_sets;
    const float* th = p.net.th + (long)set * p.net.th_stride;
    const f32x16 zero16 = {};
    const int ft = 8 + ftl;
    bf16x8 w2[8];
    {
        const bf16* src = p.net.W2R + ((long)set * KP + 32 * ft + r) * H2 + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) w2[s] = *(const bf16x8*)(src + 16 * s);
    }
    const bf16x8 wf = layer1_wf<S, NET>(p.net, th, ft, r, h);
    f32x16 V = zero16;
    float U0 = 0.f, U1 = 0.f;
    for (int i = lane; i < 2 * 32 * 32; i += 64) {  // columns 9.. stay zero, column 8 is the ones column (bias)
        const int k = (i >> 5) & 31;
        xt[w][0][i] = (bf16)(k == 8 ? 1.f : 0.f);
    }
    __syncthreads();
    const int ntile = j0 < P ? (P - j0 + J - 1) / J : 0;
    bf16x8 dzn[8];
    float an = 0.f;
    auto fetch = [&](int k) {
        const long ri = (long)((j0 + k * J) * p.n_sets + set) * TILE + 32 * rh + r;
        const bf16* src = p.dz + ri * H2 + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) dzn[s] = *(const bf16x8*)(src + 16 * s);
        an = p.x[ri];
    };
    if (par < ntile) fetch(par);
    int buf = 0;
    for (int k = par; k < ntile; k += 2, buf ^= 1) {
        bf16x8 dzf[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) dzf[s] = dzn[s];
        const bf16x8 xf = make_xf(an, 0.f, 0.f, 0.f, h);
        if (k + 2 < ntile) fetch(k + 2);
        if (h == 0) {  // [k][row] image of [a_hi 0 0 0 | a_lo 0 0 0] of the wave's 32 rows: the h = 0 fragment, transposed
            xt[w][buf][r] = xf[0];
            xt[w][buf][4 * 32 + r] = xf[4];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bf16x8 xb[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x4 lo = *(const bf16x4*)(&xt[w][buf][r * 32 + 16 * s + 4 * h]);
            const bf16x4 hi = *(const bf16x4*)(&xt[w][buf][r * 32 + 16 * s + 8 + 4 * h]);
#pragma unroll
            for (int j = 0; j < 4; ++j) xb[s][j] = lo[j], xb[s][4 + j] = hi[j];
        }
        f32x16 dc = zero16;  // [row][feature]: feature on the lane
#pragma unroll
        for (int s = 0; s < 8; ++s) dc = mfma(dzf[s], w2[s], dc);
        const f32x16 p1 = mfma(xf, wf, zero16);
        float u0 = 0.f, u1 = 0.f;
        bf16x8 va[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float pr = relu(p1[i]);
            u0 += dc[i];
            u1 = fmaf(dc[i], pr, u1);
            va[i >> 3][i & 7] = (bf16)(pr > 0.f ? dc[i] : 0.f);
        }
        U0 += u0, U1 += u1;
        V = mfma(va[0], xb[0], V);
        V = mfma(va[1], xb[1], V);
    }
    const int pw = w & 3;  // (rh, ftl)
    if (par == 1) {
        comb[pw][lane][0] = U0, comb[pw][lane][1] = U1;
#pragma unroll
        for (int i = 0; i < 16; ++i) comb[pw][lane][2 + i] = V[i];
    }
    __syncthreads();
    if (par == 0) {
        U0 += comb[pw][lane][0], U1 += comb[pw][lane][1];
        float* pu = p.partU + ((((long)blockIdx.x * 2 + rh) * 2 + h) * KP + 32 * ft + r) * 2;
        pu[0] = U0, pu[1] = U1;
        if (r < 16) {
            float* pv = p.partV + (((long)blockIdx.x * 2 + rh) * KP + 32 * ft) * 16 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) pv[(long)acc_row(i, h) * 16] = V[i] + comb[pw][lane][2 + i];
        }
    }
}

// ---- finalize: sum the workgroups' partials of a set (fixed order), apply the BN folds, write the gradient slab -------
// One item per 32 lanes: output column n (items 0..127: output layer, BN2, b2) or first-layer feature f (items 128..):
// lane jl sums the partials of workgroups jl, jl + 32, .. of the set, then a fixed shuffle tree combines the 32 lanes.
__global__ __launch_bounds__(512) void finalize_small_kernel(const FinArgs a) {
    const int set = blockIdx.x, net = a.net_lo + blockIdx.y, tid = threadIdx.x, jl = tid & 31;
    const int item = a.item_base + blockIdx.z * 16 + (tid >> 5);
    const bool critic = net;
    const avd_mlp_layout& L = a.L;
    const float* th = a.theta + (long)set * L.theta_size + (critic ? L.actor_size : 0);
    const float* st = a.stats + (long)set * L.stats_size;
    float* g = a.grads + (long)set * L.theta_size + (critic ? L.actor_size : 0);
    const int K = critic ? Critic::K : Actor::K, KP = critic ? Critic::KP : Actor::KP;
    auto allsum = [&](float v) {
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o);
        return v;
    };
    if (item < H2) {
        const int n = item;
        const int og2 = critic ? L.cg3 : L.ag2, obe2 = critic ? L.cbe3 : L.abe2, omm2 = critic ? L.cmm3 : L.amm2, omv2 = critic ? L.cmv3 : L.amv2;
        const int oW3 = critic ? L.cW3 : L.aW3, ob3 = critic ? L.cb3 : L.ab3, ob2 = critic ? L.cb2 : L.ab2;
        const int KG = critic ? Critic::KG : Actor::KG;
        float T1 = 0.f, S2 = 0.f, D = 0.f, Lc = 0.f, La = 0.f;
        const bool t1g = a.t1_from_g[net] != 0;
        for (int j = jl; j < a.J; j += 32) {
            const long wg = (long)j * a.n_sets + set;
            for (int w = 0; w < 8; ++w) {
                if (!t1g) T1 += a.partH[net][(wg * 8 + w) * H2 + n];
                D += a.partHs[net][(wg * 8 + w) * 2];
                if (critic && n == 0) Lc += a.partHs[1][(wg * 8 + w) * 2 + 1], La += a.partLa[(wg * 8 + w) * 2 + 1];
            }
            S2 += a.partG[net][(wg * KG + K) * H2 + n];  // dw_kernel's constant-one feature: sum over rows of dZ2 = db2
        }
        T1 = allsum(T1), S2 = allsum(S2), D = allsum(D);
        if (t1g && jl == 0) a.s2raw[((long)net * a.n_sets + set) * H2 + n] = S2;  // (before c3: sum_rows g3 mask[n])
        if (a.c3[net]) S2 *= a.c3[net][(long)set * VEC + H2 + n];  // fsplit: the column factor c3[n] of dZ2 is applied here
        if (critic && n == 0) Lc = allsum(Lc), La = allsum(La);
        if (a.bad && *a.bad) T1 = S2 = D = Lc = La = __uint_as_float(0x7fc00000u);  // non-finite input: NaN out, like the f32 engines
        if (jl == 0) {
            const float rs2 = 1.0f / sqrtf(st[omv2 + n] + BN_EPS), inv2 = rs2 * th[og2 + n], mm2 = st[omm2 + n];
            const float sh2 = th[obe2 + n] - mm2 * inv2, w3 = th[oW3 + n];
            if (!t1g) {  // (else: finalize_t1_kernel, once the weight-gradient partials have been summed)
                g[oW3 + n] = inv2 * T1 + sh2 * D;
                g[og2 + n] = w3 * rs2 * (T1 - mm2 * D);
            }
            g[obe2 + n] = w3 * D;
            g[ob2 + n] = S2;
            if (n == 0) g[ob3] = D;
            if (critic && n == 0 && a.losses) a.losses[2 * set] = Lc * a.inv_n, a.losses[2 * set + 1] = -La * a.inv_n;
        }
        return;
    }
    const int f = item - H2;
    if (f >= K) return;
    float U0 = 0.f, U1 = 0.f, V[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) V[k] = 0.f;
    const bool have_u = a.partU[net] != nullptr;
    for (int j = jl; j < a.J; j += 32) {
        const long wg = (long)j * a.n_sets + set;
        for (int rh = 0; rh < a.nrh; ++rh) {
            if (have_u)
                for (int h = 0; h < 2; ++h) {
                    const float* pu = a.partU[net] + ((((wg * a.nrh + rh) * 2 + h) * KP) + f) * 2;
                    U0 += pu[0], U1 += pu[1];
                }
            const float* pv = a.partV[net] + ((wg * a.nrh + rh) * KP + f) * 16;
#pragma unroll
            for (int k = 0; k < 9; ++k) V[k] += pv[k];
        }
    }
    U0 = allsum(U0), U1 = allsum(U1);
#pragma unroll
    for (int k = 0; k < 9; ++k) V[k] = allsum(V[k]);
    if (!have_u) {
        // fsplit.hip (r04): the two unmasked per-feature sums are linear images of sums that exist anyway --
        //   U0 = sum_rows dC[row][f]          = sum_n W2[f][n] db2[n]                    (db2: written by the launch before this one)
        //   U1 = sum_rows dC[row][f] relu(z1) = sum_k W1[k][f] V[k] + b1[f] V[8]         (relu(z1) = mask (x . W1 + b1); V = sum (dC mask) [x | 1])
        // so dx_kernel / dxa_kernel spend nothing on them. 32 lanes share the 128-term dot, fixed shuffle tree.
        const int oW2 = critic ? L.cW2 : L.aW2, ob2 = critic ? L.cb2 : L.ab2;
        float u0 = 0.f;
        for (int n = jl; n < H2; n += 32) u0 = fmaf(th[oW2 + (long)f * H2 + n], g[ob2 + n], u0);
        U0 = allsum(u0);
        int oW, ob, ff = f, ld = H1, sin = a.S;
        if (!critic) oW = L.aW1, ob = L.ab1;
        else if (f < H1) oW = L.cWs, ob = L.cbs;
        else oW = L.cWa, ob = L.cba, ff = f - H1, ld = HA, sin = 1;
        U1 = th[ob + ff] * V[8];
        for (int k = 0; k < sin; ++k) U1 = fmaf(th[oW + k * ld + ff], V[k] + V[4 + k], U1);
    }
    if (a.bad && *a.bad) U0 = U1 = V[0] = V[4] = V[8] = __uint_as_float(0x7fc00000u);
    if (jl == 0) {
        int og, obe, omm, omv, oW, ob, ff = f, ld = H1, sin = a.S;
        if (!critic) og = L.ag1, obe = L.abe1, omm = L.amm1, omv = L.amv1, oW = L.aW1, ob = L.ab1;
        else if (f < H1) og = L.cgs, obe = L.cbes, omm = L.cmms, omv = L.cmvs, oW = L.cWs, ob = L.cbs;
        else og = L.cga, obe = L.cbea, omm = L.cmma, omv = L.cmva, oW = L.cWa, ob = L.cba, ff = f - H1, ld = HA, sin = 1;
        const float rs1 = 1.0f / sqrtf(st[omv + ff] + BN_EPS), inv1 = rs1 * th[og + ff], mm1 = st[omm + ff];
        g[obe + ff] = U0;
        g[og + ff] = rs1 * (U1 - mm1 * U0);
        g[ob + ff] = inv1 * V[8];
        for (int k = 0; k < sin; ++k) g[oW + k * ld + ff] = inv1 * (V[k] + V[4 + k]);
    }
}
// W2 gradients: dW2[f][n] = inv1[f] * G[f][n] + sh1[f] * db2[n] (db2 read back from the slab finalize_small wrote)
__global__ __launch_bounds__(128) void finalize_w2_kernel(const FinArgs a) {
    const int f = blockIdx.x, set = blockIdx.y, net = a.net_lo + blockIdx.z, n = threadIdx.x;
    const bool critic = net;
    const int K = critic ? Critic::K : Actor::K, KG = critic ? Critic::KG : Actor::KG;
    if (f >= K) return;
    const avd_mlp_layout& L = a.L;
    const float* th = a.theta + (long)set * L.theta_size + (critic ? L.actor_size : 0);
    const float* st = a.stats + (long)set * L.stats_size;
    float* g = a.grads + (long)set * L.theta_size + (critic ? L.actor_size : 0);
    int og, obe, omm, omv, ff = f;
    if (!critic) og = L.ag1, obe = L.abe1, omm = L.amm1, omv = L.amv1;
    else if (f < H1) og = L.cgs, obe = L.cbes, omm = L.cmms, omv = L.cmvs;
    else og = L.cga, obe = L.cbea, omm = L.cmma, omv = L.cmva, ff = f - H1;
    const float inv1 = (1.0f / sqrtf(st[omv + ff] + BN_EPS)) * th[og + ff], sh1 = th[obe + ff] - st[omm + ff] * inv1;
    // (the J partials are added in index order -- the bits of the result are a function of the plan -- but their loads are
    //  independent: eight in flight instead of a chain of J dependent round trips to L2 / HBM: 27 -> ~12 us per launch)
    float G = 0.f;
    const float* pg = a.partG[net] + ((long)set * KG + f) * H2 + n;
    const long stride = (long)a.n_sets * KG * H2;
    int j = 0;
    for (; j + 8 <= a.J; j += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pg[(j + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) G += v[u];
    }
    for (; j < a.J; ++j) G += pg[j * stride];
    const int oW2 = critic ? L.cW2 : L.aW2, ob2 = critic ? L.cb2 : L.ab2;
    if (a.t1_from_g[net]) a.t1p[(((long)net * a.n_sets + set) * Critic::K + f) * H2 + n] = inv1 * th[oW2 + (long)f * H2 + n] * G;  // W2'[f][n] G[f][n]
    if (a.c3[net]) G *= a.c3[net][(long)set * VEC + H2 + n];
    if (a.bad && *a.bad) G = __uint_as_float(0x7fc00000u);
    g[oW2 + (long)f * H2 + n] = inv1 * G + sh1 * g[ob2 + n];
}
// T1-dependent outputs of a net whose T1 comes from the weight-gradient partials (FinArgs::t1_from_g): one block per set, one
// thread per output column, the K products in index order with eight loads in flight
__global__ __launch_bounds__(128) void finalize_t1_kernel(const FinArgs a) {
    const int set = blockIdx.x, net = a.net_lo + blockIdx.y, n = threadIdx.x;
    if (!a.t1_from_g[net]) return;
    const bool critic = net;
    const avd_mlp_layout& L = a.L;
    const float* th = a.theta + (long)set * L.theta_size + (critic ? L.actor_size : 0);
    const float* st = a.stats + (long)set * L.stats_size;
    float* g = a.grads + (long)set * L.theta_size + (critic ? L.actor_size : 0);
    const int K = critic ? Critic::K : Actor::K;
    const int og2 = critic ? L.cg3 : L.ag2, obe2 = critic ? L.cbe3 : L.abe2, omm2 = critic ? L.cmm3 : L.amm2, omv2 = critic ? L.cmv3 : L.amv2;
    const int oW3 = critic ? L.cW3 : L.aW3, ob3 = critic ? L.cb3 : L.ab3;
    const float* tp = a.t1p + ((long)net * a.n_sets + set) * Critic::K * H2 + n;
    float T1 = 0.f;
    int f = 0;
    for (; f + 8 <= K; f += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = tp[(long)(f + u) * H2];
#pragma unroll
        for (int u = 0; u < 8; ++u) T1 += v[u];
    }
    for (; f < K; ++f) T1 += tp[(long)f * H2];
    T1 = fmaf(a.c3[net][(long)set * VEC + n], a.s2raw[((long)net * a.n_sets + set) * H2 + n], T1);  // + b2'[n] S2[n]  (vec[n] = b2')
    const float D = g[ob3];
    if (a.bad && *a.bad) T1 = __uint_as_float(0x7fc00000u);
    const float rs2 = 1.0f / sqrtf(st[omv2 + n] + BN_EPS), inv2 = rs2 * th[og2 + n], mm2 = st[omm2 + n];
    const float sh2 = th[obe2 + n] - mm2 * inv2, w3 = th[oW3 + n];
    g[oW3 + n] = inv2 * T1 + sh2 * D;
    g[og2 + n] = w3 * rs2 * (T1 - mm2 * D);
}

// nets [net_lo, net_lo + n_nets): both (default), or one block of the slab (fsplit.hip's two phases). The critic launch also writes
// the two losses (the actor loss is the mean of q(s, mu): partLa).
void launch_finalize(const FinArgs& fa0, hipStream_t st, int net_lo, int n_nets) {
    FinArgs fa = fa0;
    fa.net_lo = net_lo, fa.item_base = 0;
    const int K = (net_lo + n_nets > 1) ? Critic::K : Actor::K;
    if (fa.partU[net_lo]) {
        hipLaunchKernelGGL(finalize_small_kernel, dim3(fa.n_sets, n_nets, (H2 + K + 15) / 16), dim3(512), 0, st, fa);
    } else {  // the feature items read the db2 the column items write: two launches
        hipLaunchKernelGGL(finalize_small_kernel, dim3(fa.n_sets, n_nets, H2 / 16), dim3(512), 0, st, fa);
        fa.item_base = H2;
        hipLaunchKernelGGL(finalize_small_kernel, dim3(fa.n_sets, n_nets, (K + 15) / 16), dim3(512), 0, st, fa);
    }
    hipLaunchKernelGGL(finalize_w2_kernel, dim3(K, fa.n_sets, n_nets), dim3(H2), 0, st, fa);
    bool any = false;
    for (int net = net_lo; net < net_lo + n_nets; ++net) any = any || fa.t1_from_g[net];
    if (any) hipLaunchKernelGGL(finalize_t1_kernel, dim3(fa.n_sets, n_nets), dim3(H2), 0, st, fa);
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct Plan {
    int grid, J;
    size_t W2T[4], W2R[2], vec[4], a2, y, mu, dmu, dz, xfs, xfs2, partH[2], partHs[3], partU[2], partV[2], partG[2], bad, total;
};
// CUs of the CURRENT device (cached per device ordinal): the plan -- workgroups per set, workspace layout, the grouping of the
// partial sums and therefore the bits of the result -- is a function of (CU count, n_sets, n_agents); results are
// deterministic per device model, not across parts with different CU counts.
int cu_count() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 256;
    if (dev < 64 && cache[dev]) return cache[dev];
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    const int n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (dev < 64) cache[dev] = n;
    return n;
}
static Plan make_plan(int n_agents, int n_sets) {
    Plan pl;
    const int P = n_agents / n_sets;
    int J = cu_count() / n_sets;  // one 512-thread workgroup per CU, every workgroup bound to one set
    if (J < 1) J = 1;
    if (J > P) J = P;
    pl.J = J, pl.grid = J * n_sets;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += (bytes + 255) / 256 * 256;
        return at;
    };
    for (int i = 0; i < 4; ++i) {
        const int KP = (i & 1) ? Critic::KP : Actor::KP, KW = (i & 1) ? Critic::KW : Actor::KW;
        pl.W2T[i] = take(sizeof(bf16) * (size_t)n_sets * H2 * KW);
        pl.vec[i] = take(sizeof(float) * (size_t)n_sets * VEC);
        if (i < 2) pl.W2R[i] = take(sizeof(bf16) * (size_t)n_sets * KP * H2);
    }
    const size_t rows = (size_t)n_agents * TILE;
    pl.a2 = take(4 * rows), pl.y = take(4 * rows), pl.mu = take(4 * rows), pl.dmu = take(4 * rows);
    pl.dz = take(sizeof(bf16) * rows * H2);
    pl.xfs = take(sizeof(bf16) * rows * 16), pl.xfs2 = take(sizeof(bf16) * rows * 16);
    for (int i = 0; i < 2; ++i) {
        const int KP = i ? Critic::KP : Actor::KP, KG = i ? Critic::KG : Actor::KG;
        pl.partH[i] = take(4 * (size_t)pl.grid * 8 * H2);
        pl.partU[i] = take(4 * (size_t)pl.grid * 2 * 2 * KP * 2);
        pl.partV[i] = take(4 * (size_t)pl.grid * 2 * KP * 16);
        pl.partG[i] = take(4 * (size_t)pl.grid * KG * H2);
    }
    for (int i = 0; i < 3; ++i) pl.partHs[i] = take(4 * (size_t)pl.grid * 8 * 2);
    pl.bad = take(sizeof(int));
    pl.total = o;
    return pl;
}
static int check_fset(const avd_mlp_layout* L, int n_agents, int n_sets, const char* who) {
    AVD_REQUIRE(L, "%s: null layout", who);
    if (L->H1 != H1 || L->H2 != H2 || L->Ha != HA || L->A != 1 || (L->S != 3 && L->S != 4) || L->B != TILE) {
        set_error("%s: serves the reference widths only (layer1 256, layer2 128, action layer 48, A = 1, S in {3, 4}, B = 64); "
                  "got H1=%d H2=%d Ha=%d A=%d S=%d B=%d (avd_learn_shared_bf16 takes other widths)",
                  who, L->H1, L->H2, L->Ha, L->A, L->S, L->B);
        return AVD_E_UNSUPPORTED;
    }
    AVD_REQUIRE(n_sets > 0 && n_sets <= 64 && n_agents > 0 && n_agents % n_sets == 0, "%s: n_agents=%d n_sets=%d", who, n_agents,
                n_sets);
    return AVD_OK;
}

template <int S>
static int run(const avd_mlp_layout& L, int n_agents, int n_sets, const float* theta, const float* stats, const float* theta_t,
               const float* stats_t, const float* s, const float* a, const float* r, const float* s2, const float* aw, float gamma,
               float high, float* grads, float* losses, unsigned char* ws, const Plan& pl, hipStream_t st) {
    PrepArgs pa;
    pa.L = L, pa.theta = theta, pa.stats = stats, pa.theta_t = theta_t, pa.stats_t = stats_t;
    NetP net[4];
    for (int i = 0; i < 4; ++i) {
        const bool critic = i & 1, target = i >= 2;
        pa.W2T[i] = (bf16*)(ws + pl.W2T[i]), pa.vec[i] = (float*)(ws + pl.vec[i]);
        pa.W2R[i] = i < 2 ? (bf16*)(ws + pl.W2R[i]) : nullptr;
        NetP& n = net[i];
        n.th = (target ? theta_t : theta) + (critic ? L.actor_size : 0), n.st = target ? stats_t : stats;
        n.th_stride = L.theta_size, n.st_stride = L.stats_size;
        n.oW1 = critic ? L.cWs : L.aW1, n.ob1 = critic ? L.cbs : L.ab1;
        n.oWa = critic ? L.cWa : 0, n.oba = critic ? L.cba : 0, n.oga = critic ? L.cga : 0, n.omva = critic ? L.cmva : 0;
        n.W2T = pa.W2T[i], n.W2R = pa.W2R[i], n.vec = pa.vec[i];
    }
    pa.bad = (int*)(ws + pl.bad);
    (void)hipMemsetAsync(ws + pl.bad, 0, sizeof(int), st);
    hipLaunchKernelGGL(prep_kernel, dim3(H2, 4, n_sets), dim3(320), 0, st, pa);
    const long nrows = (long)n_agents * TILE;
    bf16 *xfs = (bf16*)(ws + pl.xfs), *xfs2 = (bf16*)(ws + pl.xfs2);
    int* bad = (int*)(ws + pl.bad);
    hipLaunchKernelGGL(pack_x_kernel<S>, dim3((unsigned)((2 * nrows + 255) / 256)), dim3(256), 0, st, s, a, nrows, xfs, bad);
    hipLaunchKernelGGL(pack_x_kernel<S>, dim3((unsigned)((2 * nrows + 255) / 256)), dim3(256), 0, st, s2, r, nrows, xfs2, bad);
    const int P = n_agents / n_sets;
    const float inv_n = 1.0f / ((float)P * TILE);
    float *a2 = (float*)(ws + pl.a2), *y = (float*)(ws + pl.y), *mu = (float*)(ws + pl.mu), *dmu = (float*)(ws + pl.dmu);
    bf16* dz = (bf16*)(ws + pl.dz);
    auto F = [&](size_t off) { return (float*)(ws + off); };
    const dim3 grid(pl.grid), block(NT);
    HeadArgs h;
    h.n_agents = n_agents, h.n_sets = n_sets, h.gamma = gamma, h.high = high, h.inv_n = inv_n, h.aw = aw, h.dz = dz, h.dmu = dmu;
    auto head = [&](auto kern, int mode, int ni, const bf16* x, const float* act, const float* rr, const float* yin, float* out,
                    float* part, float* part_s) {
        h.net = net[ni], h.xf = x, h.act = act, h.r = rr, h.yin = yin, h.out = out, h.part = part, h.part_s = part_s;
        hipLaunchKernelGGL(kern, grid, dim3(64 * head_waves(mode)), 0, st, h);
    };
    DwArgs dw;
    dw.n_agents = n_agents, dw.n_sets = n_sets, dw.dz = dz;
    DxArgs dx;
    dx.n_agents = n_agents, dx.n_sets = n_sets, dx.dz = dz;
    // 1-2: targets
    head(head_kernel<S, Actor, OUT_TANH>, OUT_TANH, 2, xfs2, nullptr, nullptr, nullptr, a2, nullptr, nullptr);
    head(head_kernel<S, Critic, OUT_TD>, OUT_TD, 3, xfs2, a2, r, nullptr, y, nullptr, nullptr);
    // 3: mu (independent of the critic passes)
    head(head_kernel<S, Actor, OUT_TANH>, OUT_TANH, 0, xfs, nullptr, nullptr, nullptr, mu, nullptr, nullptr);
    // 4-6: critic loss and gradients
    head(head_kernel<S, Critic, HEAD_CRITIC>, HEAD_CRITIC, 1, xfs, a, nullptr, y, nullptr, F(pl.partH[1]), F(pl.partHs[1]));
    dw.net = net[1], dw.xf = xfs, dw.act = a, dw.partG = F(pl.partG[1]);
    hipLaunchKernelGGL((dw_kernel<S, Critic>), grid, block, 0, st, dw);
    dx.net = net[1], dx.partU = F(pl.partU[1]), dx.partV = F(pl.partV[1]);
    dx.xf = xfs, dx.x = nullptr;
    hipLaunchKernelGGL((dx_kernel<S, Critic>), grid, block, 0, st, dx);
    dx.x = a;
    hipLaunchKernelGGL((dxa_kernel<S>), grid, block, 0, st, dx);
    // 7-8: actor loss through the critic, gradient w.r.t. the action
    head(head_kernel<S, Critic, HEAD_CONST>, HEAD_CONST, 1, xfs, mu, nullptr, nullptr, nullptr, nullptr, F(pl.partHs[2]));
    // 9-11: actor gradients
    head(head_kernel<S, Actor, HEAD_ACTOR>, HEAD_ACTOR, 0, xfs, nullptr, nullptr, dmu, nullptr, F(pl.partH[0]), F(pl.partHs[0]));
    dw.net = net[0], dw.xf = xfs, dw.act = nullptr, dw.partG = F(pl.partG[0]);
    hipLaunchKernelGGL((dw_kernel<S, Actor>), grid, block, 0, st, dw);
    dx.net = net[0], dx.partU = F(pl.partU[0]), dx.partV = F(pl.partV[0]), dx.x = nullptr;
    hipLaunchKernelGGL((dx_kernel<S, Actor>), grid, block, 0, st, dx);
    // 12: finalize
    FinArgs fa;
    fa.L = L, fa.n_sets = n_sets, fa.J = pl.J, fa.S = S, fa.theta = theta, fa.stats = stats, fa.grads = grads, fa.losses = losses;
    fa.inv_n = inv_n, fa.partLa = F(pl.partHs[2]);
    for (int i = 0; i < 2; ++i)
        fa.partH[i] = F(pl.partH[i]), fa.partHs[i] = F(pl.partHs[i]), fa.partU[i] = F(pl.partU[i]), fa.partV[i] = F(pl.partV[i]),
        fa.partG[i] = F(pl.partG[i]);
    fa.nrh = 2, fa.c3[0] = fa.c3[1] = nullptr, fa.bad = (const int*)(ws + pl.bad);
    fa.t1_from_g[0] = fa.t1_from_g[1] = 0, fa.t1p = nullptr, fa.s2raw = nullptr;
    launch_finalize(fa, st);
    return check_launch("avd_learn_set_fused_bf16");
}

}  // namespace fset
}  // namespace avd

using namespace avd;

extern "C" int avd_learn_set_fused_workspace(const avd_mlp_layout* lay, int n_agents, int n_sets, size_t* bytes) {
    int rc = fset::check_fset(lay, n_agents, n_sets, "avd_learn_set_fused_workspace");
    if (rc) return rc;
    AVD_REQUIRE(bytes, "avd_learn_set_fused_workspace: null pointer");
    *bytes = fset::make_plan(n_agents, n_sets).total;
    return AVD_OK;
}

extern "C" int avd_learn_set_fused_bf16(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                                        const float* theta_t, const float* stats_t, const float* s, const float* a, const float* r,
                                        const float* s2, const float* agent_weight, float gamma, float high, float* grads,
                                        float* losses, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = fset::check_fset(lay, n_agents, n_sets, "avd_learn_set_fused_bf16");
    if (rc) return rc;
    AVD_REQUIRE(theta && stats && theta_t && stats_t && s && a && r && s2 && grads && workspace,
                "avd_learn_set_fused_bf16: null pointer");
    const fset::Plan pl = fset::make_plan(n_agents, n_sets);
    AVD_REQUIRE(workspace_bytes >= pl.total, "avd_learn_set_fused_bf16: workspace %zu B < %zu B", workspace_bytes, pl.total);
    // (padding floats of the slab are never written by finalize: keep them zero like every other gradient producer)
    if (hipMemsetAsync(grads, 0, sizeof(float) * (size_t)n_sets * lay->theta_size, (hipStream_t)stream) != hipSuccess)
        return check_launch("avd_learn_set_fused_bf16: hipMemsetAsync(grads)");
    if (lay->S == 4)
        return fset::run<4>(*lay, n_agents, n_sets, theta, stats, theta_t, stats_t, s, a, r, s2, agent_weight, gamma, high, grads,
                            losses, (unsigned char*)workspace, pl, (hipStream_t)stream);
    return fset::run<3>(*lay, n_agents, n_sets, theta, stats, theta_t, stats_t, s, a, r, s2, agent_weight, gamma, high, grads, losses,
                        (unsigned char*)workspace, pl, (hipStream_t)stream);
}
