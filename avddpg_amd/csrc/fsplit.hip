// Split-operand fused shared-weight-set learner ("fsplit"): the same quantity as fset.hip -- Trainer.learn
// (workers/trainer.py:472-508) + the federated mean over the platoons (src/server/federated.py:47-63, 99-118;
// workers/trainer.py:400-431) for agents that SHARE their networks (interfrl, every step federated: trainer.py:121-128) at the
// reference widths 256 / 128 / 48 (src/config.py:112-117) -- but with f32-class results: the reference computes in float32
// (agent/model.py:26-36, 63-83) and single-rounded bf16 operands miss it by 1.7e-2 on the actor gradients (fset.hip).
//
// Every matrix product runs on v_mfma_f32_32x32x16_f16 with BOTH operands carried as fp16 PAIRS x = hi + lo
// (hi = rn16(x), lo = rn16(x - hi): |x - hi - lo| <= 2^-22 |x|, measured worst 2^-23; since r04 no operand is a bf16 pair --
// 2^-17 -- any more), accumulated in f32:
//     A . B  ~=  A_hi . B_hi + A_lo . B_hi + A_hi . B_lo          (the dropped lo . lo term is below the pairs' own residual)
// fp16's 5-bit exponent is handled by exact power-of-two scales folded into f32 constants (SW, SWC per set and net on the static
// second-layer operands; S1 = 64 on the first layers; 2^kg per set on the row factor |g3|, from the maximum the backward heads
// leave: set_gscale), and every conversion that could overflow is watched: *bad -> finalize writes NaN (see F16_OVERFLOW).
// Three structural facts keep this from costing 3x fset.hip:
//   * the output layers are one unit wide, so the second-layer gradient is rank one times a mask:
//         dZ2[row][n] = g3[row] * c3[n] * [z2[row][n] > 0]
//     The mask (with the sign of g3) is EXACT in fp16: it is what the heads write (as fp16 +-1 / 0, "sm"), c3[n] is folded
//     into the other operand (dx: W2c[f][n] = c3[n] W2[f][n], split once per call) or applied at the end (dw: finalize), and
//     |g3[row]| is folded into the first-layer input (dw: relu(W1 (|g| x) + |g| b1) = |g| P1, one MFMA with split operands
//     like every first layer here) or multiplied onto the f32 result (dx). dw and dx then need TWO MFMAs per product, and the
//     heads split nothing on their way out.
//   * second-layer weights are split once per call (prep) and stay in LDS (heads: hi + lo images, 135 / 160 KB, stored in the
//     k order in which a first-layer accumulator tile IS the next MFMA's B operand: no cross-lane exchange of activations)
//     or in registers (dx); only the activations are split on the fly (4 VALU per pair, in the shadow of the MFMAs).
//   * the action gradient of the actor loss needs no dZ2 at all: dmu[row] = g3 sum_n c3[n] [z2 > 0] M[n][row] with
//     M = W2T[:, action features] . (mask_a * wa), a second product over the action k-steps with the weight fragments the
//     forward pass has just read.
// Kernel chain, workspace, partial-sum layouts and the deterministic two-stage reduction are fset.hip's (finalize_* are
// shared, fset_common.h). Tested against the float64 oracle at 2e-5 of each tensor's max (tests/test_gpu_fsplit.py,
// test_gpu_configs_full.py: 5 x tighter than the exact-f32 kernels' 1e-4) -- not the 2 % the bf16 learners are allowed.
#include <type_traits>

#include "fset_common.h"

namespace avd {
namespace fsplit {
using namespace fset;

struct ActorS {
    static constexpr int K = 256, KP = 256, NFT = 8, NKS = 16, LD = 264, NGT = 9, KG = 288;
    static constexpr bool critic = false;
};
struct CriticS {
    static constexpr int K = 304, KP = 320, NFT = 10, NKS = 19, LD = 312, NGT = 10, KG = 320;  // 256 state + 48 action features
    static constexpr bool critic = true;
};
constexpr int NGT_MAX = 10;

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N - 1 (every index a constant inside f: `if constexpr`,
// no reliance on the unroller for register arrays)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// fp16 pairs (heads): 11 + 11 significant bits, |x - hi - lo| <= 2^-24 |x| -- an f32 value almost exactly. The heads need
// it: the hi + lo residual of a WEIGHT is the same for every batch row, so it biases q and y instead of averaging out, and
// the TD error q - y amplifies the bias by |q| / |q - y| (measured with bf16 pairs: 1e-3 of a gradient tensor's max at 51
// platoons, same for the mean of 51 single-platoon calls). fp16 has a 5-bit exponent, so the static operands are scaled by
// powers of two (exact) into its upper range and the scales are folded into f32 constants downstream: second-layer weights
// by SW (per set and net, max |W| SW in [2^12, 2^13)), first-layer weights and biases by S1 (activations then sit 2^6 higher,
// their lo parts stay normal numbers). An activation S1 P1 >= 65520 (P1 >= 1023.75) would round to fp16 inf -- and, the file being
// built with -fno-honor-nans (relu = one v_max_f32, which returns its non-NaN operand), the resulting NaN accumulators would
// silently become zeros. Every conversion to fp16 that can overflow is therefore TESTED: the scaled static operands in the prep
// kernels, the inputs in pack_x_kernel, and the running maximum of every first-layer accumulator tile in the forward heads
// (8 v_max3_f32 per tile); a hit sets *bad and finalize writes NaN into the whole gradient slab, like a non-finite input.
typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float S1 = 64.f;
constexpr float F16_OVERFLOW = 65520.f;  // the smallest float that rounds to fp16 infinity
// timing ablations (AVD_FSPLIT_ABL, results wrong by design) exist in the DIAGNOSTIC build only: in the product the bits are the
// constant 0 and every branch on them folds away
#if defined(FSPLIT_ABL_CONST)
#define FSPLIT_ABL(x) FSPLIT_ABL_CONST  // (A/B builds with the ablation compiled in: tools/fsplit_abl_const.sh)
#elif defined(AVD_DIAG)
#define FSPLIT_ABL(x) (x)
#else
#define FSPLIT_ABL(x) 0
#endif
__device__ __forceinline__ void split2h(float a, float b, unsigned& hi, unsigned& lo) {
    const f32x2 f = {a, b};
    const f16x2 h = __builtin_convertvector(f, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    // residuals a - hi.lo16, b - hi.hi16 in one v_fma_mix_f32 each (the f16 halves are read in place: no conversion back)
    f32x2 l;
#ifdef __HIP_DEVICE_COMPILE__
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l[0]) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l[1]) : "v"(hi), "v"(b));
#else
    l[0] = a - (float)h[0], l[1] = b - (float)h[1];
#endif
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(l, f16x2));
}
__device__ __forceinline__ void split1h(float x, f16& hi, f16& lo) {
    hi = (f16)x;
    lo = (f16)(x - (float)hi);
}
// an optimisation barrier on 16 accumulator registers: what was computed into them is computed HERE
template <class V>
__device__ __forceinline__ void pin16(V& x) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float a = x[8 * q], b = x[8 * q + 1], c = x[8 * q + 2], d = x[8 * q + 3], e = x[8 * q + 4], f = x[8 * q + 5], g = x[8 * q + 6], h = x[8 * q + 7];
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        x[8 * q] = a, x[8 * q + 1] = b, x[8 * q + 2] = c, x[8 * q + 3] = d, x[8 * q + 4] = e, x[8 * q + 5] = f, x[8 * q + 6] = g, x[8 * q + 7] = h;
    }
}
__device__ __forceinline__ f16x8 fragh(unsigned a, unsigned b, unsigned c, unsigned d) {
    uint4 o;
    o.x = a, o.y = b, o.z = c, o.w = d;
    return __builtin_bit_cast(f16x8, o);
}
__device__ __forceinline__ f32x16 mfmah(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// the first-layer fragments, fp16 pairs (slot layout: make_xg below with g = 1; sc scales weights and bias)
__device__ __forceinline__ f16x8 make_xh(float x0, float x1, float x2, float x3, int h) {
    f16 a0, a1, a2, a3, l0, l1, l2, l3;
    split1h(x0, a0, l0), split1h(x1, a1, l1), split1h(x2, a2, l2), split1h(x3, a3, l3);
    const f16 one = (f16)1.f, zero = (f16)0.f;
    f16x8 v;
    v[0] = a0, v[1] = a1, v[2] = a2, v[3] = a3;
    v[4] = h ? one : l0, v[5] = h ? one : l1, v[6] = h ? zero : l2, v[7] = h ? zero : l3;
    return v;
}
__device__ __forceinline__ f16x8 make_wh(float w0, float w1, float w2, float w3, float b, float sc, int h) {
    f16 a0, a1, a2, a3, l0, l1, l2, l3, bh, bl;
    split1h(sc * w0, a0, l0), split1h(sc * w1, a1, l1), split1h(sc * w2, a2, l2), split1h(sc * w3, a3, l3), split1h(sc * b, bh, bl);
    const f16 zero = (f16)0.f;
    f16x8 v;
    v[0] = h ? l0 : a0, v[1] = h ? l1 : a1, v[2] = h ? l2 : a2, v[3] = h ? l3 : a3;
    v[4] = h ? bh : a0, v[5] = h ? bl : a1, v[6] = h ? bh : a2, v[7] = h ? zero : a3;  // (k slot 14 = b_hi: meets g_lo in dw_kernel, 0 in the heads)
    return v;
}
// dw_kernel's input fragment: the row factor g = |g3| 2^kg rides in the input, (g x) and g as fp16 pairs:
//   k slot      0..3        4..7        8..11       12       13       14      15
//   input  x:   (gx)_hi     (gx)_lo     (gx)_hi     g_hi     g_hi     g_lo    0
//   weight w:   w_hi        w_hi        w_lo        b_hi     b_lo     b_hi    0        (make_wh, scaled by S1)
// = S1 g (x . w + b): relu(W1 (g x) + g b1) = g relu(z1), one MFMA like every first layer here.
__device__ __forceinline__ f16x8 make_xg(float x0, float x1, float x2, float x3, float g, int h) {
    f16 a0, a1, a2, a3, l0, l1, l2, l3, gh, gl;
    split1h(x0, a0, l0), split1h(x1, a1, l1), split1h(x2, a2, l2), split1h(x3, a3, l3), split1h(g, gh, gl);
    const f16 zero = (f16)0.f;
    f16x8 v;
    v[0] = a0, v[1] = a1, v[2] = a2, v[3] = a3;
    v[4] = h ? gh : l0, v[5] = h ? gh : l1, v[6] = h ? gl : l2, v[7] = h ? zero : l3;
    return v;
}

template <int S>
__device__ __forceinline__ void load_x(const float* xa, long row, float (&x)[4]) {
    if (S == 4) {
        const float4 v = *(const float4*)(xa + 4 * row);
        x[0] = v.x, x[1] = v.y, x[2] = v.z, x[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = k < S ? xa[S * row + k] : 0.f;
    }
}

// position of feature f in the heads' weight image: inside each tile of 32, bits 2 and 3 of the index trade places. Element
// jj of lane half h of k-step s of a first-layer accumulator tile [feature][row] is feature 16 s + 8 (jj >> 2) + 4 h + (jj & 3)
// (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"); stored at 16 s + 8 h + jj the weights of those 8
// features are ONE 16-byte LDS read of the A operand's lane (n, h).
__host__ __device__ constexpr int wpos(int f) { return (f & ~12) | ((f & 4) << 1) | ((f & 8) >> 1); }

// one network's operands (device pointers into the caller's slabs and the workspace)
struct NetP {
    const float* th;   // theta or theta_t (+ actor_size for a critic); set stride th_stride
    long th_stride;
    const f16* Whi;    // [sets][H2][K]   fp16 hi of SW * inv1[f] * W2[f][n] at [n][wpos(f)]: the heads' LDS image
    const f16* Wlo;    //                 its lo
    const f16* Wchi;   // [sets][KP][H2]  fp16 hi of SWC * c3[n] * W2[f][n], rows >= K zero: dx_kernel's resident operand (online nets)
    const f16* Wclo;   //                 its lo (SWC: vec[2 H2 + 2])
    const f16x8* wf1h; // [sets][NGT_MAX][64] first-layer weight fragments per feature tile and lane, fp16 pairs scaled by S1 (feature K:
                       // the constant one, relu(0 x + 1): its row of dw_kernel's G is db2)
    const float* vec;  // [sets][VEC]: b2'[128] = b2 + sh1 . W2, c3[128] = inv2 * w3, d3 = b3 + sh2 . w3, SW, SWC
    const unsigned* wap;  // [sets][2 hi/lo][2 h][3 k-steps][4] packed fp16 pairs of S1 * wa, the critic's action-layer weights (HEAD_BOTH's sweep B2)
};

// ---- pack: split first-layer input fragments of every batch row, once per learn call (+ the finiteness test) ----------
// outh: fp16 pairs (heads, dx_kernel); extra: the per-row scalar that travels with x (a: an fp16 fragment too; r: only tested)
// One launch packs both inputs of a learn call: blockIdx.y = 0 -> (s, a: the action travels as an fp16 fragment too), 1 -> (s2, r: only
// tested for finiteness).
struct PackArgs {
    const float* x[2];
    const float* extra[2];
    f16x8* outh[2];
    long rows;
    int* bad;
};
template <int S>
__global__ __launch_bounds__(256) void pack_x_kernel(const PackArgs a) {
    const int which = blockIdx.y;
    const float* x = a.x[which];
    const float* extra = a.extra[which];
    const bool extra_is_action = which == 0;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * a.rows) return;
    float v[4];
    load_x<S>(x, i >> 1, v);
    bool nf = not_finite(v[0]) || not_finite(v[1]) || not_finite(v[2]) || not_finite(v[3]);
    nf = nf || fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) >= F16_OVERFLOW;  // (fp16 input fragments)
    if (!(i & 1)) {
        const float e = extra[i >> 1];  // a (an fp16 input fragment of the critic too) or r
        nf = nf || not_finite(e) || (extra_is_action && fabsf(e) >= F16_OVERFLOW);
    }
    if (nf) atomicOr(a.bad, 1);
    a.outh[which][i] = make_xh(v[0], v[1], v[2], v[3], (int)(i & 1));
}

// ---- operand preparation ---------------------------------------------------------------------------------------------
struct PrepArgs {
    avd_mlp_layout L;
    int S;
    const float *theta, *stats, *theta_t, *stats_t;
    f16 *Whi[4], *Wlo[4];  // net 0 actor, 1 critic, 2 target actor, 3 target critic
    f16 *Wchi[4], *Wclo[4];  // online nets only
    f16x8* wf1h[4];
    float* vec[4];
    unsigned* wap;
    unsigned* smax;  // [4 nets][sets][2] maxima of scale_kernel (zeroed with *bad)
    int* bad;
};
// SW per (net, set): the power of two that puts max |inv1[f] W2[f][n]| into [2^12, 2^13) -> vec[2 H2 + 1]; online nets also
// SWC, the same for dx_kernel's operand c3[n] W2[f][n] -> vec[2 H2 + 2]. Two steps: scale_kernel leaves the two maxima per (net,
// set) in smax (atomicMax on the bits of non-negative floats: order-independent, so deterministic; 16 blocks per matrix, coalesced
// -- r03's one block per matrix walked 304 rows of 512 bytes with one thread each: 11 us, 33 with the second maximum), every
// thread of prep_kernel turns them into the powers of two, block n = 0 writes them for the later kernels.
constexpr int SCALE_SLICES = 16;
__device__ __forceinline__ float pow2_for(float mx) {  // the power of two that puts mx into [2^12, 2^13)
    int e = 0;
    if (mx > 0.f && !not_finite(mx)) (void)frexpf(mx, &e);  // mx = fr * 2^e, fr in [0.5, 1)
    int k = 13 - e;
    k = k < -14 ? -14 : (k > 30 ? 30 : k);
    return ldexpf(1.f, k);
}
__global__ __launch_bounds__(256) void scale_kernel(const PrepArgs a) {
    __shared__ float red[256], red2[256], c3a[H2], inva[CriticS::K];
    const int net = blockIdx.x, set = blockIdx.y, slice = blockIdx.z, tid = threadIdx.x;
    const bool critic = net & 1, target = net >= 2;
    const avd_mlp_layout& L = a.L;
    const float* th = (target ? a.theta_t : a.theta) + (long)set * L.theta_size + (critic ? L.actor_size : 0);
    const float* st = (target ? a.stats_t : a.stats) + (long)set * L.stats_size;
    const int K = critic ? CriticS::K : ActorS::K, oW2 = critic ? L.cW2 : L.aW2;
    const int og2 = critic ? L.cg3 : L.ag2, omv2 = critic ? L.cmv3 : L.amv2, oW3 = critic ? L.cW3 : L.aW3;
    if (tid < H2) c3a[tid] = fabsf((1.0f / sqrtf(st[omv2 + tid] + BN_EPS)) * th[og2 + tid] * th[oW3 + tid]);  // |c3[n]| = |inv2[n] w3[n]|
    for (int f = tid; f < K; f += 256) {
        int og, omv, ff = f;
        if (!critic) og = L.ag1, omv = L.amv1;
        else if (f < H1) og = L.cgs, omv = L.cmvs;
        else og = L.cga, omv = L.cmva, ff = f - H1;
        inva[f] = fabsf((1.0f / sqrtf(st[omv + ff] + BN_EPS)) * th[og + ff]);
    }
    __syncthreads();
    const int total = K * H2, per = (total + SCALE_SLICES - 1) / SCALE_SLICES, lo = slice * per, hi = min(total, lo + per);
    float m = 0.f, m2 = 0.f;
    for (int i = lo + tid; i < hi; i += 256) {
        const float w = fabsf(th[oW2 + i]);
        m = fmaxf(m, inva[i / H2] * w);
        m2 = fmaxf(m2, c3a[i % H2] * w);
    }
    red[tid] = m, red2[tid] = m2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]), red2[tid] = fmaxf(red2[tid], red2[tid + o]);
        __syncthreads();
    }
    if (tid == 0) {
        unsigned* dst = a.smax + ((long)net * gridDim.y + set) * 2;
        atomicMax(dst, __float_as_uint(red[0]));  // (non-negative floats order like their bits; a NaN / inf input lands on top and is
        atomicMax(dst + 1, __float_as_uint(red2[0]));  //  caught by the finiteness tests of prep_kernel)
    }
}
// one block per (output column n, net, set), one thread per feature f
__global__ __launch_bounds__(320) void prep_kernel(const PrepArgs a) {
    __shared__ float red[320];
    const int n = blockIdx.x, net = blockIdx.y, set = blockIdx.z, f = threadIdx.x;
    const bool critic = net & 1, target = net >= 2;
    const avd_mlp_layout& L = a.L;
    const float* th = (target ? a.theta_t : a.theta) + (long)set * L.theta_size + (critic ? L.actor_size : 0);
    const float* st = (target ? a.stats_t : a.stats) + (long)set * L.stats_size;
    const int K = critic ? CriticS::K : ActorS::K, KP = critic ? CriticS::KP : ActorS::KP;
    const int oW2 = critic ? L.cW2 : L.aW2, ob2 = critic ? L.cb2 : L.ab2, oW3 = critic ? L.cW3 : L.aW3, ob3 = critic ? L.cb3 : L.ab3;
    const int og2 = critic ? L.cg3 : L.ag2, obe2 = critic ? L.cbe3 : L.abe2, omm2 = critic ? L.cmm3 : L.amm2, omv2 = critic ? L.cmv3 : L.amv2;
    const float inv2n = (1.0f / sqrtf(st[omv2 + n] + BN_EPS)) * th[og2 + n], c3n = inv2n * th[oW3 + n];
    const unsigned* smax = a.smax + ((long)net * gridDim.z + set) * 2;
    const float SW = pow2_for(__uint_as_float(smax[0])), SWC = pow2_for(__uint_as_float(smax[1]));
    float shw = 0.f;
    if (f < KP) {
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
            f16 hi, lo;
            const float ws = SW * (inv * w);  // in [-2^13, 2^13] unless pow2_for had to clamp
            if (fabsf(ws) >= F16_OVERFLOW) atomicOr(a.bad, 1);
            split1h(ws, hi, lo);
            const long at = ((long)set * H2 + n) * K + wpos(f);
            a.Whi[net][at] = hi, a.Wlo[net][at] = lo;
        }
        if (!target) {
            // dx_kernel's static operand c3[n] W2[f][n]: an fp16 pair too (as a bf16 pair its 2^-17 residual, the same for every
            // batch row, stood at 1e-5 ... 2e-5 of max in the first-layer gradients: r04, tests/test_gpu_configs_full.py)
            f16 hi, lo;
            const float wc = SWC * (c3n * w);
            if (fabsf(wc) >= F16_OVERFLOW) atomicOr(a.bad, 1);
            split1h(wc, hi, lo);
            const long at = ((long)set * KP + f) * H2 + n;
            a.Wchi[net][at] = hi, a.Wclo[net][at] = lo;
        }
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
    if (f == 0) {
        if (n == 0) vec[2 * H2 + 1] = SW, vec[2 * H2 + 2] = SWC;
        vec[n] = th[ob2 + n] + red[0];
        vec[H2 + n] = c3n;
        if (not_finite(vec[n]) || not_finite(c3n) || not_finite(th[ob3])) atomicOr(a.bad, 1);
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
// first-layer weight fragments: one block per (feature tile, net, set), one thread per lane of the fragment; tiles >= 8 of a
// critic are its action layer; feature K of every net is the constant one relu(0 x + 1) (dw_kernel: its row of G is db2)
__global__ __launch_bounds__(64) void prep1_kernel(const PrepArgs a) {
    const int ft = blockIdx.x, net = blockIdx.y, set = blockIdx.z, lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const bool critic = net & 1, target = net >= 2;
    const avd_mlp_layout& L = a.L;
    const float* th = (target ? a.theta_t : a.theta) + (long)set * L.theta_size + (critic ? L.actor_size : 0);
    const int K = critic ? CriticS::K : ActorS::K;
    const int f = 32 * ft + r;
    float w[4] = {0.f, 0.f, 0.f, 0.f}, b = 0.f;
    if (f == K) {
        b = 1.f;
    } else if (f < K) {
        if (critic && ft >= 8) {
            w[0] = th[L.cWa + (f - H1)], b = th[L.cba + (f - H1)];
        } else {
            const float* W = th + (critic ? L.cWs : L.aW1);
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = k < a.S ? W[k * H1 + f] : 0.f;
            b = th[(critic ? L.cbs : L.ab1) + f];
        }
        if (not_finite(w[0]) || not_finite(w[1]) || not_finite(w[2]) || not_finite(w[3]) || not_finite(b)) atomicOr(a.bad, 1);
        // the heads' and dx's fragments carry S1 w, S1 b as fp16 pairs
        if (S1 * fmaxf(fmaxf(fabsf(w[0]), fabsf(w[1])), fmaxf(fmaxf(fabsf(w[2]), fabsf(w[3])), fabsf(b))) >= F16_OVERFLOW) atomicOr(a.bad, 1);
    }
    a.wf1h[net][((long)set * NGT_MAX + ft) * 64 + lane] = make_wh(w[0], w[1], w[2], w[3], b, S1, h);
    if (net == 1 && ft == 0 && lane < 48) {
        // HEAD_BOTH's B operand of M = W2T[:, action] . (mask_a * wa): element jj of lane half hh of action k-step ks is action
        // feature fa = 16 ks + 8 (jj >> 2) + 4 hh + (jj & 3); entry [hl][hh][ks][m] packs elements jj = 2 m, 2 m + 1
        const int hl = lane / 24, hh = (lane / 12) & 1, ks = (lane / 4) % 3, m = lane & 3;
        unsigned v = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int jj = 2 * m + e, fa = 16 * ks + 8 * (jj >> 2) + 4 * hh + (jj & 3);
            f16 hi, lo;
            split1h(S1 * th[L.cWa + fa], hi, lo);
            v |= (unsigned)__builtin_bit_cast(unsigned short, hl ? lo : hi) << (16 * e);
        }
        a.wap[(long)set * 48 + lane] = v;
    }
}

// ---- head: first layer -> second-layer GEMM (three MFMAs per product) -> output layer [-> its backward] -------------------
// (2..4 were r03's HEAD_CRITIC / HEAD_CONST / HEAD_ACTOR: two-launch critic heads and the second actor forward, gone in r04)
enum HeadMode { OUT_TANH = 0, OUT_TD = 1, HEAD_BOTH = 5, OUT_TANH_SAVE = 6 };
struct HeadArgs {
    unsigned long long* stamp;  // diagnostic build (-DAVD_STAMP) only: [8 waves][8] accumulated s_memtime deltas of workgroup 16
    NetP net;
    int n_agents, n_sets;
    const f16x8* xf;   // [n_agents][64][2] packed first-layer input fragments of the states, fp16 pairs (pack_x_kernel)
    const float* act;  // [n_agents][64] the critic's action input (a, a' or mu)
    const float* act2; // HEAD_BOTH: mu (act = a)
    const float* r;    // OUT_TD: rewards [n_agents][64]
    const float* yin;  // HEAD_BOTH: TD targets
    const float* aw;   // per-agent factor on the loss seeds (weighted federated mean) or NULL
    float* out;        // OUT_*: per-row result
    f16* sm;           // HEAD_BOTH: sign(g3) * [z2 > 0] as fp16 +-1 / 0 (EXACT); OUT_TANH_SAVE: [z2 > 0] as 1 / 0; [n_agents][64][128]
    float* g3;         // HEAD_BOTH: the row factor of dZ2 [n_agents][64]
    float* dmu;        // HEAD_BOTH: dLa/dmu per row [n_agents][64]
    float* part_s;     // HEAD_*: [grid][8 waves][2] sums of the seeds and of the loss terms
    float* part_s2;    // HEAD_BOTH: the same sums of the critic(s, mu) branch (the actor loss)
    float* part_m;     // backward modes: [grid][8 waves] max |g3| over the wave's rows (dw / dx scale their fp16 operands by it)
    float* tz;         // OUT_TANH_SAVE: tanh(z) per row [n_agents][64] (the actor's backward seed needs 1 - t^2: actor_seed_kernel)
    float gamma, high, inv_n;
    int* bad;          // set when an activation would overflow fp16 (S1 P1 >= 65520): finalize then writes NaN gradients
    int abl;           // diagnostic build only (AVD_FSPLIT_ABL bit 4; 0 in the product): the relu / hi-lo split VALU of every first-layer tile
                       // is skipped (raw bits as operands: WRONG results) -- what the heads' time owes to that VALU work
};
// Workgroup = 8 waves bound to one weight set; LDS holds the fp16 hi and lo images of its BN-folded, scaled second-layer
// weights for the workgroup's whole life. Wave w owns rows [32 (w & 1), +32) of every 4th tile. Per feature tile: the first layer of ITS rows
// on the matrix cores (lane = batch row, registers = features), relu, hi / lo split of the 16 values (VALU), and per k-step
// and 32-column tile three MFMAs (A = weight fragments from LDS, B = the split activations as they stand in the registers).
//
// Modes: OUT_TANH (target actor: a'), OUT_TANH_SAVE (actor: mu, + the relu masks and tanh(z) its backward pass needs), OUT_TD (target
// critic -> y) and HEAD_BOTH: critic(s, a) and critic(s, mu) see the same states through the same weights, so 8 of their 10
// first-layer tiles and 192 of their 228 second-layer MFMAs per 32 rows are the SAME work -- one pass over the state tiles, then
// three sweeps over the action k-steps on the one accumulator set: A (input a: TD seed, signed masks), B1 (input mu, by
// linearity), B2 (the action gradient through M).
// (r04, tried: the forward-only modes with 64 rows per wave -- both row halves against every weight fragment pair, six MFMAs per
// two LDS reads instead of three, 128 accumulator registers: OUT_TANH 190.9 us against 190, OUT_TD 220.8 against 223 on the same
// box. Halving the LDS bytes per MFMA buys nothing: the heads are not LDS-bound. Not kept.)
#ifndef HEAD_FAST
#define HEAD_FAST 5
#define HEAD_SLOW 3
#endif

template <int S, class NET, int MODE>
__global__ __launch_bounds__(NT) void head_kernel(const HeadArgs p) {
    constexpr int K = NET::K, NKS = NET::NKS, NFT = NET::NFT, LD = NET::LD;  // LD/2 = 4 (mod 8) dwords: conflict-free b128
    constexpr int NTH = NT;  // 8 waves: two per SIMD at <= 256 registers
    constexpr bool BOTH = (MODE == HEAD_BOTH);
    // (T1 = sum_rows g3 relu(z2) is not accumulated here: finalize derives it from dw_kernel's partials, FinArgs::t1_from_g)
    static_assert(!BOTH || NET::critic, "HEAD_BOTH is a critic mode");
    __shared__ __attribute__((aligned(16))) f16 wimg[2][H2 * LD];
    __shared__ __attribute__((aligned(16))) float b2s[H2];
    __shared__ __attribute__((aligned(16))) float c3s[H2];
    __shared__ __attribute__((aligned(16))) unsigned waps[BOTH ? 48 : 4];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5, q = w >> 1, rh = w & 1;
    const int set = blockIdx.x % p.n_sets, j0 = blockIdx.x / p.n_sets, J = gridDim.x / p.n_sets, P = p.n_agents / p.n_sets;
    for (int i = tid; i < 2 * H2 * (K / 8); i += NTH) {
        const int hl = i / (H2 * (K / 8)), rem = i - hl * (H2 * (K / 8)), n = rem / (K / 8), c = rem - n * (K / 8);
        const f16* src = (hl ? p.net.Wlo : p.net.Whi) + ((long)set * H2 + n) * K + 8 * c;
        *(uint4*)(&wimg[hl][n * LD + 8 * c]) = *(const uint4*)src;
    }
    const float* vec = p.net.vec + (long)set * VEC;
    // acc = SW S1 z2: the scales of the fp16 operands are folded into the f32 tables (bias in, output weights out)
    const float SW = vec[2 * H2 + 1], sc = SW * S1, isc = 1.f / sc;
    if (tid < H2) b2s[tid] = vec[tid] * sc, c3s[tid] = vec[H2 + tid] * isc;
    if (BOTH && tid < 48) waps[tid] = p.net.wap[(long)set * 48 + tid];
    const float d3 = vec[2 * H2];
    const f16x8* wf1 = p.net.wf1h + (long)set * NGT_MAX * 64 + lane;  // + 64 ft
    const f32x16 zero16 = {};
#ifdef AVD_STAMP
    unsigned long long hacc[4] = {0, 0, 0, 0}, hlast = __builtin_amdgcn_s_memtime();
    const unsigned long long hstart = hlast;
#define HSTAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); hacc[i] += t_ - hlast; hlast = t_; }
#else
#define HSTAMP(i)
#endif
    float Dacc = 0.f, Lacc = 0.f, Dacc2 = 0.f, Lacc2 = 0.f, gmax = 0.f;
    // fp16 overflow watch. An activation S1 P1 >= 65520 converts to the pair (hi, lo) = (+inf, -inf), and whatever the weights,
    // w_hi inf + w_lo inf - w_hi inf is NaN in EVERY second-layer accumulator of that batch row (0 inf is NaN too): one
    // accumulator per row and sweep is looked at (bit test: the file is built with -fno-honor-nans), |bits| max-accumulated.
    unsigned watch = 0;
    auto look = [&](float x) {
        unsigned u = __float_as_uint(x);
        asm volatile("" : "+v"(u));  // (keeps the no-nans optimiser from reasoning about the float)
        u &= 0x7fffffffu;
        watch = watch > u ? watch : u;
    };
    __syncthreads();

    const int ntile = j0 < P ? (P - j0 + J - 1) / J : 0;  // tiles of this workgroup: j0, j0 + J, ..
    const int row = 32 * rh + r;
    f16x8 nx = {};
    float na = 0.f, nb = 0.f, ny = 0.f, nw = 1.f;
    auto fetch_in = [&](int k) {  // one unit (32 rows of a tile) ahead
        const int agent = (j0 + k * J) * p.n_sets + set;
        const long ri = (long)agent * TILE + row;
        nx = p.xf[2 * ri + h];
        if (NET::critic) na = p.act[ri];
        if (BOTH) nb = p.act2[ri];
        if (MODE == OUT_TD) ny = p.r[ri];
        if (BOTH) ny = p.yin[ri];
        if (BOTH && p.aw) nw = p.aw[agent];
    };
    const f16* whi0 = &wimg[0][r * LD + 8 * h];  // + 32 t LD + 16 ks
    const f16* wlo0 = &wimg[1][r * LD + 8 * h];
    // the sequence of feature tiles of a unit: tiles 0 .. NFT - 1; HEAD_BOTH: + the action tiles 8, 9 once more (input mu)
    constexpr int NSEQ = BOTH ? 8 : NFT;  // HEAD_BOTH: the state tiles; the action tiles are streamed per column tile below
    HSTAMP(3);  // image fill, tables, first fetch
    // The two waves of a SIMD (w and w + 4) are arbitrated oldest first: with equal shares, waves 0..3 ran 12.0 k cycles per unit
    // and waves 4..7 20.4 k until the old ones were done, then finished alone at 61 % of the matrix pipe (s_memtime stamps, r03;
    // s_setprio does not change it). The tiles are therefore dealt FAST : SLOW per wave pair in periods of 2 (FAST + SLOW) tiles
    // -- a static map, so the grouping of the partial sums (the bits of the result) stays a function of the plan.
    constexpr int FAST = HEAD_FAST, SLOW = HEAD_SLOW, PERIOD = 2 * (FAST + SLOW);
    const int qs = __builtin_amdgcn_readfirstlane(q);  // (scalar tile arithmetic)
    const int cnt = qs < 2 ? FAST : SLOW, start = qs < 2 ? qs * FAST : 2 * FAST + (qs - 2) * SLOW;
    int k = start, pos = 0;  // the wave pair's current tile and its place in the pair's group of cnt
    auto next_tile = [&](int kk, int pp) { return pp + 1 == cnt ? kk + PERIOD - cnt + 1 : kk + 1; };
    if (k < ntile) fetch_in(k);
    for (int ui = 0; k < ntile; ++ui, k = next_tile(k, pos), pos = pos + 1 == cnt ? 0 : pos + 1) {
        const int agent = (j0 + k * J) * p.n_sets + set;
        const long ri = (long)agent * TILE + row;
        const f16x8 xs = nx, xa = make_xh(na, 0.f, 0.f, 0.f, h), xb = BOTH ? make_xh(nb, 0.f, 0.f, 0.f, h) : xa;
        const float ty = ny, tw = nw;
        if (next_tile(k, pos) < ntile) fetch_in(next_tile(k, pos));
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *(const float4*)(b2s + 32 * t + 8 * g + 4 * h);
                acc[t][4 * g] = b.x, acc[t][4 * g + 1] = b.y, acc[t][4 * g + 2] = b.z, acc[t][4 * g + 3] = b.w;
            }
        // relu + hi / lo split of a first-layer tile [feature][row] (row on the lane): pair m of k-step s = registers 8 s + 2 m, + 1
        unsigned ph[8], pl[8];
        auto split16 = [&](const f32x16& p1) {
            if (FSPLIT_ABL(p.abl) & 4) {  // (timing ablation: no VALU at all)
#pragma unroll
                for (int m = 0; m < 8; ++m) ph[m] = __float_as_uint(p1[2 * m]), pl[m] = __float_as_uint(p1[2 * m + 1]);
                return;
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) split2h(relu(p1[2 * m]), relu(p1[2 * m + 1]), ph[m], pl[m]);
        };
        // ---- epilogues -----------------------------------------------------------------------------------------------
        // per 32-column tile t (the lane holds columns 32 t + 8 g + 4 h + j of its row, its partner the other half):
        auto zdot_t = [&](f32x16& a, int t) {  // relu in place; returns this lane's part of sum_n c3[n] relu(z2[n])
            float zp = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 c = *(const float4*)(c3s + 32 * t + 8 * g + 4 * h);
                const float cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[4 * g + j] = relu(a[4 * g + j]);
                    zp = fmaf(a[4 * g + j], cc[j], zp);
                }
            }
            return zp;
        };
        auto zdot_nd = [&](const f32x16& a, int t) {  // the same without touching a
            float zp = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 c = *(const float4*)(c3s + 32 * t + 8 * g + 4 * h);
                const float cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) zp = fmaf(relu(a[4 * g + j]), cc[j], zp);
            }
            return zp;
        };
        // the relu mask of tile t as fp16 (a = z2 or relu(z2)): s16 = 0x3c00 (+1: unsigned, OUT_TANH_SAVE) or the seed's sign, 0x3c00 / 0xbc00
        auto mask_t = [&](const f32x16& a, int t, unsigned s16, f16* dst) {
            unsigned pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 4 * g + 2 * e;
                    pk[g][e] = (a[i] > 0.f ? s16 : 0u) | (a[i + 1] > 0.f ? s16 << 16 : 0u);
                }
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {  // 16-byte row-major pieces: the row's two lanes cover 32 contiguous bytes
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gg][0], pk[2 * gg + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gg][1], pk[2 * gg + 1][1], false, false);
                uint4 o;
                o.x = s0[0], o.y = s1[0], o.z = s0[1], o.w = s1[1];
                *(uint4*)(dst + 32 * t + 16 * gg) = o;
            }
        };
        auto out_z = [&](f32x16 (&ac)[4]) {  // relu in place, output layer: z = d3 + sum_n c3[n] relu(z2[n])
            float zp = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) zp += zdot_t(ac[t], t);
            zp += __shfl_xor(zp, 32);
            return d3 + zp;
        };
        // ---- the tiles ---------------------------------------------------------------------------------------------------
        split16(mfmah(wf1[0], xs, zero16));
        f16x8 wfn = wf1[64];
        // Weight fragments are read ONE GROUP AHEAD (group = k-step x pair of column tiles: hi and lo of two tiles, 16
        // registers, six MFMAs): a ds_read_b128 issued right in front of the MFMA that consumes it exposes the LDS latency
        // four times per feature tile. wg[parity][2 t][hi, lo].
        f16x8 wg[2][2][2];
        auto read_group = [&](int g, f16x8 (&dst)[2][2]) {  // g = 2 ks + th
            const int ks = g >> 1, th = g & 1;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                dst[tt][0] = *(const f16x8*)(whi0 + 32 * (2 * th + tt) * LD + 16 * ks);
                dst[tt][1] = *(const f16x8*)(wlo0 + 32 * (2 * th + tt) * LD + 16 * ks);
            }
        };
        read_group(0, wg[0]);
        auto tile = [&](auto si_c, f32x16 (&ac)[4]) {
            constexpr int si = decltype(si_c)::value;
            constexpr int ft = si;                          // feature tile
            // the first layer of the next tile of the sequence
            f32x16 p1n = zero16;
            if constexpr (si + 1 < NSEQ) {
                p1n = mfmah(wfn, (NET::critic && si + 1 >= 8) ? xa : xs, zero16);
                if constexpr (si + 2 < NSEQ) wfn = wf1[64 * (si + 2)];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ks = 2 * ft + s;
                if (ks >= NKS) continue;  // (the critic's last tile holds only 16 features)
                const f16x8 bhi = fragh(ph[4 * s], ph[4 * s + 1], ph[4 * s + 2], ph[4 * s + 3]);
                const f16x8 blo = fragh(pl[4 * s], pl[4 * s + 1], pl[4 * s + 2], pl[4 * s + 3]);
#pragma unroll
                for (int th = 0; th < 2; ++th) {
                    const int g = 2 * ks + th;
                    if (g + 1 < 2 * (BOTH ? 16 : NKS)) read_group(g + 1, wg[(g + 1) & 1]);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int t = 2 * th + tt;
                        const f16x8 whi = wg[g & 1][tt][0], wlo = wg[g & 1][tt][1];
                        ac[t] = mfmah(whi, bhi, ac[t]);
                        ac[t] = mfmah(wlo, bhi, ac[t]);
                        ac[t] = mfmah(whi, blo, ac[t]);
                    }
                }
            }
            if constexpr (si + 1 < NSEQ) split16(p1n);
            // Issue order inside the tile (an MFMA holds the SIMD's issue port for 8 of its 32 cycles: the next tile's relu /
            // split VALU goes INTO the gaps, the next group's four fragment reads in front of each group of six MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // first-layer MFMA of the next tile
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
            }
            // (tiles stay in program order: hoisting every tile's first-layer MFMA and weight reads costs hundreds of registers)
            __builtin_amdgcn_sched_barrier(0);
        };
        HSTAMP(0);  // unit start: inputs, bias
        static_for<0, NSEQ>([&](auto si_c) { tile(si_c, acc); });
        HSTAMP(1);  // the feature tiles
        if constexpr (BOTH) {
            // acc = the state part of z2 (+ bias); the action part is 3 k-steps. Three sweeps of 36 MFMAs over the four column
            // tiles, all on the ONE set of accumulators (acc + T1 + M never coexist: <= 256 registers, two waves per SIMD):
            //   A : acc += W2T[:, action] . f(a)             -> z, TD seed, masks, T1 (acc left as z2, not relu'd)
            //   B1: acc += W2T[:, action] . (f(mu) - f(a))   -> z2 of the mu branch by linearity: q and 64 relu bits per lane
            //   B2: M = W2T[:, action] . (mask_a (.) wa)     -> dmu
            // (an earlier form kept both branches' accumulators and M alive: 256 + 244 registers, one wave per SIMD, 456 us)
            f16x8 fh[3], fl[3];
            auto act_sweep = [&](f32x16 (&c)[4]) {  // c[t] += W2T[tile t, action k-steps] . (fh, fl)
                f16x8 wq[2][2][2];
                read_group(2 * 16, wq[0]);
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    if (g + 1 < 6) read_group(2 * 16 + g + 1, wq[(g + 1) & 1]);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int t = 2 * (g & 1) + tt;
                        const f16x8 whi = wq[g & 1][tt][0], wlo = wq[g & 1][tt][1];
                        c[t] = mfmah(whi, fh[g >> 1], c[t]);
                        c[t] = mfmah(wlo, fh[g >> 1], c[t]);
                        c[t] = mfmah(whi, fl[g >> 1], c[t]);
                    }
                }
            };
            // ---- branch A: critic(s, a) -> TD seed, masks, T1 (workers/trainer.py:494-498)
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const f32x16 p1 = mfmah(wf1[64 * (8 + tl)], xa, zero16);
                split16(p1);
#pragma unroll
                for (int ss = 0; ss < 2 - tl; ++ss) {
                    fh[2 * tl + ss] = fragh(ph[4 * ss], ph[4 * ss + 1], ph[4 * ss + 2], ph[4 * ss + 3]);
                    fl[2 * tl + ss] = fragh(pl[4 * ss], pl[4 * ss + 1], pl[4 * ss + 2], pl[4 * ss + 3]);
                }
            }
            act_sweep(acc);
            look(acc[0][0]);
            {
                float zp = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) zp += zdot_nd(acc[t], t);
                zp += __shfl_xor(zp, 32);
                const float diff = d3 + zp - ty, g3a = 2.f * diff * p.inv_n * tw;
                if (h == 0) p.g3[ri] = g3a, Dacc += g3a, Lacc += diff * diff;
                gmax = fmaxf(gmax, fabsf(g3a));
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    mask_t(acc[t], t, g3a < 0.f ? 0xbc00u : 0x3c00u, p.sm + ri * H2 + 8 * h);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- branch B: critic(s, mu) -> the actor loss -mean(q) and its gradient w.r.t. the action (:501-504)
            f16x8 mh[3], ml[3];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const f32x16 pa = mfmah(wf1[64 * (8 + tl)], xa, zero16), pm = mfmah(wf1[64 * (8 + tl)], xb, zero16);
                unsigned dh[8], dl[8], mk[8];
#pragma unroll
                for (int m = 0; m < 4 * (2 - tl); ++m) {
                    split2h(relu(pm[2 * m]) - relu(pa[2 * m]), relu(pm[2 * m + 1]) - relu(pa[2 * m + 1]), dh[m], dl[m]);
                    mk[m] = (pm[2 * m] > 0.f ? 0xffffu : 0u) | (pm[2 * m + 1] > 0.f ? 0xffff0000u : 0u);
                }
#pragma unroll
                for (int ss = 0; ss < 2 - tl; ++ss) {
                    const int kk = 2 * tl + ss;
                    fh[kk] = fragh(dh[4 * ss], dh[4 * ss + 1], dh[4 * ss + 2], dh[4 * ss + 3]);
                    fl[kk] = fragh(dl[4 * ss], dl[4 * ss + 1], dl[4 * ss + 2], dl[4 * ss + 3]);
                    // B operand of M = mask_a * wa (hi, lo): the packed constants where the mu branch's activation is positive
                    const uint4 ch = *(const uint4*)(waps + (0 * 2 + h) * 12 + 4 * kk), cl = *(const uint4*)(waps + (1 * 2 + h) * 12 + 4 * kk);
                    mh[kk] = fragh(ch.x & mk[4 * ss], ch.y & mk[4 * ss + 1], ch.z & mk[4 * ss + 2], ch.w & mk[4 * ss + 3]);
                    ml[kk] = fragh(cl.x & mk[4 * ss], cl.y & mk[4 * ss + 1], cl.z & mk[4 * ss + 2], cl.w & mk[4 * ss + 3]);
                }
            }
            act_sweep(acc);
            look(acc[0][0]);  // (f(mu) - f(a) with f(mu) overflowed: +inf)
            float zq = 0.f;
            unsigned pos[2] = {0u, 0u};  // bit 16 (t & 1) + i of pos[t >> 1]: column i of tile t is active in the mu branch
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                zq += zdot_nd(acc[t], t);
#pragma unroll
                for (int i = 0; i < 16; ++i) pos[t >> 1] |= acc[t][i] > 0.f ? 1u << (16 * (t & 1) + i) : 0u;
            }
            zq += __shfl_xor(zq, 32);
            asm volatile("" : "+v"(pos[0]), "+v"(pos[1]));  // (acc dies here)
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) fh[kk] = mh[kk], fl[kk] = ml[kk];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = zero16;
            act_sweep(acc);
            // dmu[row] = g3 sum_n c3[n] [z2 > 0] M[n][row] (the accumulators hold SW S1 M, c3s carries 1 / (SW S1))
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 c = *(const float4*)(c3s + 32 * t + 8 * g + 4 * h);
                    const float cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) sum += (pos[t >> 1] >> (16 * (t & 1) + 4 * g + j) & 1u) ? cc[j] * acc[t][4 * g + j] : 0.f;
                }
            sum += __shfl_xor(sum, 32);
            const float g3b = -p.inv_n * tw;
            if (h == 0) p.dmu[ri] = g3b * sum, Dacc2 += g3b, Lacc2 += d3 + zq;
            continue;
        }
        look(acc[0][0]);
        const float z = out_z(acc);
        if (MODE == OUT_TANH) {
            const float o = tanhf(z) * p.high;
            if (h == 0) p.out[ri] = o;
        } else if (MODE == OUT_TANH_SAVE) {
            // mu = actor(s) AND what the actor's backward pass needs from this forward pass: the relu mask of z2 (fp16 1 / 0,
            // UNSIGNED: the seed's sign is not known yet -- it rides on the row factor in dw / dx) and tanh(z). With them
            // r03's HEAD_ACTOR launch -- the same 200 MFMAs per 32 rows once more, 239 us -- is not needed at all (r04).
            const float t = tanhf(z);
            if (h == 0) p.out[ri] = t * p.high, p.tz[ri] = t;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) mask_t(acc[tt], tt, 0x3c00u, p.sm + ri * H2 + 8 * h);
        } else {  // OUT_TD
            if (h == 0) p.out[ri] = ty + p.gamma * z;
        }
        HSTAMP(2);  // epilogue
#ifdef AVD_STAMP
        if (blockIdx.x == 16 && lane == 0 && p.stamp && (w == 0 || w == 4) && ui < 28)
            p.stamp[64 + (w >> 2) * 28 + ui] = hlast - hstart;  // end time of every unit of waves 0 and 4
#endif
    }
#ifdef AVD_STAMP
    if (blockIdx.x == 16 && lane == 0 && p.stamp)
        for (int i = 0; i < 4; ++i) p.stamp[w * 8 + i] = hacc[i];
#endif
    if (watch >= 0x7f800000u) atomicOr(p.bad, 1);
    if (BOTH) {
        // one partial per wave: sums over the 32 row lanes of each half (fixed shuffle tree), one writer per half
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            Dacc += __shfl_xor(Dacc, o), Lacc += __shfl_xor(Lacc, o);
            Dacc2 += __shfl_xor(Dacc2, o), Lacc2 += __shfl_xor(Lacc2, o);
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
        if (lane == 0) {
            const long slot = (long)blockIdx.x * 8 + w;
            p.part_m[slot] = gmax;
            p.part_s[slot * 2] = Dacc, p.part_s[slot * 2 + 1] = Lacc;
            p.part_s2[slot * 2] = Dacc2, p.part_s2[slot * 2 + 1] = Lacc2;
        }
    }
}

// ---- the actor's backward seed (r04: what is left of r03's HEAD_ACTOR launch) --------------------------------------------------
// g3[row] = d mu[row] * high * (1 - tanh(z)^2) (workers/trainer.py:502-506 through agent/model.py:34-36), from the d q / d mu the
// critic head left and the tanh(z) the mu pass stored; per wave max |g3| (set_gscale) and sum g3 (-> d b3) in head_kernel's layout.
struct SeedArgs {
    int n_agents, n_sets;
    const float *dmu, *tz;
    float high;
    float *g3, *part_m, *part_s;
};
__global__ __launch_bounds__(NT) void actor_seed_kernel(const SeedArgs p) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int set = blockIdx.x % p.n_sets, j0 = blockIdx.x / p.n_sets, J = gridDim.x / p.n_sets, P = p.n_agents / p.n_sets;
    float gmax = 0.f, D = 0.f;
    for (int pi = j0 + w * J; pi < P; pi += 8 * J) {  // wave w: tiles j0 + w J, + 8 J, .. of the workgroup's set; lane = row
        const long ri = (long)(pi * p.n_sets + set) * TILE + lane;
        const float t = p.tz[ri], g = p.dmu[ri] * p.high * (1.f - t * t);
        p.g3[ri] = g;
        gmax = fmaxf(gmax, fabsf(g));
        D += g;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o)), D += __shfl_xor(D, o);
    if (lane == 0) {
        p.part_m[(long)blockIdx.x * 8 + w] = gmax;
        p.part_s[((long)blockIdx.x * 8 + w) * 2] = D, p.part_s[((long)blockIdx.x * 8 + w) * 2 + 1] = 0.f;
    }
}

// ---- the row factor's scale --------------------------------------------------------------------------------------------------
// dw and dx carry |g3[row]| inside fp16 operands; g3 is 2 (q - y) / N or d mu * high * (1 - t^2): anything from 1e-12 to 1e-3. The
// backward heads leave max |g3| per wave (part_m); every workgroup of dw / dx reduces its set's J x 8 values to the power of two
// 2^kg with max |g3| 2^kg in [0.5, 1) (exact scaling; max() is order-independent: every workgroup gets the same bits) and
// divides it out of its partial sums at the end. With |g3| 2^kg < 1 an operand |g3| 2^kg S1 P1 stays below the S1 P1 the heads
// have already watched for fp16 overflow. Block-cooperative (one barrier); red: 8 floats of LDS.
__device__ __forceinline__ float set_gscale(const float* part_m, int set, int n_sets, int J, float* red) {
    const int tid = threadIdx.x;
    float m = 0.f;
    for (int i = tid; i < J * 8; i += NT) m = fmaxf(m, part_m[((long)(i >> 3) * n_sets + set) * 8 + (i & 7)]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
    int e = 0;
    if (m > 0.f && !not_finite(m)) (void)frexpf(m, &e);  // m = fr 2^e, fr in [0.5, 1)
    const int k = e > 100 ? -100 : (e < -100 ? 100 : -e);
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ldexpf(1.f, k))));  // (uniform: keep it in an SGPR)
}

// ---- dw: G[f][n] += sum_rows (|g3| P1)[row][f] * sm[row][n] over all tiles of the workgroup -------------------------------
struct DwArgs {
    NetP net;
    int n_agents, n_sets;
    const float* x;    // states [n_agents][64][S]
    const float* act;  // the critic's action input [n_agents][64]
    const float* g3;   // [n_agents][64]
    const float* part_m;  // [grid][8] max |g3| per wave of the kernel that wrote g3 (set_gscale)
    const f16* sm;     // [n_agents][64][128] fp16 +-1 / 0 (critic: sign(g3) inside, HEAD_BOTH) or 1 / 0 (actor: OUT_TANH_SAVE; the sign of
                       // g3 then goes onto the A operand: one v_xor per packed pair, from a per-row sign table in LDS)
    float* partG;      // [grid][KG][128] (row K: the constant-one feature = sum over rows of g3 * mask -> db2 / c3)
    int abl;           // diagnostic build only (AVD_FSPLIT_ABL; 0 in the product): timing ablations with WRONG results -- 1: every tile
                       // fetch reads the workgroup's first tile (no HBM traffic), 2: no workgroup barrier in the tile loop
    unsigned long long* stamp;  // diagnostic build (-DAVD_STAMP) only: [8 waves][8] accumulated s_memtime deltas of workgroup 16
};
// Wave w owns feature tile w (both row halves, all four column tiles: 64 accumulator registers). The tiles past the eight
// state tiles (critic: the action tiles 8, 9; actor: the constant-one tile 8) are cut into eight equal pieces, one per wave
// (critic: tile x row half x pair of column tiles; actor: row half x column tile; 32 more accumulator registers), whose
// row-half partials meet in LDS at the end -- every wave then does the same work per tile. A first-layer tile is evaluated
// with the batch rows as the M index and the row factor |g3| 2^kg in its input fragment (fp16 pairs: make_xg; result: feature on
// the lane, rows in the registers = the A operand of G = Q^T . sm in permuted k order), relu'd and split ONCE into an fp16 pair
// (r04: was a bf16 pair, 2^-17 per row -- in a noise-dominated sum that IS the relative error of the result) and used against
// all four column tiles: 48 VALU per 17 MFMAs. The sm tile goes through LDS (row stride 320 B: conflict-free ds_read_b64_tr_b16), fetched
// TWO tiles ahead; the |g3|-scaled input fragments are built once per tile by 128 threads and shared through LDS.
template <int S, class NET>
__global__ __launch_bounds__(NT) void dw_kernel(const DwArgs p) {
    constexpr int KG = NET::KG, LDZ = 160;
    constexpr int XC = NET::critic ? 2 : 1;  // column tiles of a wave's extra piece
    constexpr bool SGN = !NET::critic;       // unsigned masks: the sign of g3 rides on the A operand
    __shared__ __attribute__((aligned(16))) unsigned short sgn16[2][TILE];  // 0x8000 where g3[row] < 0: two rows = one packed-pair sign word
    __shared__ __attribute__((aligned(16))) f16 smimg[2][TILE * LDZ];  // (40 KB: reused for the extra pieces' partial sums)
    __shared__ __attribute__((aligned(16))) f16x8 fq[2][TILE * 2];      // scaled input fragments [row][lane half] of the states
    __shared__ __attribute__((aligned(16))) f16x8 fa[2][TILE * 2];      // ... of the action (critic)
    __shared__ float gred[8];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int set = blockIdx.x % p.n_sets, j0 = blockIdx.x / p.n_sets, J = gridDim.x / p.n_sets, P = p.n_agents / p.n_sets;
    const float gsc = set_gscale(p.part_m, set, p.n_sets, J, gred);  // 2^kg
    const f16x8* wf1 = p.net.wf1h + (long)set * NGT_MAX * 64 + lane;
    // this wave's extra piece
    const int xt = NET::critic ? 8 + (w >> 2) : 8, xrh = NET::critic ? (w >> 1) & 1 : (w >> 2) & 1, xc0 = NET::critic ? 2 * (w & 1) : (w & 3);
    const f16x8 wf0 = wf1[64 * w], wfx = wf1[64 * xt];
    f32x16 G0[4], G1[XC];
    const f32x16 zero16 = {};
#pragma unroll
    for (int c = 0; c < 4; ++c) G0[c] = zero16;
#pragma unroll
    for (int c = 0; c < XC; ++c) G1[c] = zero16;

    // Staging. sm (335 MB per pass, from HBM): 64 rows x 8 chunks of 32 bytes, one chunk per thread, TWO tiles ahead in two
    // register sets used alternately -- one tile of lookahead (~2.5 us) does not cover the tail of the HBM latency over 512
    // threads and a barrier: measured, a third of the kernel was the wait for it (tools/fsplit_ablate.sh @ tag r06-pre-prune). The inputs
    // (states, g3, actions: 31 MB, L2 / Infinity-Cache resident): one tile ahead, threads 0..127 build the |g3|-scaled fragment
    // of (row, lane half) = (tid >> 1, tid & 1) once per tile.
    const int srow = tid >> 3, sch = tid & 7, frow = tid >> 1, fh = tid & 1;
    uint4 a0 = {}, a1 = {}, b0 = {}, b1 = {};
    float sx[4] = {0.f, 0.f, 0.f, 0.f}, sg = 0.f, sa = 0.f;
    auto smsrc = [&](int pl) { return (const uint4*)(p.sm + ((long)(((FSPLIT_ABL(p.abl) & 1) ? j0 : pl) * p.n_sets + set) * TILE + srow) * H2 + 16 * sch); };
    auto fetch_x = [&](int pl) {
        if (pl >= P || tid >= 2 * TILE) return;
        const long ri = (long)(pl * p.n_sets + set) * TILE + frow;
        load_x<S>(p.x, ri, sx);
        sg = p.g3[ri];
        if (NET::critic) sa = p.act[ri];
    };
    auto stage = [&](int pl, int buf, const uint4& d0, const uint4& d1) {
        if (pl >= P) return;
        uint4* dst = (uint4*)(smimg[buf] + srow * LDZ + 16 * sch);
        dst[0] = d0, dst[1] = d1;
        if (tid < 2 * TILE) {
            const float g = fabsf(sg) * gsc;  // in [0, 1)
            fq[buf][tid] = make_xg(g * sx[0], g * sx[1], g * sx[2], g * sx[3], g, fh);
            if (NET::critic) fa[buf][tid] = make_xg(g * sa, 0.f, 0.f, 0.f, g, fh);
            if (SGN && fh == 0) sgn16[buf][frow] = sg < 0.f ? 0x8000 : 0;
        }
    };
    const int g4 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    // Three units per tile and wave -- own tile row half 0, row half 1, the extra piece -- as a software pipeline: the relu /
    // split VALU of unit k + 1 is issued INTO the gaps between the MFMAs of unit k (sched_group_barrier). Issued one unit after
    // the other, the two waves of a SIMD run their VALU phases together and their MFMA phases together (same program, one
    // barrier per tile) and neither pipe overlaps the other: measured 5400 cycles per tile and SIMD for 2750 of MFMA.
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    auto read_tr = [&](const f16* at) {  // transposed 16-bit read (the type of the builtin's element is immaterial: bit pattern)
        return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)at));
    };
    auto read_b = [&](int buf, int e, int s, f16x8 (&bfr)[4]) {  // sm fragments [column tile] of row half e, k-step s
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int R0 = 32 * e + 16 * s + 8 * hf + 4 * (g4 >> 1);
                const f16x4 t = read_tr(smimg[buf] + (R0 + q) * LDZ + 32 * c + 16 * (g4 & 1) + 4 * pp);
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[c][4 * hf + j] = t[j];
            }
    };
    // pair m of a unit's 16 rows = registers 2m, 2m + 1 = rows acc_row(2m, h), + 1 of row half e: sign word 16 e + 4 (m >> 1) + 2 h + (m & 1)
    auto split16e = [&](const f32x16& p1, unsigned (&qh)[8], unsigned (&ql)[8], int buf, int e) {
        if (FSPLIT_ABL(p.abl) & 4) {  // (timing ablation, diagnostic build: no relu / split VALU: WRONG results)
#pragma unroll
            for (int m = 0; m < 8; ++m) qh[m] = __float_as_uint(p1[2 * m]), ql[m] = __float_as_uint(p1[2 * m + 1]);
            return;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) split2h(relu(p1[2 * m]), relu(p1[2 * m + 1]), qh[m], ql[m]);
        if (SGN) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const uint2 sw = *(const uint2*)((const unsigned*)sgn16[buf] + 16 * e + 4 * qd + 2 * h);
                qh[2 * qd] ^= sw.x, ql[2 * qd] ^= sw.x, qh[2 * qd + 1] ^= sw.y, ql[2 * qd + 1] ^= sw.y;
            }
        }
    };
    auto compute = [&](int buf) {
        unsigned ah[8], al[8], bh[8], bl[8];
        // stage 0: first layer + split of row half 0 (exposed)
        split16e(mfmah(fq[buf][r * 2 + h], wf0, zero16), ah, al, buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        // stage 1: MFMAs of row half 0 | first layer + split of row half 1
        {
            const f32x16 p1 = mfmah(fq[buf][(32 + r) * 2 + h], wf0, zero16);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 bfr[4];
                read_b(buf, 0, s, bfr);
                const f16x8 hi = fragh(ah[4 * s], ah[4 * s + 1], ah[4 * s + 2], ah[4 * s + 3]);
                const f16x8 lo = fragh(al[4 * s], al[4 * s + 1], al[4 * s + 2], al[4 * s + 3]);
#pragma unroll
                for (int c = 0; c < 4; ++c) G0[c] = mfmah(hi, bfr[c], G0[c]), G0[c] = mfmah(lo, bfr[c], G0[c]);
            }
            split16e(p1, bh, bl, buf, 1);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, SGN ? 4 : 3, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // stage 2: MFMAs of row half 1 | first layer + split of the extra piece (input fragment of ITS row half)
        {
            const f16x8 xin = NET::critic ? fa[buf][(32 * xrh + r) * 2 + h] : fq[buf][(32 * xrh + r) * 2 + h];
            const f32x16 px = mfmah(xin, wfx, zero16);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 bfr[4];
                read_b(buf, 1, s, bfr);
                const f16x8 hi = fragh(bh[4 * s], bh[4 * s + 1], bh[4 * s + 2], bh[4 * s + 3]);
                const f16x8 lo = fragh(bl[4 * s], bl[4 * s + 1], bl[4 * s + 2], bl[4 * s + 3]);
#pragma unroll
                for (int c = 0; c < 4; ++c) G0[c] = mfmah(hi, bfr[c], G0[c]), G0[c] = mfmah(lo, bfr[c], G0[c]);
            }
            split16e(px, ah, al, buf, xrh);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, SGN ? 4 : 3, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // stage 3: MFMAs of the extra piece: column tile(s) xc0.. of row half xrh (runtime: the fragments are read at their
        // address, not selected from registers)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f16x8 hi = fragh(ah[4 * s], ah[4 * s + 1], ah[4 * s + 2], ah[4 * s + 3]);
            const f16x8 lo = fragh(al[4 * s], al[4 * s + 1], al[4 * s + 2], al[4 * s + 3]);
#pragma unroll
            for (int c = 0; c < XC; ++c) {
                f16x8 bx;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int R0 = 32 * xrh + 16 * s + 8 * hf + 4 * (g4 >> 1);
                    const f16x4 t = read_tr(smimg[buf] + (R0 + q) * LDZ + 32 * (xc0 + c) + 16 * (g4 & 1) + 4 * pp);
#pragma unroll
                    for (int j = 0; j < 4; ++j) bx[4 * hf + j] = t[j];
                }
                G1[c] = mfmah(hi, bx, G1[c]);
                G1[c] = mfmah(lo, bx, G1[c]);
            }
        }
    };
    if (j0 < P) a0 = smsrc(j0)[0], a1 = smsrc(j0)[1];
    fetch_x(j0);
    if (j0 + J < P) b0 = smsrc(j0 + J)[0], b1 = smsrc(j0 + J)[1];
    stage(j0, 0, a0, a1);
    __syncthreads();
#ifdef AVD_STAMP
    unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; }
#else
#define STAMP(i)
#endif
    for (int pi = j0; pi < P; pi += 2 * J) {
        // tile pi from buffer 0; set a takes tile pi + 2J; set b (tile pi + J) goes to buffer 1
        fetch_x(pi + J);
        if (pi + 2 * J < P) a0 = smsrc(pi + 2 * J)[0], a1 = smsrc(pi + 2 * J)[1];
        STAMP(0);
        compute(0);
        STAMP(1);
        stage(pi + J, 1, b0, b1);
        STAMP(2);
        if (!(FSPLIT_ABL(p.abl) & 2)) __syncthreads();
        STAMP(3);
        if (pi + J < P) {
            // tile pi + J from buffer 1; set b takes tile pi + 3J; set a (tile pi + 2J) goes to buffer 0
            fetch_x(pi + 2 * J);
            if (pi + 3 * J < P) b0 = smsrc(pi + 3 * J)[0], b1 = smsrc(pi + 3 * J)[1];
            STAMP(0);
            compute(1);
            STAMP(1);
            stage(pi + 2 * J, 0, a0, a1);
            STAMP(2);
            if (!(FSPLIT_ABL(p.abl) & 2)) __syncthreads();
            STAMP(3);
        }
    }
#ifdef AVD_STAMP
    if (p.stamp && blockIdx.x == 16 && lane == 0)
        for (int i = 0; i < 4; ++i) p.stamp[w * 8 + i] = tacc[i];
#endif
    // the extra pieces: row-half partials summed through LDS (the sm images are dead now), fixed order: row half 0 + row half 1
    const float dsc = 1.f / (gsc * S1);  // (G holds 2^kg S1 G: exact powers of two)
    float* comb = (float*)&smimg[0][0];  // [4 pieces of row half 1][XC][16 registers][64 lanes] <= 32 KB
    const int cidx = NET::critic ? 2 * (w >> 2) + (w & 1) : (w & 3);  // the same for a piece's two row-half waves
    if (xrh == 1) {
#pragma unroll
        for (int c = 0; c < XC; ++c)
#pragma unroll
            for (int k = 0; k < 16; ++k) comb[((cidx * XC + c) * 16 + k) * 64 + lane] = G1[c][k];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float* dst = p.partG + ((long)blockIdx.x * KG + 32 * w) * H2 + 32 * c + r;
#pragma unroll
        for (int k = 0; k < 16; ++k) dst[(long)acc_row(k, h) * H2] = G0[c][k] * dsc;
    }
    if (xrh == 0) {
#pragma unroll
        for (int c = 0; c < XC; ++c) {
            float* dst = p.partG + ((long)blockIdx.x * KG + 32 * xt) * H2 + 32 * (xc0 + c) + r;
#pragma unroll
            for (int k = 0; k < 16; ++k) dst[(long)acc_row(k, h) * H2] = (G1[c][k] + comb[((cidx * XC + c) * 16 + k) * 64 + lane]) * dsc;
        }
    }
}

// ---- dx: dC = |g3| (sm . W2c^T), BN/ReLU backward of the first layer and its parameter sums -----------------------------------
struct DxArgs {
    NetP net;
    int n_agents, n_sets;
    const f16x8* xfh;  // packed state fragments, fp16 pairs (pack_x_kernel): the rows' first-layer input AND (lane half 0: [x_hi | x_lo])
                       // the [k][row] image of V's B operand. The first layer's SIGN is the relu mask: as bf16 pairs (2^-16) about
                       // 1e-5 of the pre-activations landed on the wrong side of zero, and one flipped row moves an entry of dW1 / db1
                       // by ~1e-3 of the tensor's max (r03, measured); fp16 pairs put z1 at the f32 level
    const float* act;  // dxa_kernel: the per-row action input [n_agents][64]
    int L_cWa, L_cba;  // dxa_kernel: offsets of the critic's action-layer weights / bias inside net.th
    const float* g3;
    const float* part_m;  // [grid][8] max |g3| per wave of the kernel that wrote g3 (set_gscale)
    const f16* sm;
    int unsigned_mask;  // actor: sm holds 1 / 0 (OUT_TANH_SAVE), the row factor keeps the sign of g3; critic: sm holds sign(g3) inside
    float* partV;      // [grid][KP][16]       sum_rows (dC * mask) * [x_hi | x_lo | 1] per feature
    int abl;           // diagnostic build only (see DwArgs)
    unsigned long long* stamp;  // diagnostic build (-DAVD_STAMP) only
};
// Wave w = state feature tile w (8 tiles), both row halves. Resident: fp16 hi and lo of the tile's rows of SWC W2c as B fragments
// (reduction over the 128 columns; A = the sm rows, exact). dC comes out [row][feature] (feature on the lane) like the
// recomputed first layer: the per-feature sums over rows are per-lane sums over the registers, and the masked gradient tile,
// scaled by vs = 2^kg SWC 2^-5 (|dC_s| <= 128 x 2^13, |g3| 2^kg < 1: below 2^15, cannot overflow) and split into an fp16 pair, is
// the A operand of V = (dC * mask)^T . [x_hi | x_lo | 1] -> dW1, db1 (k order permuted as in dw_kernel; the [k][row] image of the
// inputs is staged once per tile by waves 0 and 1). r04: every operand of V an fp16 pair (were bf16 pairs: 2^-17 per row, which in
// a noise-dominated sum over rows IS the relative error of the result -- 2e-5 on cWa at 4096 x 10).
//
// (r04: the critic's 48 action features were tried INSIDE this kernel, as one extra 32 x 32 unit per wave every second tile from
// the sm image in LDS -- correct, and slower than the two kernels: 409 us against 229 + 91. One barrier per tile makes every tile
// wait for the four waves that carry a unit, the unit is a serial chain -- LDS reads, 16 dependent MFMAs, ~150 VALU -- that no
// other wave of the workgroup can overlap, and its 20 persistent registers came back as scratch reloads in the tile loop. They
// stay in dxa_kernel.)
template <int S, class NET>
__global__ __launch_bounds__(NT) void dx_kernel(const DxArgs p) {
    constexpr int KP = NET::KP, LDZ = 136;  // 272-byte rows: conflict-free b128 row reads
    constexpr float VSH = 1.f / 32.f;
    __shared__ __attribute__((aligned(16))) f16 smimg[2][TILE * LDZ];
    __shared__ __attribute__((aligned(16))) f16 xt[2][2][32 * 32];  // per buffer, per row half: [k column][row]
    __shared__ __attribute__((aligned(16))) float g3s[2][TILE];
    __shared__ float gred[8];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5, ft = w;
    const int set = blockIdx.x % p.n_sets, j0 = blockIdx.x / p.n_sets, J = gridDim.x / p.n_sets, P = p.n_agents / p.n_sets;
    const f32x16 zero16 = {};
    const float gsc = set_gscale(p.part_m, set, p.n_sets, J, gred);  // 2^kg
    f16x8 wch[8], wcl[8];
    {
        const long at = ((long)set * KP + 32 * ft + r) * H2 + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) wch[s] = *(const f16x8*)(p.net.Wchi + at + 16 * s), wcl[s] = *(const f16x8*)(p.net.Wclo + at + 16 * s);
    }
    // Everything behind dC runs in a SCALED domain: d_s = dC_s |g3| 2^kg / 32 = vs d with vs = 2^kg SWC / 32 (SWC: the power of two
    // on the fp16 W2c operand) -- |dC_s| <= 128 x 2^13 and |g3| 2^kg < 1 give |d_s| < 2^15: the fp16 pair of the masked gradient
    // cannot overflow -- and the per-feature sums are divided by vs once, at the end (powers of two: exact).
    const float swc = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p.net.vec[(long)set * VEC + 2 * H2 + 2])));
    const float gv = gsc * VSH, vs = gsc * swc * VSH;
    const f16x8 wf = p.net.wf1h[((long)set * NGT_MAX + ft) * 64 + lane];  // (scaled by S1: p1 = S1 z1, U1 rescaled at the end)
    f32x16 V = zero16;
    for (int i = tid; i < 2 * 2 * 32 * 32; i += NT) {  // columns 9.. stay zero, column 8 is the ones column (bias)
        const int k = (i >> 5) & 31;
        (&xt[0][0][0])[i] = (f16)(k == 8 ? 1.f : 0.f);
    }
    const int srow = tid >> 3, sch = tid & 7;
    uint4 d0 = {}, d1 = {};
    float gn = 0.f;
    auto fetch = [&](int agent) {
        if (FSPLIT_ABL(p.abl) & 1) agent = j0 * p.n_sets + set;
        const uint4* src = (const uint4*)(p.sm + ((long)agent * TILE + srow) * H2 + 16 * sch);
        d0 = src[0], d1 = src[1];
        if (tid < TILE) gn = p.g3[(long)agent * TILE + tid];
    };
    auto stage = [&](int buf) {
        uint4* dst = (uint4*)(smimg[buf] + srow * LDZ + 16 * sch);
        dst[0] = d0, dst[1] = d1;
        if (tid < TILE) g3s[buf][tid] = (p.unsigned_mask ? gn : fabsf(gn)) * gv;
    };
    f16x8 xfn0 = {}, xfn1 = {};  // the rows' input fragments [x_hi | x_lo] (h = 0) / [x_hi | 1 1 0 0] (h = 1), both row halves (two
                                 // variables, not an array: LLVM merges the two image copies below into one indexed by the wave number
                                 // and then keeps the array in scratch memory)
    auto fetch_x = [&](int agent) {
        xfn0 = p.xfh[((long)agent * TILE + r) * 2 + h];
        xfn1 = p.xfh[((long)agent * TILE + 32 + r) * 2 + h];
    };
    auto stage_x = [&](int buf) {  // [k][row] image of [x_hi | x_lo]: the h = 0 fragments, transposed (waves 0, 1: one row half each)
        if (w < 2 && h == 0) {
            f16x8 src = xfn0;
            if (w == 1) src = xfn1;
            asm volatile("" : "+v"(src));  // (a register copy selected by the scalar wave number, not a memory index)
#pragma unroll
            for (int k = 0; k < 8; ++k) xt[buf][w][k * 32 + r] = src[k];
        }
    };
    __syncthreads();  // the zero / ones fill above before the first stage_x
    if (j0 < P) fetch(j0 * p.n_sets + set), fetch_x(j0 * p.n_sets + set), stage(0), stage_x(0);
    __syncthreads();
#ifdef AVD_STAMP
    unsigned long long dacc[4] = {0, 0, 0, 0}, dlast = __builtin_amdgcn_s_memtime();
#define DXSTAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); dacc[i] += t_ - dlast; dlast = t_; }
#else
#define DXSTAMP(i)
#endif
    int buf = 0;
    for (int pi = j0; pi < P; pi += J, buf ^= 1) {
        const bool more = pi + J < P;
        const f16x8 xf[2] = {xfn0, xfn1};
        if (more) fetch((pi + J) * p.n_sets + set), fetch_x((pi + J) * p.n_sets + set);
        DXSTAMP(0);
        // The two row halves as a software pipeline: [dC of half 0] [dC of half 1 | BN/ReLU backward + split VALU of half 0]
        // [V of half 0 | VALU of half 1] [V of half 1]. One half after the other, the two waves of a SIMD run their MFMA phases
        // together and their VALU phases together (one barrier per tile) and the pipes never overlap (dw_kernel, measured).
        auto read_a = [&](int e, f16x8 (&smf)[8]) {
            const f16* arow = smimg[buf] + (32 * e + r) * LDZ + 8 * h;
#pragma unroll
            for (int s = 0; s < 8; ++s) smf[s] = *(const f16x8*)(arow + 16 * s);
        };
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        auto read_img = [&](const f16* img, f16x8 (&xb)[2]) {  // B operand of V from a [k][row] image
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x4 lo = *(const f16x4*)(img + r * 32 + 16 * s + 4 * h);
                const f16x4 hi = *(const f16x4*)(img + r * 32 + 16 * s + 8 + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) xb[s][j] = lo[j], xb[s][4 + j] = hi[j];
            }
        };
        // BN/ReLU backward of the first layer: only the MASKED gradient is needed per element. The two unmasked per-feature sums of
        // r03 -- sum_rows dC (-> d beta1) and sum_rows dC relu(z1) (-> d gamma1): 3 of this kernel's ~8 VALU per element, and the
        // kernel is VALU-bound at 40 % of the matrix pipe -- are linear images of sums that exist anyway (r04, finalize_feat_kernel):
        //     sum_rows dC[row][f]           = sum_n W2[f][n] db2[n]                         (backprop of the row sum through the layer)
        //     sum_rows dC[row][f] relu(z1)  = sum_k W1[k][f] V[f][k] + b1[f] V[f][8]        (relu(z1) = mask z1, z1 = x . W1 + b1)
        auto backward = [&](int e, const f32x16& dc, const f32x16& p1, unsigned (&vh)[8], unsigned (&vl)[8]) {
            if (FSPLIT_ABL(p.abl) & 4) {  // (timing ablation, diagnostic build: the BN / ReLU backward + split VALU skipped, raw bits as operands: WRONG results)
#pragma unroll
                for (int m = 0; m < 8; ++m) vh[m] = __float_as_uint(dc[2 * m]) ^ __float_as_uint(p1[2 * m]), vl[m] = __float_as_uint(dc[2 * m + 1]);
                return;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 gq = *(const float4*)(&g3s[buf][32 * e + 8 * g + 4 * h]);
                const float gg[4] = {gq.x, gq.y, gq.z, gq.w};
                float dm[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * g + j;
                    dm[j] = p1[k] > 0.f ? dc[k] * gg[j] : 0.f;
                }
                split2h(dm[0], dm[1], vh[2 * g], vl[2 * g]);
                split2h(dm[2], dm[3], vh[2 * g + 1], vl[2 * g + 1]);
            }
        };
        f16x8 smf[8], xb0[2], xb1[2];
        unsigned vh0[8], vl0[8], vh1[8], vl1[8];
        // stage 0: dC of row half 0
        read_a(0, smf);
        f32x16 dc0 = zero16;
#pragma unroll
        for (int s = 0; s < 8; ++s) dc0 = mfmah(smf[s], wch[s], dc0), dc0 = mfmah(smf[s], wcl[s], dc0);
        const f32x16 p10 = mfmah(xf[0], wf, zero16);
        __builtin_amdgcn_sched_barrier(0);
        // stage 1: dC of row half 1 | backward VALU of row half 0
        read_a(1, smf);
        read_img(xt[buf][0], xb0);
        f32x16 dc1 = zero16;
#pragma unroll
        for (int s = 0; s < 8; ++s) dc1 = mfmah(smf[s], wch[s], dc1), dc1 = mfmah(smf[s], wcl[s], dc1);
        const f32x16 p11 = mfmah(xf[1], wf, zero16);
        backward(0, dc0, p10, vh0, vl0);
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // stage 2: V of row half 0 | backward VALU of row half 1; then V of row half 1
        read_img(xt[buf][1], xb1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            V = mfmah(fragh(vh0[4 * s], vh0[4 * s + 1], vh0[4 * s + 2], vh0[4 * s + 3]), xb0[s], V);
            V = mfmah(fragh(vl0[4 * s], vl0[4 * s + 1], vl0[4 * s + 2], vl0[4 * s + 3]), xb0[s], V);
        }
        backward(1, dc1, p11, vh1, vl1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            V = mfmah(fragh(vh1[4 * s], vh1[4 * s + 1], vh1[4 * s + 2], vh1[4 * s + 3]), xb1[s], V);
            V = mfmah(fragh(vl1[4 * s], vl1[4 * s + 1], vl1[4 * s + 2], vl1[4 * s + 3]), xb1[s], V);
        }
        DXSTAMP(1);
        if (more) stage(buf ^ 1), stage_x(buf ^ 1);
        DXSTAMP(2);
        if (!(FSPLIT_ABL(p.abl) & 2)) __syncthreads();
        DXSTAMP(3);
    }
#ifdef AVD_STAMP
    if (p.stamp && blockIdx.x == 16 && lane == 0)
        for (int i = 0; i < 4; ++i) p.stamp[w * 8 + i] = dacc[i];
#endif
    const float ivs = 1.f / vs;  // (powers of two: exact)
    if (r < 16) {
        float* pv = p.partV + ((long)blockIdx.x * KP + 32 * ft) * 16 + r;
#pragma unroll
        for (int k = 0; k < 16; ++k) pv[(long)acc_row(k, h) * 16] = V[k] * ivs;
    }
}

// ---- dxa: the critic's ACTION feature tiles (48 features = tiles 8, 9) of dC and their parameter sums ----------------------
// A quarter of dx_kernel's matrix work per tile. Every wave is on its own: wave w = (rh, ft, par) takes row half rh of feature tile
// 8 + ft of every second tile (parity par) of the workgroup and fetches its 32 sm rows (8 KB, contiguous) itself, TWO of its
// tiles ahead, as eight COALESCED 1-KiB pieces (16 lanes per 256-byte row) that go through a wave-private LDS image (272-byte
// rows) into the A-fragment layout -- no workgroup barrier anywhere in the loop. (r03 / early r04: the fragments straight from
// global memory, i.e. one 16-byte piece of 32 different rows per load instruction: 91-93 us whatever the prefetch depth. Also
// tried in r04: the tiles in pairs through a shared image, one barrier per pair like dx_kernel -- 118 us: with one pair of
// lookahead every barrier interval pays the HBM latency; and as extra units INSIDE dx_kernel -- 409 us against 229 + 91.)
// The action layer has ONE input (agent/model.py:68-71), so everything behind dC is f32 VALU work, no second product (r03: a
// first-layer MFMA, a [k][row] LDS image per wave, four more MFMAs and their operand splits per unit): p1 = relu(a wa + ba) as
// one fma + max -- the relu mask from an exact f32 pre-activation --, dWa = sum (dC mask) a, dba = sum (dC mask). The two
// parities and the two row halves are combined once, at the end, in a fixed order.
template <int S>
__global__ __launch_bounds__(NT) void dxa_kernel(const DxArgs p) {
    typedef CriticS NET;
    constexpr int KP = NET::KP;
    constexpr float VSH = 1.f / 32.f;
    constexpr int LDZ = 136;
    __shared__ __attribute__((aligned(16))) f16 simg[8][32 * LDZ];     // per wave: its unit's 32 sm rows
    __shared__ __attribute__((aligned(16))) float ga[8][2][2][32];  // per wave, per buffer: |g3| gv and a of its 32 rows
    __shared__ float comb[8][64][2];
    __shared__ float gred[8];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int rh = w & 1, ftl = (w >> 1) & 1, par = w >> 2;
    const int set = blockIdx.x % p.n_sets, j0 = blockIdx.x / p.n_sets, J = gridDim.x / p.n_sets, P = p.n_agents / p.n_sets;
    const f32x16 zero16 = {};
    const int ft = 8 + ftl, f = 32 * ftl + r;  // this lane's action feature (f >= 48: padding, zero weights)
    const float gsc = set_gscale(p.part_m, set, p.n_sets, J, gred);
    f16x8 wch[8], wcl[8];
    {
        const long at = ((long)set * KP + 32 * ft + r) * H2 + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) wch[s] = *(const f16x8*)(p.net.Wchi + at + 16 * s), wcl[s] = *(const f16x8*)(p.net.Wclo + at + 16 * s);
    }
    const float swc = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p.net.vec[(long)set * VEC + 2 * H2 + 2])));
    const float gv = gsc * VSH, vs = gsc * swc * VSH;  // the scaled domain of dx_kernel (nothing here needs it for range: one scale for both kernels)
    const float* th = p.net.th + (long)set * p.net.th_stride;
    const float wa = f < HA ? th[p.L_cWa + f] : 0.f, ba = f < HA ? th[p.L_cba + f] : 0.f;
    float Sa = 0.f, Sb = 0.f;
    const int ntile = j0 < P ? (P - j0 + J - 1) / J : 0;
    // two register sets of prefetched operands, used alternately (k = par, par + 2, ..). (Plain variables through macros: as arrays
    // handed to lambdas by reference the two sets ended up in scratch memory and -- promoted by the compiler -- in 64 KB of LDS.)
    uint4 A0 = {}, A1 = {}, A2 = {}, A3 = {}, A4 = {}, A5 = {}, A6 = {}, A7 = {}, B0 = {}, B1 = {}, B2 = {}, B3 = {}, B4 = {}, B5 = {}, B6 = {}, B7 = {};
    float gA = 0.f, aA = 0.f, gB = 0.f, aB = 0.f;
#define DXA_FETCH(k_, X, g_, a_)                                                                                    \
    {                                                                                                               \
        const long r0_ = (long)((j0 + (k_) * J) * p.n_sets + set) * TILE + 32 * rh;                                 \
        const uint4* src_ = (const uint4*)(p.sm + r0_ * H2) + lane; /* piece i = rows 4 i .. 4 i + 3: lane = 16 (row & 3) + chunk */ \
        X##0 = src_[0], X##1 = src_[64], X##2 = src_[128], X##3 = src_[192], X##4 = src_[256], X##5 = src_[320], X##6 = src_[384], X##7 = src_[448]; \
        g_ = p.g3[r0_ + r], a_ = p.act[r0_ + r];                                                                    \
    }
#define DXA_PUT(buf_, X, g_, a_)                                                                                    \
    {                                                                                                               \
        if (h == 0) ga[w][buf_][0][r] = fabsf(g_) * gv, ga[w][buf_][1][r] = a_;                                     \
        f16* dst_ = simg[w] + (lane >> 4) * LDZ + 8 * (lane & 15);                                                   \
        *(uint4*)(dst_) = X##0, *(uint4*)(dst_ + 4 * LDZ) = X##1, *(uint4*)(dst_ + 8 * LDZ) = X##2, *(uint4*)(dst_ + 12 * LDZ) = X##3; \
        *(uint4*)(dst_ + 16 * LDZ) = X##4, *(uint4*)(dst_ + 20 * LDZ) = X##5, *(uint4*)(dst_ + 24 * LDZ) = X##6, *(uint4*)(dst_ + 28 * LDZ) = X##7; \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                      \
        __builtin_amdgcn_wave_barrier();                                                                            \
    }
    auto unit = [&](int buf) {
        f32x16 dc = zero16;  // [row][feature]: feature on the lane
        const f16* arow = simg[w] + r * LDZ + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const f16x8 af = *(const f16x8*)(arow + 16 * s);
            dc = mfmah(af, wch[s], dc), dc = mfmah(af, wcl[s], dc);
        }
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 gq = *(const float4*)(&ga[w][buf][0][8 * q + 4 * h]), aq = *(const float4*)(&ga[w][buf][1][8 * q + 4 * h]);
            const float gg[4] = {gq.x, gq.y, gq.z, gq.w}, av[4] = {aq.x, aq.y, aq.z, aq.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m = fmaf(av[j], wa, ba) > 0.f ? dc[4 * q + j] * gg[j] : 0.f;  // (the unmasked sums: finalize_feat_kernel, see dx_kernel)
                sa = fmaf(m, av[j], sa);
                sb += m;
            }
        }
        Sa += sa, Sb += sb;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the image is rewritten by this wave's next unit)
        __builtin_amdgcn_wave_barrier();
    };
    if (par < ntile) DXA_FETCH(par, A, gA, aA);
    if (par + 2 < ntile) DXA_FETCH(par + 2, B, gB, aB);
    for (int k = par; k < ntile; k += 4) {
        DXA_PUT(0, A, gA, aA);  // (the set's registers are free again: its next fetch goes out before the unit's arithmetic)
        if (k + 4 < ntile) DXA_FETCH(k + 4, A, gA, aA);
        unit(0);
        if (k + 2 < ntile) {
            DXA_PUT(1, B, gB, aB);
            if (k + 6 < ntile) DXA_FETCH(k + 6, B, gB, aB);
            unit(1);
        }
    }
#undef DXA_FETCH
#undef DXA_PUT
    // combine: the two lane halves (16 rows each) and the four (rh, par) waves of a feature tile, fixed order; one writer per feature
    comb[w][lane][0] = Sa, comb[w][lane][1] = Sb;
    __syncthreads();
    if (rh == 0 && par == 0 && h == 0) {
        const int ws[4] = {w, w + 1, w + 4, w + 5};  // (rh 0, par 0), (rh 1, par 0), (rh 0, par 1), (rh 1, par 1)
        float t[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c) t[c] += comb[ws[i]][r][c] + comb[ws[i]][32 + r][c];
        const float ivs = 1.f / vs;  // (powers of two: exact)
        float* pv = p.partV + ((long)blockIdx.x * KP + 32 * ft + r) * 16;  // [feature][16]: 0 = sum (dC mask) a, 4 = its lo part (none), 8 = sum (dC mask)
#pragma unroll
        for (int c = 0; c < 9; ++c) pv[c] = c == 0 ? t[0] * ivs : (c == 8 ? t[1] * ivs : 0.f);
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct Plan {
    int grid, J;
    size_t Whi[4], Wlo[4], Wchi[2], Wclo[2], wf1h[4], vec[4], wap, a2, y, mu, dmu, g3, sm, sma, tz, t1p, s2raw, xfs, xfs2, partHs[3], partM[2],
        partV[2], partG[2], bad, total;
};
static Plan make_plan(int n_agents, int n_sets) {
    Plan pl;
    const int P = n_agents / n_sets;
    int J = cu_count() / n_sets;  // one 512-thread workgroup per CU, every workgroup bound to one set
    if (const char* e = AVD_DIAG_ENV("FSPLIT_J")) J = atoi(e);  // diagnostics
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
        const int K = (i & 1) ? CriticS::K : ActorS::K, KP = (i & 1) ? CriticS::KP : ActorS::KP;
        pl.Whi[i] = take(sizeof(f16) * (size_t)n_sets * H2 * K), pl.Wlo[i] = take(sizeof(f16) * (size_t)n_sets * H2 * K);
        pl.vec[i] = take(sizeof(float) * (size_t)n_sets * VEC);
        pl.wf1h[i] = take(16 * (size_t)n_sets * NGT_MAX * 64);
        if (i < 2) pl.Wchi[i] = take(sizeof(f16) * (size_t)n_sets * KP * H2), pl.Wclo[i] = take(sizeof(f16) * (size_t)n_sets * KP * H2);
    }
    pl.wap = take(4 * (size_t)n_sets * 48);
    const size_t rows = (size_t)n_agents * TILE;
    pl.a2 = take(4 * rows), pl.y = take(4 * rows), pl.mu = take(4 * rows), pl.dmu = take(4 * rows), pl.g3 = take(4 * rows);
    pl.sm = take(sizeof(f16) * rows * H2);   // the critic's signed masks (HEAD_BOTH)
    pl.sma = take(sizeof(f16) * rows * H2);  // the actor's unsigned masks (OUT_TANH_SAVE: written before the critic's are used)
    pl.tz = take(4 * rows);
    pl.t1p = take(2 * 4 * (size_t)n_sets * CriticS::K * H2), pl.s2raw = take(2 * 4 * (size_t)n_sets * H2);  // (per net)
    pl.xfs = take(32 * rows), pl.xfs2 = take(32 * rows);
    for (int i = 0; i < 2; ++i) {
        const int KP = i ? CriticS::KP : ActorS::KP, KG = i ? CriticS::KG : ActorS::KG;
        pl.partM[i] = take(4 * (size_t)pl.grid * 8);
        pl.partV[i] = take(4 * (size_t)pl.grid * KP * 16);
        pl.partG[i] = take(4 * (size_t)pl.grid * KG * H2);
    }
    for (int i = 0; i < 3; ++i) pl.partHs[i] = take(4 * (size_t)pl.grid * 8 * 2);
    pl.bad = take(sizeof(int) + sizeof(unsigned) * 4 * (size_t)n_sets * 2);  // the flag, then scale_kernel's maxima: one memset
    pl.total = o;
    return pl;
}
static int check_shape(const avd_mlp_layout* L, int n_agents, int n_sets, const char* who) {
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

// The chain in two phases over one workspace (avd_learn_set_split_critic / _actor; avd_learn_set_split_f16x3 = both):
//   CRITIC: operand preparation, targets, mu, critic loss + gradients + the action gradient d q / d mu, finalize of the critic block
//   ACTOR : actor gradients from the d mu the critic phase left in the workspace, finalize of the actor block
// so that a multi-GPU caller can put the critic block's all-reduce on a side stream while the actor phase still runs
// (avddpg_amd/trainer.py; workers/trainer.py:400-431 averages the two gradient lists independently).
enum Phase { PH_CRITIC = 1, PH_ACTOR = 2, PH_BOTH = 3 };
template <int S>
static int run(int phases, const avd_mlp_layout& L, int n_agents, int n_sets, const float* theta, const float* stats, const float* theta_t,
               const float* stats_t, const float* s, const float* a, const float* r, const float* s2, const float* aw, float gamma,
               float high, float* grads, float* losses, unsigned char* ws, const Plan& pl, hipStream_t st) {
    PrepArgs pa;
    pa.L = L, pa.S = S, pa.theta = theta, pa.stats = stats, pa.theta_t = theta_t, pa.stats_t = stats_t;
    pa.wap = (unsigned*)(ws + pl.wap), pa.bad = (int*)(ws + pl.bad), pa.smax = (unsigned*)(ws + pl.bad + sizeof(int));
    NetP net[4];
    for (int i = 0; i < 4; ++i) {
        const bool critic = i & 1, target = i >= 2;
        pa.Whi[i] = (f16*)(ws + pl.Whi[i]), pa.Wlo[i] = (f16*)(ws + pl.Wlo[i]), pa.vec[i] = (float*)(ws + pl.vec[i]);
        pa.wf1h[i] = (f16x8*)(ws + pl.wf1h[i]);
        pa.Wchi[i] = i < 2 ? (f16*)(ws + pl.Wchi[i]) : nullptr, pa.Wclo[i] = i < 2 ? (f16*)(ws + pl.Wclo[i]) : nullptr;
        NetP& n = net[i];
        n.th = (target ? theta_t : theta) + (critic ? L.actor_size : 0), n.th_stride = L.theta_size;
        n.Whi = pa.Whi[i], n.Wlo = pa.Wlo[i], n.Wchi = pa.Wchi[i], n.Wclo = pa.Wclo[i], n.wf1h = pa.wf1h[i], n.vec = pa.vec[i];
        n.wap = pa.wap;
    }
    const int P = n_agents / n_sets;
    const float inv_n = 1.0f / ((float)P * TILE);
    const long nrows = (long)n_agents * TILE;
    f16x8 *xfs = (f16x8*)(ws + pl.xfs), *xfs2 = (f16x8*)(ws + pl.xfs2);
    int* bad = (int*)(ws + pl.bad);
    float *a2 = (float*)(ws + pl.a2), *y = (float*)(ws + pl.y), *mu = (float*)(ws + pl.mu), *dmu = (float*)(ws + pl.dmu);
    float* g3 = (float*)(ws + pl.g3);
    f16 *sm = (f16*)(ws + pl.sm), *sma = (f16*)(ws + pl.sma);
    auto F = [&](size_t off) { return (float*)(ws + off); };
    const dim3 grid(pl.grid), block(NT);
    HeadArgs h;
    h.n_agents = n_agents, h.n_sets = n_sets, h.gamma = gamma, h.high = high, h.inv_n = inv_n, h.aw = aw, h.sm = sm, h.g3 = g3, h.dmu = dmu;
    h.act2 = nullptr, h.part_s2 = nullptr, h.stamp = nullptr, h.bad = bad, h.part_m = nullptr, h.tz = F(pl.tz), h.abl = 0;
    if (const char* e = AVD_DIAG_ENV("FSPLIT_ABL")) h.abl = atoi(e);
    auto head = [&](auto kern, int ni, const f16x8* x, const float* act, const float* rr, const float* yin, float* out, float* part_s) {
        h.net = net[ni], h.xf = x, h.act = act, h.r = rr, h.yin = yin, h.out = out, h.part_s = part_s;
        hipLaunchKernelGGL(kern, grid, block, 0, st, h);
    };
    DwArgs dw;
    dw.n_agents = n_agents, dw.n_sets = n_sets, dw.sm = sm, dw.g3 = g3, dw.x = s, dw.stamp = nullptr, dw.abl = 0;
    if (const char* e = AVD_DIAG_ENV("FSPLIT_ABL")) dw.abl = atoi(e);
#ifdef AVD_STAMP
    static unsigned long long* d_stamp = nullptr;
    if (!d_stamp) (void)hipMalloc(&d_stamp, 128 * 8);
    dw.stamp = d_stamp;
#endif
    DxArgs dx;
    dx.n_agents = n_agents, dx.n_sets = n_sets, dx.sm = sm, dx.g3 = g3, dx.xfh = xfs, dx.stamp = nullptr, dx.L_cWa = dx.L_cba = 0, dx.unsigned_mask = 0;
    dx.abl = dw.abl;
#ifdef AVD_STAMP
    dx.stamp = d_stamp;
#endif
    FinArgs fa;
    fa.L = L, fa.n_sets = n_sets, fa.J = pl.J, fa.S = S, fa.nrh = 1, fa.theta = theta, fa.stats = stats, fa.grads = grads, fa.losses = losses;
    fa.inv_n = inv_n, fa.partLa = F(pl.partHs[2]), fa.bad = bad;
    for (int i = 0; i < 2; ++i)
        fa.partH[i] = nullptr, fa.partHs[i] = F(pl.partHs[i]), fa.partU[i] = nullptr, fa.partV[i] = F(pl.partV[i]),
        fa.partG[i] = F(pl.partG[i]), fa.c3[i] = F(pl.vec[i]);
    fa.t1_from_g[0] = fa.t1_from_g[1] = 1, fa.t1p = F(pl.t1p), fa.s2raw = F(pl.s2raw);  // T1 of both nets: from the weight-gradient partials
    if (phases & PH_CRITIC) {
        if (hipMemsetAsync(ws + pl.bad, 0, sizeof(int) + sizeof(unsigned) * 4 * (size_t)n_sets * 2, st) != hipSuccess)
            return check_launch("avd_learn_set_split: memset");
        // (r05, measured and not kept: the preparation chain -- three small dependent launches over 1.5 MB of weights, ~30 us -- on a side
        //  stream beside the input packing, forked and joined by events: 1969-1978 us per learn against 1957-1961 serial on the same
        //  box; the fork / join costs more than the overlap of two sub-30-us stages returns)
        hipLaunchKernelGGL(scale_kernel, dim3(4, n_sets, SCALE_SLICES), dim3(256), 0, st, pa);
        hipLaunchKernelGGL(prep_kernel, dim3(H2, 4, n_sets), dim3(320), 0, st, pa);
        hipLaunchKernelGGL(prep1_kernel, dim3(NGT_MAX, 4, n_sets), dim3(64), 0, st, pa);
        PackArgs pk;
        pk.x[0] = s, pk.extra[0] = a, pk.outh[0] = xfs, pk.x[1] = s2, pk.extra[1] = r, pk.outh[1] = xfs2, pk.rows = nrows, pk.bad = bad;
        hipLaunchKernelGGL(pack_x_kernel<S>, dim3((unsigned)((2 * nrows + 255) / 256), 2), dim3(256), 0, st, pk);
        // 1-2: targets
        head(head_kernel<S, ActorS, OUT_TANH>, 2, xfs2, nullptr, nullptr, nullptr, a2, nullptr);
        head(head_kernel<S, CriticS, OUT_TD>, 3, xfs2, a2, r, nullptr, y, nullptr);
        // 3: mu -- and the actor's relu masks + tanh(z) for its backward pass (no second actor forward: r03's HEAD_ACTOR)
        h.sm = sma;
        head(head_kernel<S, ActorS, OUT_TANH_SAVE>, 0, xfs, nullptr, nullptr, nullptr, mu, nullptr);
        h.sm = sm;
        // 4-7: critic loss and gradients; critic(s, mu) and the action gradient ride in the same launch (HEAD_BOTH)
        h.act2 = mu, h.part_s2 = F(pl.partHs[2]), h.part_m = F(pl.partM[1]);
        head(head_kernel<S, CriticS, HEAD_BOTH>, 1, xfs, a, nullptr, y, nullptr, F(pl.partHs[1]));
        dw.net = net[1], dw.act = a, dw.partG = F(pl.partG[1]), dw.part_m = F(pl.partM[1]);
        hipLaunchKernelGGL((dw_kernel<S, CriticS>), grid, block, 0, st, dw);
        dx.net = net[1], dx.partV = F(pl.partV[1]), dx.act = a, dx.part_m = F(pl.partM[1]);
        hipLaunchKernelGGL((dx_kernel<S, CriticS>), grid, block, 0, st, dx);
        dx.L_cWa = L.cWa, dx.L_cba = L.cba;
        hipLaunchKernelGGL((dxa_kernel<S>), grid, block, 0, st, dx);
        // 8: the critic block of the slab (+ both losses: the actor loss is the mean of q(s, mu), summed by HEAD_BOTH) -- when the
        // caller asked for this phase alone; the single call finalizes both blocks together at its end (three launches, not six)
        if (phases == PH_CRITIC) launch_finalize(fa, st, 1, 1);
    }
    if (phases & PH_ACTOR) {
        // 9-11: actor gradients: the seed from d mu and the stored tanh(z), then dw / dx on the stored (unsigned) masks
        SeedArgs sd;
        sd.n_agents = n_agents, sd.n_sets = n_sets, sd.dmu = dmu, sd.tz = F(pl.tz), sd.high = high, sd.g3 = g3, sd.part_m = F(pl.partM[0]),
        sd.part_s = F(pl.partHs[0]);
        hipLaunchKernelGGL(actor_seed_kernel, grid, block, 0, st, sd);
        dw.sm = sma, dx.sm = sma;
        dw.net = net[0], dw.act = nullptr, dw.partG = F(pl.partG[0]), dw.part_m = F(pl.partM[0]);
        hipLaunchKernelGGL((dw_kernel<S, ActorS>), grid, block, 0, st, dw);
        dx.net = net[0], dx.partV = F(pl.partV[0]), dx.act = nullptr, dx.part_m = F(pl.partM[0]), dx.L_cWa = dx.L_cba = 0, dx.unsigned_mask = 1;
        hipLaunchKernelGGL((dx_kernel<S, ActorS>), grid, block, 0, st, dx);
        // 12: the actor block of the slab (the single call: both blocks)
        if (phases == PH_ACTOR) launch_finalize(fa, st, 0, 1);
        else launch_finalize(fa, st, 0, 2);
    }
    return check_launch("avd_learn_set_split");
}

}  // namespace fsplit
}  // namespace avd

using namespace avd;

// Wave-level v_mfma_f32_32x32x16_f16 instructions (32 768 FLOP each) the chain ISSUES per 64-row tile (one agent's batch), from
// the kernels' loop structure -- every product of two f32-class operands is three MFMAs on 16-bit pairs (two where one operand is
// an exact +-1 / 0 mask), first layers ride on the matrix cores too:
//   head, per 32-row unit: first layer NFT + second layer 3 x 4 column tiles x NKS k-steps
//     OUT_TANH / OUT_TANH_SAVE (actor) 8 + 192 = 200, OUT_TD (critic) 10 + 228 = 238 (r03's HEAD_ACTOR, 200 more, is gone),
//     HEAD_BOTH 8 + 192 (state part, once) + (2 + 36) (branch A) + (4 + 36) (branch B1) + 36 (B2: M) = 314
//   dw, per tile: 8 waves x (2 first-layer + 2 row halves x 2 k-steps x 4 column tiles x 2) + the extra pieces 8 x (1 + 2 x XC x 2)
//   dx, per tile: 8 waves x (2 x 16 (dC) + 2 (first layer) + 2 x 4 (V));  dxa, per tile: 4 units x 16
namespace avd { namespace fsplit {
constexpr long MFMA_PER_TILE = 2 * (200 + 238 + 200 + 314)            /* four head launches, two 32-row units per tile */
                               + 8 * (2 + 32 + 1 + 2 * 2 * 2)         /* dw critic */
                               + 8 * (2 + 32 + 1 + 2 * 1 * 2)         /* dw actor */
                               + 2 * 8 * (32 + 2 + 8)                 /* dx critic, actor */
                               + 4 * 16;                              /* dxa */
static_assert(MFMA_PER_TILE == 3296, "update DESIGN.md 3.4 and the count above together");
}}
extern "C" int avd_learn_set_split_mfma_count(const avd_mlp_layout* lay, int n_agents, int n_sets, unsigned long long* mfma_32x32x16) {
    int rc = fsplit::check_shape(lay, n_agents, n_sets, "avd_learn_set_split_mfma_count");
    if (rc) return rc;
    AVD_REQUIRE(mfma_32x32x16, "avd_learn_set_split_mfma_count: null pointer");
    *mfma_32x32x16 = (unsigned long long)fsplit::MFMA_PER_TILE * (unsigned long long)n_agents;
    return AVD_OK;
}

extern "C" int avd_learn_set_split_workspace(const avd_mlp_layout* lay, int n_agents, int n_sets, size_t* bytes) {
    int rc = fsplit::check_shape(lay, n_agents, n_sets, "avd_learn_set_split_workspace");
    if (rc) return rc;
    AVD_REQUIRE(bytes, "avd_learn_set_split_workspace: null pointer");
    *bytes = fsplit::make_plan(n_agents, n_sets).total;
    return AVD_OK;
}

static int split_entry(int phases, const char* who, const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                       const float* theta_t, const float* stats_t, const float* s, const float* a, const float* r, const float* s2,
                       const float* agent_weight, float gamma, float high, float* grads, float* losses, void* workspace,
                       size_t workspace_bytes, void* stream) {
    int rc = fsplit::check_shape(lay, n_agents, n_sets, who);
    if (rc) return rc;
    const bool cr = phases & fsplit::PH_CRITIC;
    AVD_REQUIRE(theta && stats && s && grads && workspace && (!cr || (theta_t && stats_t && a && r && s2)), "%s: null pointer", who);
    const fsplit::Plan pl = fsplit::make_plan(n_agents, n_sets);
    AVD_REQUIRE(workspace_bytes >= pl.total, "%s: workspace %zu B < %zu B", who, workspace_bytes, pl.total);
    // (padding floats of the slab are never written by finalize: keep them zero like every other gradient producer)
    if (cr && hipMemsetAsync(grads, 0, sizeof(float) * (size_t)n_sets * lay->theta_size, (hipStream_t)stream) != hipSuccess)
        return check_launch("avd_learn_set_split: hipMemsetAsync(grads)");
    if (lay->S == 4)
        return fsplit::run<4>(phases, *lay, n_agents, n_sets, theta, stats, theta_t, stats_t, s, a, r, s2, agent_weight, gamma, high, grads,
                              losses, (unsigned char*)workspace, pl, (hipStream_t)stream);
    return fsplit::run<3>(phases, *lay, n_agents, n_sets, theta, stats, theta_t, stats_t, s, a, r, s2, agent_weight, gamma, high, grads, losses,
                          (unsigned char*)workspace, pl, (hipStream_t)stream);
}

extern "C" int avd_learn_set_split_f16x3(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                                          const float* theta_t, const float* stats_t, const float* s, const float* a, const float* r,
                                          const float* s2, const float* agent_weight, float gamma, float high, float* grads,
                                          float* losses, void* workspace, size_t workspace_bytes, void* stream) {
    return split_entry(fsplit::PH_BOTH, "avd_learn_set_split_f16x3", lay, n_agents, n_sets, theta, stats, theta_t, stats_t, s, a, r, s2,
                       agent_weight, gamma, high, grads, losses, workspace, workspace_bytes, stream);
}

// deprecated alias (r03's name: the operand pairs were bf16 then; every pair has been fp16 since r04)
extern "C" int avd_learn_set_split_bf16x3(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                                          const float* theta_t, const float* stats_t, const float* s, const float* a, const float* r,
                                          const float* s2, const float* agent_weight, float gamma, float high, float* grads,
                                          float* losses, void* workspace, size_t workspace_bytes, void* stream) {
    return avd_learn_set_split_f16x3(lay, n_agents, n_sets, theta, stats, theta_t, stats_t, s, a, r, s2, agent_weight, gamma, high, grads,
                                     losses, workspace, workspace_bytes, stream);
}

extern "C" int avd_learn_set_split_critic(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                                          const float* theta_t, const float* stats_t, const float* s, const float* a, const float* r,
                                          const float* s2, const float* agent_weight, float gamma, float high, float* grads,
                                          float* losses, void* workspace, size_t workspace_bytes, void* stream) {
    return split_entry(fsplit::PH_CRITIC, "avd_learn_set_split_critic", lay, n_agents, n_sets, theta, stats, theta_t, stats_t, s, a, r, s2,
                       agent_weight, gamma, high, grads, losses, workspace, workspace_bytes, stream);
}

extern "C" int avd_learn_set_split_actor(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                                         const float* s, float high, float* grads, void* workspace, size_t workspace_bytes, void* stream) {
    return split_entry(fsplit::PH_ACTOR, "avd_learn_set_split_actor", lay, n_agents, n_sets, theta, stats, nullptr, nullptr, s, nullptr, nullptr,
                       nullptr, nullptr, 0.f, high, grads, nullptr, workspace, workspace_bytes, stream);
}
