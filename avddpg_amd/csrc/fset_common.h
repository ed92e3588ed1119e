// Pieces shared by the two fused shared-weight-set learners: fset.hip (bf16 GEMM operands) and fsplit.hip (every GEMM operand
// an exact bf16 hi + lo pair, f32-class results). Reference: workers/trainer.py:472-508, src/server/federated.py:47-63.
#pragma once
#include "common.h"

namespace avd {
namespace fset {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float BN_EPS = 1e-3f;  // tf.keras BatchNormalization default epsilon (agent/model.py:28)
constexpr int TILE = 64, H1 = 256, H2 = 128, HA = 48, NT = 512, VEC = 264;

// K features enter the second layer; KP = K rounded up to feature tiles of 32 (dx operands); KW = K + 16: the head's weight
// image with the folded bias as feature K; NFT first-layer tiles; dw also accumulates a constant-one feature K (its G row is
// the column sum of dZ2 = the gradient of b2): NGT tiles, KG rows.
struct Actor {
    static constexpr int K = 256, KP = 256, KW = 272, NFT = 8, NGT = 9, KG = 288;
    static constexpr bool critic = false;
};
struct Critic {
    static constexpr int K = 304, KP = 320, KW = 320, NFT = 10, NGT = 10, KG = 320;  // 256 state + 48 action features
    static constexpr bool critic = true;
};

// row (M index) of accumulator register i of a 32x32 MFMA result in lane half h; the column is lane & 31
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// relu: ONE v_max_f32 when the file is compiled with -fno-honor-nans (Makefile); otherwise fmaxf is lowered to
// canonicalize + max (it must quiet signalling NaNs) -- two VALU instructions per element in loops whose VALU count bounds
// them. (Inline assembly is not an option: hipcc does not pad the MFMA -> VALU read hazard for an opaque instruction.)
__device__ __forceinline__ float relu(float x) { return fmaxf(x, 0.f); }
// relu + bf16 of a PAIR in two instructions: v_cvt_pk_bf16_f32, then v_pk_max_i16 against 0 -- a negative float is a negative
// int16 in its upper 16 bits, so the signed 16-bit max IS relu on the packed pair (-0.0 -> +0) -- instead of two v_max_f32 and
// the conversion
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_bf16x2(float a, float b) {
    const f32x2 f = {a, b};
    s16x2 v = __builtin_bit_cast(s16x2, __builtin_convertvector(f, bf16x2));
    const s16x2 z = {0, 0};
    v = __builtin_elementwise_max(v, z);
    return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}


// ---- finalize (fset.hip): sum the workgroups' partials of a set in a fixed order, apply the BN folds, write the slab ------
struct FinArgs {
    avd_mlp_layout L;
    int n_sets, J, S;  // J = workgroups per set; workgroup j of set m has block index j * n_sets + m
    int nrh;           // row-half slots in partU / partV (2: fset.hip, one per wave row half; 1: fsplit.hip)
    const float *theta, *stats;
    const float* partH[2];   // [0] actor (pass 9), [1] critic (pass 4)
    const float* partHs[2];
    const float* partLa;     // pass 7's part_s (actor loss sums)
    const float* partU[2];   // NULL (fsplit.hip since r04): d beta1 / d gamma1 are derived from partV, db2 and the weights (finalize_small_kernel)
    const float* partV[2];
    const float* partG[2];
    const float* c3[2];      // fsplit.hip: [n_sets][VEC] vectors whose [H2 + n] entry scales column n of partG (dZ2 = g3 c3[n] mask
                             // is carried as sign * mask, |g3| on the other operand); NULL: partG already holds P1^T dZ2
    const int* bad;          // device flag: non-zero when an input of the learn call was not finite -> the slab is set to NaN
    float* grads;   // [n_sets][theta_size]
    float* losses;  // [n_sets][2] or NULL
    float inv_n;
    int net_lo;     // first net of this launch (0 actor, 1 critic): set by launch_finalize
    int item_base;  // first item of this launch of finalize_small_kernel (items < H2: output columns; >= H2: first-layer features)
    // fsplit.hip (r04), per net: T1[n] = sum_rows g3 relu(z2)[n] (-> dW3, d gamma2) is NOT accumulated by a head kernel but derived from
    // the weight-gradient partials: relu(z2) = mask (P1 . W2' + b2'), so T1[n] = sum_f W2'[f][n] G[f][n] + b2'[n] S2[n] with
    // G = P1^T (g3 mask) (dw_kernel's raw sums), S2 = its constant-one row, W2' = inv1 (.) W2, b2' = vec[n]. finalize_w2_kernel leaves
    // the products in t1p [2 nets][n_sets][Critic::K][H2], finalize_small_kernel S2 in s2raw [2][n_sets][H2], finalize_t1_kernel adds them up.
    int t1_from_g[2];
    float* t1p;
    float* s2raw;
};
void launch_finalize(const FinArgs& fa, hipStream_t st, int net_lo = 0, int n_nets = 2);
int cu_count();  // CUs of the current device (cached per device ordinal)

// Non-finite test on the bits: these files are built with -fno-honor-nans, under which isnan() / isfinite() fold to constants.
// (the bits pass through an empty asm statement: LLVM otherwise recognises the mask-and-compare as an fpclass test of a float
// and, under no-nans-fp-math, folds it to false)
__device__ __forceinline__ bool not_finite(float x) {
    unsigned u = __float_as_uint(x);
    asm volatile("" : "+v"(u));
    return (u & 0x7f800000u) == 0x7f800000u;
}

}  // namespace fset
}  // namespace avd
