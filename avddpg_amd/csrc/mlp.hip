// Actor/critic MLP kernels for gfx950.
//
//  * mlp_rows_kernel  -- batch-1 actor / critic forward per agent (GEMV; weights streamed once).
//  * learn_kernel_t / learn_kernel_g -- Trainer.learn (workers/trainer.py:472-508) for one agent's batch of 64 rows
//                        per 256-thread workgroup: 5 forwards + 2 backwards fused, activations
//                        resident in LDS (~150 KB, one workgroup per CU), the [64xK]x[KxN] products
//                        on the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32), weights streamed
//                        from HBM/L2 straight into MFMA B operands, gradients written once.
//
// BatchNormalization is always the inference form y = p*inv + (beta - mean*inv),
// inv = rsqrt(var + 1e-3)*gamma (reference models are never called with training=True).  LDS holds
// the post-ReLU value p; the affine is applied when p is fetched as an MFMA operand, and folded into
// the weight-gradient epilogue (dW = inv*(p^T dz) + shift*(1^T dz)).
#include "learn_common.h"

namespace avd {

// ------------------------------------------------------------------------------------------
// General learn kernel: any widths that fit one workgroup's LDS, multi-action heads (centralized framework:
// S = 4L states, A = L actions, widths x1.2 zero-padded to MFMA multiples -- reference src/environment.py:35-52,
// agent/model.py hidd_mult, workers/trainer.py:102-108). Same four-pass plan and the same MFMA GEMM routines as
// the specialised kernel; the first/last layers (tiny K or N) are plain VALU loops. One LDS buffer fewer: the
// output-layer backward overwrites p with dz in place. Throughput is secondary here; correctness and generality
// are the point (the reference widths take learn_kernel_t).
// Zero padding is exact: a padded unit has zero weights/bias, beta = mean = 0, so it outputs 0, receives zero
// gradient for every parameter and stays zero under Adam.
// ------------------------------------------------------------------------------------------
namespace gen {

constexpr int MAX_A = 16, MAX_S = 64;

// Out-of-line helpers (one copy of each in the kernel: it runs them from up to five call sites and would not fit the
// instruction cache inlined) take their pointers with explicit address spaces -- a plain pointer argument is generic,
// and a generic access to LDS is a slow flat_load instead of ds_read.
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(1))) float glb_f;
#define LDSP(p) ((lds_f*)(p))
#define GLBP(p) ((glb_f*)(p))
#define CGLBP(p) ((const glb_f*)(p))

struct GLds {
    float *bufA, *bufB;          // [64][ldA] first-layer outputs (state | action), [64][ldB] second layer / its gradient
    float *invA, *shA;           // H1+Ha
    float *invB, *shB, *rsB, *mmB, *db;  // H2
    float *sX;                   // [64][S]
    float *sR;                   // [64]
    float *sAct, *sY, *sQ, *sD, *sA1, *sT, *sDa;  // [64][A]
    float* red;                  // [8]
};
__host__ __device__ inline size_t lds_floats(const avd_mlp_layout& L) {
    return (size_t)TILE * ld_of(L.H1 + L.Ha) + (size_t)TILE * ld_of(L.H2) + 2 * (L.H1 + L.Ha) + 5 * L.H2 +
           TILE * L.S + TILE + 7 * TILE * L.A + 8;
}
__device__ __forceinline__ GLds carve(float* p, const avd_mlp_layout& L) {
    GLds l;
    l.bufA = p, p += TILE * ld_of(L.H1 + L.Ha);
    l.bufB = p, p += TILE * ld_of(L.H2);
    l.invA = p, p += L.H1 + L.Ha;
    l.shA = p, p += L.H1 + L.Ha;
    l.invB = p, p += L.H2;
    l.shB = p, p += L.H2;
    l.rsB = p, p += L.H2;
    l.mmB = p, p += L.H2;
    l.db = p, p += L.H2;
    l.sX = p, p += TILE * L.S;
    l.sR = p, p += TILE;
    l.sAct = p, p += TILE * L.A;
    l.sY = p, p += TILE * L.A;
    l.sQ = p, p += TILE * L.A;
    l.sD = p, p += TILE * L.A;
    l.sA1 = p, p += TILE * L.A;
    l.sT = p, p += TILE * L.A;
    l.sDa = p, p += TILE * L.A;
    l.red = p;
    return l;
}

// first layer of a branch: out[r][col0+k] = relu(sum_j X[r*K+j]*W[j*H+k] + b[k]) and the BN coefficients of column k,
// on the matrix cores: 16 x 16 output tiles (row tile, column tile) dealt over the waves, the K <= 64 inputs in MFMA steps
// of 4 (indices past K are clamped and their operands zeroed). A tile's weight operands are all requested up front.
// ST = compile-time bound on the MFMA steps (K <= 4 ST): with 16 (any K <= 64) a K = 20 layer -- the centralized framework's state
// input at L = 5 -- issued 16 weight loads and 16 LDS reads per tile for 5 MFMAs (r04: ST = 5 and ST = 2 copies).
template <int ST>
__device__ __attribute__((noinline)) void l1_fwd_t(const lds_f* X, int K, const glb_f* __restrict__ W, const glb_f* __restrict__ b,
                                       const glb_f* __restrict__ g, const glb_f* __restrict__ be,
                                       const glb_f* __restrict__ mm, const glb_f* __restrict__ mv, int H, lds_f* out,
                                       int ld, int col0, lds_f* inv, lds_f* sh) {
    for (int k = threadIdx.x; k < H; k += NTHREADS) {
        const float iv = (1.0f / sqrtf(mv[k] + BN_EPS)) * g[k];
        inv[col0 + k] = iv;
        sh[col0 + k] = be[k] - mm[k] * iv;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    const int steps = (K + 3) >> 2;
    // a wave's items are the column tiles t = wave, wave + 4, ..; each serves the four row tiles with ONE set of weight
    // operands, and the next column tile's are requested before the current one is used
    const int ctiles = H >> 4;
    float wn[ST], bn_ = 0.f;
    auto load_w = [&](int t, float(&w)[ST], float& bc) {
        const int col = 16 * min(t, ctiles - 1) + lr;
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            const int j = 4 * st + lg;
            w[st] = W[min(j, K - 1) * H + col] * (j < K ? 1.f : 0.f);
        }
        bc = b[col];
    };
    load_w(wave, wn, bn_);
    for (int t = wave; t < ctiles; t += 4) {
        float wv[ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) wv[st] = wn[st];
        const float bc = bn_;
        load_w(t + 4, wn, bn_);
        const int col = 16 * t + lr;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const lds_f* xr = X + (16 * m + lr) * K;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < ST; ++st)
                if (st < steps) acc = MFMA16(xr[min(4 * st + lg, K - 1)], wv[st], acc);  // (clamped x meets a zero weight)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) out[(16 * m + 4 * lg + reg) * ld + col0 + col] = fmaxf(acc[reg] + bc, 0.f);
        }
    }
}

__device__ __forceinline__ void l1_fwd(const lds_f* X, int K, const glb_f* __restrict__ W, const glb_f* __restrict__ b,
                                       const glb_f* __restrict__ g, const glb_f* __restrict__ be, const glb_f* __restrict__ mm,
                                       const glb_f* __restrict__ mv, int H, lds_f* out, int ld, int col0, lds_f* inv, lds_f* sh) {
    if (K <= 8) l1_fwd_t<2>(X, K, W, b, g, be, mm, mv, H, out, ld, col0, inv, sh);
    else if (K <= 20) l1_fwd_t<5>(X, K, W, b, g, be, mm, mv, H, out, ld, col0, inv, sh);
    else l1_fwd_t<16>(X, K, W, b, g, be, mm, mv, H, out, ld, col0, inv, sh);
}

// BN coefficient tables of the layer in front of the output layer
__device__ __forceinline__ void coefs_b(const float* __restrict__ g, const float* __restrict__ be,
                                        const float* __restrict__ mm, const float* __restrict__ mv, int H2, GLds& l) {
    for (int k = threadIdx.x; k < H2; k += NTHREADS) {
        const float rs = 1.0f / sqrtf(mv[k] + BN_EPS);
        const float iv = rs * g[k];
        l.invB[k] = iv, l.shB[k] = be[k] - mm[k] * iv, l.rsB[k] = rs, l.mmB[k] = mm[k];
    }
}

// Narrow GEMM on the matrix cores: out[r][a] = sum_k x(r, k) * W[k * wk + a * wa] (+ bias[a]) for a < A <= 16, K % 16 == 0,
// x = X[r * ldx + k] * inv[k] + sh[k] (inv == nullptr: x = X). Used for the A-wide output layers (K = H2: W3[k][a]) and for
// the gradient w.r.t. the actions (K = Ha: Wa[a][k]). Wave m owns batch rows 16m .. 16m+15; the A columns sit in a
// 16-wide MFMA tile whose unused columns are fed zeros. All weight loads of the lane are issued up front with clamped
// indices (K <= 256: at most 64) -- 64 rows x 160 x 5 as a scalar loop per output took ~29 k cycles, this takes ~3 k.
__device__ __attribute__((noinline)) void narrow_gemm(const lds_f* X, int ldx, const lds_f* inv, const lds_f* sh, int K,
                                                       const glb_f* __restrict__ W, int wk, int wa, int A,
                                                       const glb_f* __restrict__ bias, lds_f* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    const int nblk = K >> 4;
    const float colmask = (lr < A) ? 1.f : 0.f;
    const int ac = min(lr, A - 1);
    float wv[16][4];
#pragma unroll
    for (int blk = 0; blk < 16; ++blk)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) wv[blk][jj] = W[min(16 * blk + 4 * lg + jj, K - 1) * wk + ac * wa] * colmask;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const lds_f* xr = X + (wave * 16 + lr) * ldx + 4 * lg;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {
        if (blk < nblk) {
            f32x4 x = *(const lds_f4*)(xr + 16 * blk);
            if (inv) x = x * *(const lds_f4*)(inv + 16 * blk + 4 * lg) + *(const lds_f4*)(sh + 16 * blk + 4 * lg);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc = MFMA16(x[jj], wv[blk][jj], acc);
        }
    }
    if (lr < A) {
        const float bb = bias ? bias[lr] : 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) out[(wave * 16 + 4 * lg + reg) * A + lr] = acc[reg] + bb;
    }
}

// output layer backward through the BN below it, in place: bufB[r][k] (p) -> dz[r][k]
//   dW3[k][a] = sum_r y[r][k]*D[r][a]; db3[a] = sum_r D[r][a]; dy[r][k] = sum_a D[r][a]*W3[k][a]
// One column k per thread; its 64 activations stay in registers while the outputs go by in chunks of 4 (see out_fwd).
__device__ __attribute__((noinline)) void out_bwd(lds_f* bufB, const lds_f* invB, const lds_f* shB, const lds_f* rsB, const lds_f* mmB, int ldB,
                                        const lds_f* D, const glb_f* __restrict__ W3, int H2, int A, glb_f* __restrict__ gW3,
                                        glb_f* __restrict__ gb3, glb_f* __restrict__ gg, glb_f* __restrict__ gbe) {
    for (int k = threadIdx.x; k < H2; k += NTHREADS) {
        const float iv = invB[k], s = shB[k], rs = rsB[k], mean = mmB[k];
        float dgm = 0.f, dbt = 0.f;
        constexpr int HR = TILE / 2;  // rows per half: 2 x 32 registers instead of 2 x 64
#pragma nounroll
        for (int h = 0; h < 2; ++h) {
            const int rb = h * HR;
            float p[HR], dy[HR];
#pragma unroll
            for (int r = 0; r < HR; ++r) p[r] = bufB[(rb + r) * ldB + k], dy[r] = 0.f;
#pragma nounroll
            for (int a0 = 0; a0 < A; a0 += 4) {
                int ax[4];
                float w[4], dw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ax[i] = min(a0 + i, A - 1);
                    w[i] = (a0 + i < A) ? 1.f : 0.f;  // tail: clamped index, zero weight
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] *= W3[k * A + ax[i]];
#pragma unroll
                for (int r = 0; r < HR; ++r) {
                    const float y = fmaf(p[r], iv, s);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float d = D[(rb + r) * A + ax[i]];
                        dy[r] = fmaf(d, w[i], dy[r]);
                        dw[i] = fmaf(y, d, dw[i]);
                    }
                }
                if (gW3) {  // the second half adds to what the first one stored
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (a0 + i < A) gW3[k * A + a0 + i] = (h ? gW3[k * A + a0 + i] : 0.f) + dw[i];
                }
            }
#pragma unroll
            for (int r = 0; r < HR; ++r) {
                dgm = fmaf(dy[r] * (p[r] - mean), rs, dgm);
                dbt += dy[r];
                bufB[(rb + r) * ldB + k] = (p[r] > 0.f) ? dy[r] * iv : 0.f;
            }
        }
        if (gW3) gg[k] = dgm, gbe[k] = dbt;
    }
    if (gb3 && threadIdx.x < A) {
        float sum = 0.f;
        for (int r = 0; r < TILE; ++r) sum += D[r * A + threadIdx.x];
        gb3[threadIdx.x] = sum;
    }
}

// first-layer gradients from dz[r][c0..c0+H): dW[j][k] = sum_r X[r*K+j]*dz[r][k], db[k] = sum_r dz[r][k].
// dW = X^T dz on the matrix cores: (16 inputs) x (16 columns) tiles dealt over the waves, the 64 batch rows in 16 MFMA
// steps; both operands come from LDS. db: one column per thread.
__device__ __attribute__((noinline)) void l1_grads(const lds_f* X, int K, const lds_f* DZ, int ldz, int c0, int H, glb_f* __restrict__ gW,
                                         glb_f* __restrict__ gb) {
    for (int k = threadIdx.x; k < H; k += NTHREADS) {
        const lds_f* dzk = DZ + c0 + k;
        float sb = 0.f;
#pragma unroll 16
        for (int r = 0; r < TILE; ++r) sb += dzk[r * ldz];
        gb[k] = sb;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    const int jtiles = (K + 15) >> 4, ctiles = H >> 4;
    for (int item = wave; item < jtiles * ctiles; item += 4) {
        const int jt = item % jtiles, t = item / jtiles;
        const int j = 16 * jt + lr;
        const float jm = (j < K) ? 1.f : 0.f;
        const lds_f* xc = X + min(j, K - 1);
        const lds_f* dc = DZ + c0 + 16 * t + lr;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < TILE / 4; ++st) {
            const int r = 4 * st + lg;
            acc = MFMA16(xc[r * K] * jm, dc[r * ldz], acc);
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
    for (int i = threadIdx.x; i < n; i += NTHREADS) s += v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    lds_barrier();
    const float t = red[0] + red[1] + red[2] + red[3];
    lds_barrier();
    return t;
}

// One load per 128-byte line of [lo, hi): pulls a network's small tensors (everything but W2) into L2 ahead of the passes
// that read them a few values at a time -- cold, each such read is a serial trip to HBM.
__device__ __forceinline__ float warm(const float* __restrict__ base, int lo, int hi) {
    float t = 0.f;
    for (int i = lo + threadIdx.x * 32; i < hi; i += NTHREADS * 32) t += base[i];
    return t;
}

// FUSED (avd_learn_update_f32 for general widths): Adam + Polyak of the two W2 matrices in the epilogue of the weight-
// gradient GEMM that produces their gradient, theta ping-pong as in learn_kernel_t; the small tensors go through the gradient
// slab and adam_polyak_ranges_kernel.
template <bool FUSED>
__global__ __launch_bounds__(NTHREADS) void learn_kernel_g(avd_mlp_layout L, int set_mod,
                                                            const float* __restrict__ theta,
                                                            const float* __restrict__ stats,
                                                            float* __restrict__ theta_t,
                                                            float* __restrict__ stats_t,
                                                            const float* __restrict__ s, const float* __restrict__ a,
                                                            const float* __restrict__ r, const float* __restrict__ s2,
                                                            float gamma, float high, float* __restrict__ grads,
                                                            float* __restrict__ losses, UpdArgs upd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    GLds l = carve(smem, L);
    const int S = L.S, A = L.A, H1 = L.H1, H2 = L.H2, Ha = L.Ha, KC = H1 + Ha;
    const int ldA = ld_of(KC), ldB = ld_of(H2);
    const int agent = blockIdx.x, tid = threadIdx.x;
    const int set = set_mod > 0 ? agent % set_mod : agent;
    const Net net = {theta + (long)set * L.theta_size, stats + (long)set * L.stats_size};
    const Net tgt = {theta_t + (long)set * L.theta_size, stats_t + (long)set * L.stats_size};
    float* ga = grads + (long)agent * L.theta_size;
    float* gc = ga + L.actor_size;
    const float invn = 1.0f / (float)(TILE * A);
    typedef typename std::conditional<FUSED, AdamSink, StoreSink>::type BulkSink;
    BulkSink bulk;
    float* gw2 = ga;  // where gemm_dw "stores": the gradient slab, or the agent's slab of theta_out
    if constexpr (FUSED) {
        const int t = upd.step[agent];
        const float b1p = (float)pow((double)0.9f, (double)t), b2p = (float)pow((double)0.999f, (double)t);
        const float root = sqrtf(1.0f - b2p);
        const long o = (long)agent * L.theta_size;
        gw2 = upd.theta_out + o;
        bulk.wo = gw2, bulk.wi = net.th, bulk.wt = theta_t + o, bulk.m = upd.m + o, bulk.v = upd.v + o;
        bulk.alpha_a = (upd.actor_lr * root) / (1.0f - b1p), bulk.alpha_c = (upd.critic_lr * root) / (1.0f - b1p);
        bulk.tau = upd.tau, bulk.omt = upd.omt, bulk.actor_size = L.actor_size;
    }

    if (tid < TILE) l.sR[tid] = r[(long)agent * TILE + tid];
    for (int i = tid; i < TILE * A; i += NTHREADS) l.sAct[i] = a[(long)agent * TILE * A + i];
    if (tid == 0) {  // alignment padding of the gradient slab (only the A-wide biases can end off a 4-float boundary)
        for (int i = L.ab3 + A; i < L.actor_size; ++i) ga[i] = 0.f;
        for (int i = L.cb3 + A; i < L.theta_size - L.actor_size; ++i) gc[i] = 0.f;
    }

    {
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
        const Net n = (it == 0) ? tgt : net;
        if (it < 2) {  // state batch of this pass: s2 for the targets, s afterwards
            lds_barrier();
            const float* src = (it == 0 ? s2 : s) + (long)agent * TILE * S;
            for (int i = tid; i < TILE * S; i += NTHREADS) l.sX[i] = src[i];
            lds_barrier();
        }
        PH(20);
        if (it != 1) {  // ---- actor forward (agent/model.py:26-36)
            const float* th = n.th;
            l1_fwd(LDSP(l.sX), S, CGLBP(th + L.aW1), CGLBP(th + L.ab1), CGLBP(th + L.ag1), CGLBP(th + L.abe1), CGLBP(n.st + L.amm1),
                   CGLBP(n.st + L.amv1), H1, LDSP(l.bufA), ldA, 0, LDSP(l.invA), LDSP(l.shA));
            PH(21);
            coefs_b(th + L.ag2, th + L.abe2, n.st + L.amm2, n.st + L.amv2, H2, l);
            lds_barrier();
            PH(1);
            gemm_fwd_relu(l.bufA, ldA, l.invA, l.shA, H1, th + L.aW2, th + L.ab2, H2, l.bufB, ldB);
            lds_barrier();
            PH(2);
            narrow_gemm(LDSP(l.bufB), ldB, LDSP(l.invB), LDSP(l.shB), H2, CGLBP(th + L.aW3), A, 1, A, CGLBP(th + L.ab3), LDSP(l.sQ));
            lds_barrier();
            for (int i = tid; i < TILE * A; i += NTHREADS) {
                const float t = tanhf(l.sQ[i]);
                l.sT[i] = t, l.sA1[i] = t * high;
            }
            lds_barrier();
            PH(3);
        }
        if (it != 3) {  // ---- critic forward (agent/model.py:63-83)
            const float* th = n.th + L.actor_size;
            const float* act = (it == 1) ? l.sAct : l.sA1;
            l1_fwd(LDSP(l.sX), S, CGLBP(th + L.cWs), CGLBP(th + L.cbs), CGLBP(th + L.cgs), CGLBP(th + L.cbes), CGLBP(n.st + L.cmms),
                   CGLBP(n.st + L.cmvs), H1, LDSP(l.bufA), ldA, 0, LDSP(l.invA), LDSP(l.shA));
            PH(22);
            l1_fwd(LDSP(act), A, CGLBP(th + L.cWa), CGLBP(th + L.cba), CGLBP(th + L.cga), CGLBP(th + L.cbea), CGLBP(n.st + L.cmma),
                   CGLBP(n.st + L.cmva), Ha, LDSP(l.bufA), ldA, H1, LDSP(l.invA), LDSP(l.shA));
            PH(23);
            coefs_b(th + L.cg3, th + L.cbe3, n.st + L.cmm3, n.st + L.cmv3, H2, l);
            lds_barrier();
            PH(4);
            gemm_fwd_relu(l.bufA, ldA, l.invA, l.shA, KC, th + L.cW2, th + L.cb2, H2, l.bufB, ldB);
            lds_barrier();
            PH(5);
            narrow_gemm(LDSP(l.bufB), ldB, LDSP(l.invB), LDSP(l.shB), H2, CGLBP(th + L.cW3), A, 1, A, CGLBP(th + L.cb3), LDSP(l.sQ));
            lds_barrier();
            PH(6);
        }
        if (it == 0) {  // y = r + gamma * Q'(s2, mu'(s2)), r broadcast over the A outputs, no done mask (trainer.py:494)
            for (int i = tid; i < TILE * A; i += NTHREADS) l.sY[i] = fmaf(gamma, l.sQ[i], l.sR[i / A]);
            if constexpr (FUSED) {  // the frozen BN statistics take part in the soft update too (ddpgagent.py:44-53)
#pragma clang fp contract(off)
                float* stt = stats_t + (long)set * L.stats_size;
                for (int i = tid; i < L.stats_size; i += NTHREADS) stt[i] = net.st[i] * upd.tau + stt[i] * upd.omt;
            }
            continue;
        }
        if (it == 1) {  // Lc = mean((y - q)^2) over B*A (trainer.py:496)
            for (int i = tid; i < TILE * A; i += NTHREADS) {
                const float e = l.sY[i] - l.sQ[i];
                l.sD[i] = -2.0f * e * invn;
                l.sT[i] = e * e;
            }
            lds_barrier();
            const float lc = block_sum(l.sT, TILE * A, l.red) * invn;
            if (tid == 0 && losses) losses[(long)agent * 2 + 0] = lc;
        } else if (it == 2) {  // La = -mean(q1) (trainer.py:504)
            const float la = -block_sum(l.sQ, TILE * A, l.red) * invn;
            if (tid == 0 && losses) losses[(long)agent * 2 + 1] = la;
            for (int i = tid; i < TILE * A; i += NTHREADS) l.sD[i] = -invn;
        } else {  // through tanh(.)*high
            for (int i = tid; i < TILE * A; i += NTHREADS) {
                const float t = l.sT[i];
                l.sD[i] = l.sDa[i] * high * (1.0f - t * t);
            }
        }
        lds_barrier();
        const bool crit = (it != 3), wg = (it != 2);
        const float* wth = crit ? net.th + L.actor_size : net.th;
        float* gout = crit ? gc : ga;
        out_bwd(LDSP(l.bufB), LDSP(l.invB), LDSP(l.shB), LDSP(l.rsB), LDSP(l.mmB), ldB, LDSP(l.sD),
                CGLBP(wth + (crit ? L.cW3 : L.aW3)), H2, A, GLBP(wg ? gout + (crit ? L.cW3 : L.aW3) : nullptr),
                GLBP(wg ? gout + (crit ? L.cb3 : L.ab3) : nullptr), GLBP(gout + (crit ? L.cg3 : L.ag2)),
                GLBP(gout + (crit ? L.cbe3 : L.abe2)));
        lds_barrier();
        PH(it == 1 ? 7 : (it == 2 ? 12 : 15));
        if (wg) {
            col_sums(l.bufB, ldB, H2, l.db, gout + (crit ? L.cb2 : L.ab2));
            lds_barrier();
            PH(it == 1 ? 8 : 16);
            gemm_dw(l.bufA, ldA, l.invA, l.shA, crit ? KC : H1, l.bufB, ldB, l.db, H2,
                    gw2 + (crit ? L.actor_size + L.cW2 : L.aW2), bulk);
            lds_barrier();
            PH(it == 1 ? 9 : 17);
        }
        const float* w2 = wth + (crit ? L.cW2 : L.aW2);
        if (it != 2) {
            if (crit)
                gemm_dx_bn(l.bufB, ldB, H2, w2, 0, H1, l.bufA, ldA, wth + L.cgs, net.st + L.cmms, net.st + L.cmvs,
                           gc + L.cgs, gc + L.cbes);
            else
                gemm_dx_bn(l.bufB, ldB, H2, w2, 0, H1, l.bufA, ldA, wth + L.ag1, net.st + L.amm1, net.st + L.amv1,
                           ga + L.ag1, ga + L.abe1);
        }
        if (crit) {
            const float* cth = net.th + L.actor_size;
            gemm_dx_bn(l.bufB, ldB, H2, w2, H1, KC, l.bufA, ldA, cth + L.cga, net.st + L.cmma, net.st + L.cmva,
                       wg ? gc + L.cga : nullptr, wg ? gc + L.cbea : nullptr);
        }
        lds_barrier();
        PH(it == 1 ? 10 : (it == 2 ? 13 : 18));
        if (it == 1) {
            l1_grads(LDSP(l.sX), S, LDSP(l.bufA), ldA, 0, H1, GLBP(gc + L.cWs), GLBP(gc + L.cbs));
            l1_grads(LDSP(l.sAct), A, LDSP(l.bufA), ldA, H1, Ha, GLBP(gc + L.cWa), GLBP(gc + L.cba));
            PH(11);
        } else if (it == 2) {  // da[r][a] = sum_j dza[r][j] * Wa[a][j]
            const float* cth = net.th + L.actor_size;
            narrow_gemm(LDSP(l.bufA + H1), ldA, (const lds_f*)nullptr, (const lds_f*)nullptr, Ha, CGLBP(cth + L.cWa), 1, Ha, A,
                        (const glb_f*)nullptr, LDSP(l.sDa));
            PH(14);
        } else {
            l1_grads(LDSP(l.sX), S, LDSP(l.bufA), ldA, 0, H1, GLBP(ga + L.aW1), GLBP(ga + L.ab1));
            PH(19);
        }
        lds_barrier();  // (the next pass's first layer overwrites bufA while slower waves may still read dz1 from it)
    }
}

}  // namespace gen

// ------------------------------------------------------------------------------------------
// Dimension-specialised learn kernel (reference widths known at compile time).
//
// Same algorithm and LDS plan as learn_kernel above; what changes is the code shape, chosen from its ISA:
//  * S/H1/H2/Ha are template constants -> row strides and weight strides fold into immediates / scalar adds
//    (the generic kernel carried one 64-bit VGPR pointer per in-flight weight load);
//  * the four network passes (targets, critic, actor-through-critic, actor) run as ONE loop whose body holds a
//    single copy of each routine, so the kernel fits the 64 KB instruction cache (the generic kernel is ~240 KB
//    of straight-line code that is fetched once per tile);
//  * pipelined loops have no guards inside their unrolled bodies: prefetch indices are clamped (a redundant
//    load at the end instead of a branch) and the remainder blocks run through a rotating tail.
// ------------------------------------------------------------------------------------------
namespace fast {

// 4 waves, one per SIMD, two 16-column MFMA tiles each. (An 8-wave / 2-per-SIMD variant measured the same time:
// the phases are paced by the MFMA pipe and by the cold loads at their heads, not by per-wave stalls; and at two
// waves per SIMD the 256-register cap leaves no room for the cross-phase prefetch state.)
constexpr int FT = 256;
constexpr int NW = FT / 64;
constexpr int R = 4;     // weight-operand register ring depth (blocks of 16 k)

struct RawA {  // LDS operands of one 16-deep block: post-ReLU activations of 4 row tiles + the block's BN coefficients
    f32x4 x[4], iv, sf;
};
template <int LDX>
__device__ __forceinline__ void read_a(RawA& q, const float* X, const float* inv, const float* sh, int k4, int lr) {
    q.iv = *(const f32x4*)(inv + k4);
    q.sf = *(const f32x4*)(sh + k4);
#pragma unroll
    for (int m = 0; m < 4; ++m) q.x[m] = *(const f32x4*)(X + (m * 16 + lr) * LDX + k4);
}

template <int NT>
__device__ __forceinline__ void ldn(float (&d)[NT], const float* p) {  // NT consecutive floats, one access
    if constexpr (NT == 2) {
        const f32x2 v = *(const f32x2*)p;
        d[0] = v[0], d[1] = v[1];
    } else {
        d[0] = p[0];
    }
}
template <int NT>
__device__ __forceinline__ void stn(float* p, const float (&d)[NT]) {
    if constexpr (NT == 2) {
        f32x2 v;
        v[0] = d[0], v[1] = d[1];
        *(f32x2*)p = v;
    } else {
        p[0] = d[0];
    }
}

// One 16-deep block: sum_k (p*inv + sh)[r][k] * W[k][n] is evaluated as  p @ (inv (.) W)  +  (sh . W[:, n]):
// the BN scale goes onto the prefetched weight operand (v_mul), the shift into a per-column constant (v_fma),
// so the A operand is the raw LDS value and no VALU result sits between an LDS read and an MFMA.
template <int NT>
__device__ __forceinline__ void mfma_block(f32x4 (&acc)[4][NT], float (&cs)[NT], const RawA& q,
                                           const float (&w)[4][NT]) {
    float bs[4][NT];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bs[jj][t] = w[jj][t] * q.iv[jj];
            cs[t] = fmaf(w[jj][t], q.sf[jj], cs[t]);
        }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = MFMA16(q.x[m][jj], bs[jj][t], acc[m][t]);
}

// First R-1 weight blocks + bias of a forward GEMM, requested one phase early (before the first-layer VALU phase
// that precedes the GEMM) so that the GEMM does not open on a cold HBM miss.
// Column map: wave w owns columns [16*NT*w, 16*NT*(w+1)); its MFMA tile t holds columns base + NT*lr + t.
template <int NT>
struct FwdPre {
    float ring[R][4][NT];
    float bc[NT];
};
template <int N>
__device__ __forceinline__ void fwd_prefetch(FwdPre<N / (16 * NW)>& p, const float* __restrict__ W,
                                             const float* __restrict__ b, int nblk, int first = 0) {
    constexpr int NT = N / (16 * NW);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    const int col = wave * 16 * NT + NT * lr;
    const float* wl = W + (4 * lg) * N + col;
    ldn<NT>(p.bc, b + col);
#pragma unroll
    for (int d = 0; d < R - 1; ++d) {
        const float* q = wl + min(first + d, nblk - 1) * (16 * N);  // first % R == 0: block first + d sits in ring[d]
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ldn<NT>(p.ring[d][jj], q + jj * N);
    }
}

// out[r][n] = relu(sum_k bn(X[r][k]) * W[k][n] + b[n]).
// Per 16-deep block: [issue weight loads R-1 blocks ahead + LDS reads one block ahead] | [16*NT MFMAs].
// sched_barrier(0) pins the two stages: hipcc's scheduler otherwise sinks every load down to its first use and
// neither the register ring nor the LDS double buffer prefetches anything.
// FIRST / SNAP split the reduction for inputs whose leading SNAP blocks do not change between two calls (the critic's
// state features in passes 1 and 2): a call with `snap` set stores the accumulators as they stand before block SNAP
// (layout of the outputs; the lane's shift sums go to cs_snap), a call with FIRST == SNAP resumes from them and runs only
// blocks FIRST.. -- the same additions in the same order as a full run, bit for bit.
template <int N, int LDX, int LDO, int NBLK, int FIRST = 0, int SNAP = 0>
__device__ __forceinline__ void gemm_fwd(const float* X, const float* inv, const float* sh,
                                         const float* __restrict__ W, FwdPre<N / (16 * NW)>& pre, float* out,
                                         float* snap = nullptr, float* cs_snap = nullptr) {
    constexpr int NT = N / (16 * NW);
    static_assert(NT == 1 || NT == 2, "one or two 16-column tiles per wave");
    static_assert(FIRST % R == 0 && FIRST < NBLK && SNAP < NBLK, "resume point must be ring-aligned");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    const int col = wave * 16 * NT + NT * lr;
    f32x4 acc[4][NT];
    float cs[NT];  // this lane's share (its k's) of sum_k sh[k]*W[k][col..]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        cs[t] = (FIRST > 0) ? cs_snap[t] : 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (FIRST > 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[NT];
                ldn<NT>(v, snap + (m * 16 + lg * 4 + j) * LDO + col);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t][j] = v[t];
            }
    }
    const float* wl = W + (4 * lg) * N + col;  // lane's columns, row 4*lg of block 0
    float(&ring)[R][4][NT] = pre.ring;
    auto load_blk = [&](float(&dst)[4][NT], int blk) {
        const float* p = wl + blk * (16 * N);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ldn<NT>(dst[jj], p + jj * N);
    };
    RawA raw[2];
    read_a<LDX>(raw[FIRST & 1], X, inv, sh, 16 * FIRST + 4 * lg, lr);
    // Fully unrolled over the NBLK reduction blocks. As a loop over groups of R blocks the compiler placed register
    // copies of the ring on the back-edge, i.e. an s_waitcnt vmcnt(0) per group: the 3-block lookahead was drained
    // every 4 blocks (measured ~30 % of the GEMM time).
#pragma unroll
    for (int blk = FIRST; blk < NBLK; ++blk) {
        if (SNAP > 0 && blk == SNAP && snap) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) v[t] = acc[m][t][j];
                    stn<NT>(snap + (m * 16 + lg * 4 + j) * LDO + col, v);
                }
#pragma unroll
            for (int t = 0; t < NT; ++t) cs_snap[t] = cs[t];
        }
        if (blk + R - 1 < NBLK) load_blk(ring[(blk + R - 1) % R], blk + R - 1);
        if (blk + 1 < NBLK) read_a<LDX>(raw[(blk + 1) & 1], X, inv, sh, 16 * (blk + 1) + 4 * lg, lr);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block<NT>(acc, cs, raw[blk & 1], ring[blk % R]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        cs[t] += __shfl_xor(cs[t], 16);
        cs[t] += __shfl_xor(cs[t], 32);
        cs[t] += pre.bc[t];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float o[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) o[t] = fmaxf(acc[m][t][j] + cs[t], 0.f);
            stn<NT>(out + (m * 16 + lg * 4 + j) * LDO + col, o);
        }
}

// dW[k][n] = inv[k] * sum_r P[r][k]*DZ[r][n] + sh[k]*db[n], k < K (runtime, % 4 == 0).
// Wave w owns columns [16*NT*w, ..); output tile (ta, t) of a 64-row block holds rows k0 + 4*i + ta
// (i = 4*lg + reg) and columns base + NT*lr + t.
template <int N, int LDP, int LDZ, class Sink>
__device__ __forceinline__ void gemm_dw(const float* P, const float* inv, const float* sh, int K, const float* DZ,
                                        const float* db, float* __restrict__ gW, Sink sink) {
    constexpr int NT = N / (16 * NW);
    constexpr bool kFused = !std::is_same<Sink, StoreSink>::value;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    const int col = wave * 16 * NT + NT * lr;
    float dbc[NT];
    ldn<NT>(dbc, db + col);
    const float* dp = DZ + lg * LDZ + col;
#pragma nounroll
    for (int k0 = 0; k0 < K; k0 += 64) {
        // Fused update: the Adam operands (w, w_target, m, v) of this block's 16 (row, column-pair) pieces are requested
        // BEFORE the MFMA loop. vmcnt retires loads and stores in issue order, so operand loads issued after the
        // previous block's update stores would wait for those stores' acknowledgements; issued here they only
        // queue behind stores that are a whole MFMA loop old.
        typename std::conditional<kFused, AdamSink::Quad, int>::type q[16];
        long base = 0;
        if constexpr (kFused) {
            static_assert(NT == 2, "fused epilogue works on column pairs");
            base = (gW - sink.wo) + col;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ta = 0; ta < 4; ++ta) {
                    const int kb = min(k0 + 4 * (lg * 4 + j), K - 4);  // clamp: loads for unused rows stay inside the tensor
                    sink.load2(q[j * 4 + ta], base + (long)(kb + ta) * N);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 acc[4][NT];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* pp = P + lg * LDP + k0 + 4 * lr;  // may read past K in the last block: those rows are not stored
        f32x4 pa[2];
        float dz[2][NT];
        pa[0] = *(const f32x4*)(pp);
        ldn<NT>(dz[0], dp);
#pragma unroll
        for (int it = 0; it < TILE / 4; ++it) {
            if (it + 1 < TILE / 4) {
                pa[(it + 1) & 1] = *(const f32x4*)(pp + 4 * (it + 1) * LDP);
                ldn<NT>(dz[(it + 1) & 1], dp + 4 * (it + 1) * LDZ);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[ta][t] = MFMA16(pa[it & 1][ta], dz[it & 1][t], acc[ta][t]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kbase = k0 + 4 * (lg * 4 + j);
            if (kbase < K) {
                const f32x4 iv = *(const f32x4*)(inv + kbase);
                const f32x4 sf = *(const f32x4*)(sh + kbase);
#pragma unroll
                for (int ta = 0; ta < 4; ++ta) {
                    float o[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) o[t] = fmaf(iv[ta], acc[ta][t][j], sf[ta] * dbc[t]);
                    if constexpr (kFused)
                        sink.update2(q[j * 4 + ta], base + (long)(kbase + ta) * N, o);
                    else
                        stn<NT>(gW + (kbase + ta) * N + col, o);
                }
            }
        }
    }
}

struct BnSet {  // BN parameters / gradient outputs of a column range, indexed by (c - base)
    const float *g, *mm, *mv;
    float *dg, *dbe;
    int base;
};

// dy[r][c] = sum_n DZ[r][n]*W[c][n] for c in [c_begin, c_end) (16-column tiles round-robin over the 8 waves), then
// the BN/ReLU backward of the layer below in place in P. Columns < split use `lo`, the others `hi`.
template <int N>
struct DxPre {  // a wave's first tile (W slice + BN parameters), requested before the phase that precedes gemm_dx
    f32x4 w[N / 16];
    float bn[3];
};
template <int N>
__device__ __forceinline__ void dx_load_tile(f32x4 (&w)[N / 16], float (&bn)[3], const float* __restrict__ W, int c0,
                                             const BnSet& lo, const BnSet& hi, int split) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    const float* wrow = W + (c0 + lr) * N + 4 * lg;
#pragma unroll
    for (int q = 0; q < N / 16; ++q) w[q] = *(const f32x4*)(wrow + 16 * q);
    const BnSet& s = (c0 < split) ? lo : hi;
    const int i = c0 + lr - s.base;
    bn[0] = s.g[i], bn[1] = s.mm[i], bn[2] = s.mv[i];
}
template <int N>
__device__ __forceinline__ void dx_prefetch(DxPre<N>& p, const float* __restrict__ W, int c_begin, int c_end,
                                            const BnSet& lo, const BnSet& hi, int split) {
    const int c0 = c_begin + (threadIdx.x >> 6) * 16;
    p.bn[0] = 0.f, p.bn[1] = 0.f, p.bn[2] = 1.f;
    if (c0 < c_end) dx_load_tile<N>(p.w, p.bn, W, c0, lo, hi, split);
}

template <int N, int LDZ, int LDP, class Sink>
__device__ __forceinline__ void gemm_dx(const float* DZ, const float* __restrict__ W, int c_begin, int c_end, float* P,
                                        BnSet lo, BnSet hi, int split, bool write_grads, DxPre<N>& pre, Sink sink) {
    static_assert(N == 128, "N/16 == 8 reduction blocks held in registers");
    constexpr int NB = N / 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    f32x4(&wc)[NB] = pre.w;
    f32x4 wn[NB];
    float(&bnc)[3] = pre.bn;
    float bnn[3] = {0.f, 0.f, 1.f};
    int c0 = c_begin + wave * 16;
#pragma nounroll
    for (; c0 < c_end; c0 += NW * 16) {
        PHX_T0();
        const int cn = c0 + NW * 16;
        if (cn < c_end) dx_load_tile<N>(wn, bnn, W, cn, lo, hi, split);
        // this tile's P values (the BN backward needs them): read now, consumed after the MFMAs
        float pv[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) pv[m][j] = P[(m * 16 + lg * 4 + j) * LDP + c0 + lr];
        __builtin_amdgcn_sched_barrier(0);  // keep the next tile's loads ahead of this tile's MFMAs
        PHX(20);
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
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = MFMA16(a[q & 1][m][jj], wc[q][jj], acc[m]);
            __builtin_amdgcn_sched_barrier(0);
        }
        PHX(21);
        const int c = c0 + lr;
        const float rs = 1.0f / sqrtf(bnc[2] + BN_EPS);
        const float gam = bnc[0], mean = bnc[1];
        float sg = 0.f, sb = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dy = acc[m][j];
                const float p = pv[m][j];
                sg = fmaf(dy * (p - mean), rs, sg);
                sb += dy;
                P[(m * 16 + lg * 4 + j) * LDP + c] = (p > 0.f) ? dy * (rs * gam) : 0.f;
            }
        sg += __shfl_xor(sg, 16);
        sg += __shfl_xor(sg, 32);
        sb += __shfl_xor(sb, 16);
        sb += __shfl_xor(sb, 32);
        if (write_grads && lg == 0) {
            const BnSet& s = (c0 < split) ? lo : hi;
            sink.put(s.dg + (c - s.base), sg);
            sink.put(s.dbe + (c - s.base), sb);
        }
        PHX(22);
#pragma unroll
        for (int q = 0; q < NB; ++q) wc[q] = wn[q];
        bnc[0] = bnn[0], bnc[1] = bnn[1], bnc[2] = bnn[2];
        PHX(23);
    }
}

// Weight-gradient and input-gradient GEMMs of one layer, interleaved per block of 64 features:
//     for k0:  dW rows [k0, k0+64) (+ the fused Adam/Polyak epilogue)  |  barrier  |  dX columns [k0, k0+64), one 16-column
//              tile per wave, BN/ReLU backward in place in P
// Legal because block k0 of dW is the only reader of P's columns [k0, k0+64). Why: the input-gradient GEMM needs W2's
// rows [k0, k0+64) -- exactly the rows the update epilogue of this block has just pulled through L2 -- so its operand
// loads are L2 hits instead of a second trip to HBM after the whole update stream has gone by. They are issued between
// the MFMA loop and the epilogue, i.e. ahead of the epilogue's stores in vmcnt order.
template <int N, int LDP, int LDZ, class Sink, class SmallSink>
__device__ __forceinline__ void gemm_dw_dx(float* P, const float* inv, const float* sh, int K, const float* DZ,
                                           const float* db, float* __restrict__ gW, Sink sink,
                                           const float* __restrict__ W, BnSet lo, BnSet hi, int split, SmallSink small) {
    static_assert(N == 128, "N/16 == 8 reduction blocks held in registers");
    constexpr int NT = N / (16 * NW), NB = N / 16;
    constexpr bool kFused = !std::is_same<Sink, StoreSink>::value;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    const int col = wave * 16 * NT + NT * lr;
    float dbc[NT];
    ldn<NT>(dbc, db + col);
    const float* dp = DZ + lg * LDZ + col;
#pragma nounroll
    for (int k0 = 0; k0 < K; k0 += 64) {
        typename std::conditional<kFused, AdamSink::Quad, int>::type q[16];
        long base = 0;
        if constexpr (kFused) {
            static_assert(NT == 2, "fused epilogue works on column pairs");
            base = (gW - sink.wo) + col;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ta = 0; ta < 4; ++ta) {
                    const int kb = min(k0 + 4 * (lg * 4 + j), K - 4);
                    sink.load2(q[j * 4 + ta], base + (long)(kb + ta) * N);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 acc[4][NT];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* pp = P + lg * LDP + k0 + 4 * lr;
        f32x4 pa[2];
        float dz[2][NT];
        pa[0] = *(const f32x4*)(pp);
        ldn<NT>(dz[0], dp);
#pragma unroll
        for (int it = 0; it < TILE / 4; ++it) {
            if (it + 1 < TILE / 4) {
                pa[(it + 1) & 1] = *(const f32x4*)(pp + 4 * (it + 1) * LDP);
                ldn<NT>(dz[(it + 1) & 1], dp + 4 * (it + 1) * LDZ);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[ta][t] = MFMA16(pa[it & 1][ta], dz[it & 1][t], acc[ta][t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // this wave's input-gradient tile of the block: W2 rows c0 .. c0+15 (L2 hits) and the tile's BN parameters
        const int c0 = k0 + wave * 16;
        const bool have = c0 < K;
        f32x4 wc[NB];
        float bn[3] = {0.f, 0.f, 1.f};
        if (have) dx_load_tile<N>(wc, bn, W, c0, lo, hi, split);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();  // every wave is done reading P[:, k0 .. k0+63]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kbase = k0 + 4 * (lg * 4 + j);
            if (kbase < K) {
                const f32x4 iv = *(const f32x4*)(inv + kbase);
                const f32x4 sf = *(const f32x4*)(sh + kbase);
#pragma unroll
                for (int ta = 0; ta < 4; ++ta) {
                    float o[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) o[t] = fmaf(iv[ta], acc[ta][t][j], sf[ta] * dbc[t]);
                    if constexpr (kFused)
                        sink.update2(q[j * 4 + ta], base + (long)(kbase + ta) * N, o);
                    else
                        stn<NT>(gW + (kbase + ta) * N + col, o);
                }
            }
        }
        if (have) {
            float pv[4][4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) pv[m][j] = P[(m * 16 + lg * 4 + j) * LDP + c0 + lr];
            f32x4 dacc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) dacc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 a[2][4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[0][m] = *(const f32x4*)(DZ + (m * 16 + lr) * LDZ + 4 * lg);
#pragma unroll
            for (int qq = 0; qq < NB; ++qq) {
                if (qq + 1 < NB)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        a[(qq + 1) & 1][m] = *(const f32x4*)(DZ + (m * 16 + lr) * LDZ + 16 * (qq + 1) + 4 * lg);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int m = 0; m < 4; ++m) dacc[m] = MFMA16(a[qq & 1][m][jj], wc[qq][jj], dacc[m]);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int c = c0 + lr;
            const float rs = 1.0f / sqrtf(bn[2] + BN_EPS);
            const float gam = bn[0], mean = bn[1];
            float sg = 0.f, sb = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dy = dacc[m][j];
                    const float p = pv[m][j];
                    sg = fmaf(dy * (p - mean), rs, sg);
                    sb += dy;
                    P[(m * 16 + lg * 4 + j) * LDP + c] = (p > 0.f) ? dy * (rs * gam) : 0.f;
                }
            sg += __shfl_xor(sg, 16);
            sg += __shfl_xor(sg, 32);
            sb += __shfl_xor(sb, 16);
            sb += __shfl_xor(sb, 32);
            if (lg == 0) {
                const BnSet& s = (c0 < split) ? lo : hi;
                small.put(s.dg + (c - s.base), sg);
                small.put(s.dbe + (c - s.base), sb);
            }
        }
    }
}

// Parameters of a network's first layers, loaded one phase before they are used: the BN parameters of the thread's own
// column (coefficient tables) and, for the layer itself -- out = relu(X W + b), K = S or 1 inputs, ONE MFMA step per
// 16 x 16 output tile -- the lane's weight / bias operands of the up to four column tiles its wave owns (tile t = wave +
// 4 i). As a VALU loop (one column per thread, 64 rows x K FMAs, an LDS round trip per register block) the first layers
// were 10 % of the kernel.
template <int K>
struct L1P {
    float mw[4], mb[4], g, be, mm, mv;
};
template <int K>
__device__ __forceinline__ L1P<K> l1p_load(const float* __restrict__ W, const float* __restrict__ b,
                                           const float* __restrict__ g, const float* __restrict__ be,
                                           const float* __restrict__ mm, const float* __restrict__ mv, int H, int k) {
    static_assert(K <= 4, "one MFMA step");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    L1P<K> c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = min(16 * (wave + 4 * i) + lr, H - 1);  // tiles past H: clamped loads, never used
        c.mw[i] = W[min(lg, K - 1) * H + col] * (lg < K ? 1.f : 0.f);
        c.mb[i] = b[col];
    }
    c.g = g[k], c.be = be[k], c.mm = mm[k], c.mv = mv[k];
    return c;
}
template <int K>
__device__ __forceinline__ void l1p_coefs(const L1P<K>& c, float* inv, float* sh, int idx) {
    const float iv = (1.0f / sqrtf(c.mv + BN_EPS)) * c.g;
    inv[idx] = iv;
    sh[idx] = c.be - c.mm * iv;
}
// out[r][col0 + n] = relu(sum_j X[r*K + j] * W[j][n] + b[n]) for n < H: tile (m, t) -> rows 16m + 4lg + reg, column 16t + lr
template <int K>
__device__ __forceinline__ void l1p_mfma(const L1P<K>& c, const float* X, float* out, int ld, int col0, int H) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    float a[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = X[(16 * m + lr) * K + min(lg, K - 1)];  // lanes lg >= K meet a zero weight
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = wave + 4 * i;
        if (16 * t < H) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 acc = MFMA16(a[m], c.mw[i], ((f32x4){0.f, 0.f, 0.f, 0.f}));
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    out[(16 * m + 4 * lg + reg) * ld + col0 + 16 * t + lr] = fmaxf(acc[reg] + c.mb[i], 0.f);
            }
        }
    }
}
template <int S>
struct ActorPar {
    L1P<S> c1;
    L2Col c2;
    float b3;
};
template <int S>
struct CriticPar {
    L1P<S> cs;
    L1P<1> ca;
    L2Col c2;
    float b3;
};
template <int S, int H1, int H2>
__device__ __forceinline__ ActorPar<S> load_actor(const avd_mlp_layout& L, Net n, int ks, int tid) {
    const float* th = n.th;
    ActorPar<S> p;
    p.c1 = l1p_load<S>(th + L.aW1, th + L.ab1, th + L.ag1, th + L.abe1, n.st + L.amm1, n.st + L.amv1, H1, ks);
    p.c2 = l2_load(th + L.ag2, th + L.abe2, n.st + L.amm2, n.st + L.amv2, th + L.aW3, H2, tid);
    p.b3 = th[L.ab3];
    return p;
}
template <int S, int H1, int H2, int HA>
__device__ __forceinline__ CriticPar<S> load_critic(const avd_mlp_layout& L, Net n, int ks, int ka, int tid) {
    const float* th = n.th + L.actor_size;
    const float* st = n.st;
    CriticPar<S> p;
    p.cs = l1p_load<S>(th + L.cWs, th + L.cbs, th + L.cgs, th + L.cbes, st + L.cmms, st + L.cmvs, H1, ks);
    p.ca = l1p_load<1>(th + L.cWa, th + L.cba, th + L.cga, th + L.cbea, st + L.cmma, st + L.cmva, HA, ka);
    p.c2 = l2_load(th + L.cg3, th + L.cbe3, st + L.cmm3, st + L.cmv3, th + L.cW3, H2, tid);
    p.b3 = th[L.cb3];
    return p;
}

template <int S, int H1, int H2, int HA, bool FUSED>
__global__ __launch_bounds__(FT) void learn_kernel_t(avd_mlp_layout L, int set_mod, const float* __restrict__ theta,
                                                      const float* __restrict__ stats, float* __restrict__ theta_t,
                                                      float* __restrict__ stats_t, const float* __restrict__ s,
                                                      const float* __restrict__ a,
                                                      const float* __restrict__ r, const float* __restrict__ s2,
                                                      float gamma, float high, float* __restrict__ grads,
                                                      float* __restrict__ losses, UpdArgs upd) {
    static_assert(H1 <= FT && HA <= FT && H2 <= FT && FT % H1 == 0 && H1 % 16 == 0 && HA % 16 == 0, "widths");
    constexpr int KC = H1 + HA, LDA = ld_of(KC), LDB = ld_of(H2);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    LearnLds l = carve(smem, L, FT);
    const int agent = blockIdx.x;
    const int set = set_mod > 0 ? agent % set_mod : agent;
    Net net = {theta + (long)set * L.theta_size, stats + (long)set * L.stats_size};
    Net tgt = {theta_t + (long)set * L.theta_size, stats_t + (long)set * L.stats_size};
    // gradient slab of this agent. Fused form: only the small tensors (biases, BN gamma/beta, first and last layers,
    // ~6 % of the parameters) are stored here -- a range-restricted Adam launch consumes them; the two W2 matrices
    // are updated in the weight-gradient GEMM epilogues through `bulk`.
    float* g = grads + (long)agent * L.theta_size;
    float* ga = g;                 // actor block
    float* gc = g + L.actor_size;  // critic block
    const int tid = threadIdx.x;
    const StoreSink sink;
    typedef typename std::conditional<FUSED, AdamSink, StoreSink>::type BulkSink;
    BulkSink bulk;
    float* gw2 = g;  // where gemm_dw "stores": the gradient slab, or the agent's slab of theta_out
    if constexpr (FUSED) {
        const int t = upd.step[agent];
        const float b1p = (float)pow((double)0.9f, (double)t), b2p = (float)pow((double)0.999f, (double)t);
        const float root = sqrtf(1.0f - b2p);
        const long o = (long)agent * L.theta_size;
        gw2 = upd.theta_out + o;
        bulk.wo = gw2, bulk.wi = net.th, bulk.wt = theta_t + o, bulk.m = upd.m + o, bulk.v = upd.v + o;
        bulk.alpha_a = (upd.actor_lr * root) / (1.0f - b1p), bulk.alpha_c = (upd.critic_lr * root) / (1.0f - b1p);
        bulk.tau = upd.tau, bulk.omt = upd.omt, bulk.actor_size = L.actor_size;
    }
    constexpr float invn = 1.0f / (float)TILE;  // A == 1
    constexpr int LPR = FT / 64;                // lanes per batch row in the width-1 output layers
    // thread -> (column, row phase) maps of the first-layer phases
    const int ks = tid % H1, rs0 = tid / H1;
    const int ka = tid % HA, ra = tid / HA;  // ra >= agroups: spare threads (they still load a valid column)

    for (int i = tid; i < TILE * S; i += FT) {
        l.sS[i] = s[(long)agent * TILE * S + i];
        l.sS2[i] = s2[(long)agent * TILE * S + i];
    }
    if (tid < TILE) {
        l.sAct[tid] = a[(long)agent * TILE + tid];
        l.sR[tid] = r[(long)agent * TILE + tid];
    }
    if (tid == 0) {  // alignment padding of the gradient slab
        for (int i = L.ab3 + 1; i < L.actor_size; ++i) ga[i] = 0.f;
        for (int i = L.cb3 + 1; i < L.theta_size - L.actor_size; ++i) gc[i] = 0.f;
    }
    lds_barrier();
    PH_INIT();
    PH(0);

    // pass 0: targets (y); pass 1: critic loss + gradient; pass 2: actor -> critic, gradient wrt the action;
    // pass 3: actor forward again (activations kept) + actor gradient          (workers/trainer.py:492-506)
    // Every phase's global loads are requested one phase early and carried in registers across the phase in
    // between (weights of the next GEMM, parameter columns of the next forward): with one workgroup per CU there
    // is nobody else to hide a cold HBM round trip at the head of each phase.
    // The number of loads issued on every path between a load and its first use is kept identical (both parameter
    // sets are always fetched together): s_waitcnt vmcnt(N) for an older load counts the younger loads in flight, and
    // where paths differ the compiler must take the smallest count -- i.e. wait for the extra loads of the longer path.
    ActorPar<S> pa = load_actor<S, H1, H2>(L, tgt, ks, tid);
    CriticPar<S> pc = load_critic<S, H1, H2, HA>(L, tgt, ks, ka, tid);
    float cs_snap[H2 / (16 * NW)] = {};  // lane-local shift sums of the critic's state blocks, pass 1 -> pass 2
#pragma nounroll
    for (int it = 0; it < 4; ++it) {
        // fences (see opaque_zero): every global / LDS address below is rebuilt inside the pass
        const int z = opaque_zero();
        net.th += z, net.st += z, tgt.th += z, tgt.st += z, g += z, gw2 += z;
        ga = g, gc = g + L.actor_size;
        if constexpr (FUSED) {
            bulk.wo = gw2, bulk.wi = net.th;
            bulk.wt += z, bulk.m += z, bulk.v += z;
        }
        l = carve(smem + z, L, FT);
        const Net n = (it == 0) ? tgt : net;
        const float* X = (it == 0) ? l.sS2 : l.sS;
        // Layer-2 buffers by pass. Pass 3 needs the online actor's layer-2 activations of pass 2 again: they stay in bufB --
        // pass 2's critic works in bufC, its backward IN PLACE (each element is read, then overwritten, by the same
        // thread) -- and pass 3 skips the actor's second-layer GEMM (one
        // W2 read less) and output layer (tanh values are still in sT).
        // Further, passes 1 and 2 run the critic on the same states with the same weights: pass 1 snapshots the second
        // layer's accumulators after the 16 state blocks into bufC (its own backward runs in place in bufB) and pass 2
        // resumes from them with the 3 action blocks only, in place in bufC -- no state first layer, 1/6 of the GEMM
        // and of the W2 read.
        float* const aP2 = l.bufB;                                   // actor layer-2 activations (passes 0, 2; kept for 3)
        float* const cP2 = (it == 2) ? l.bufC : l.bufB;              // critic layer-2 activations
        float* const bP2 = (it == 2) ? l.bufC : l.bufB;              // activations the backward pass reads
        float* const bDZ = (it == 1) ? l.bufB : l.bufC;              // ... and the gradient it writes (in place in 1, 2)
        if (it != 1) {  // ---- actor forward (agent/model.py:26-36), parameters in `pa`
            const float* th = n.th;
            FwdPre<H2 / (16 * NW)> fp;
            fwd_prefetch<H2>(fp, th + L.aW2, th + L.ab2, H1 / 16);  // (pass 3: 13 unused loads keep the load counts path-independent)
            __builtin_amdgcn_sched_barrier(0);
            const float b3 = pa.b3;
            if (it != 3) {  // pass 3: the first-layer activations and coefficients of pass 2 are still in bufA / invA / shA --
                            // pass 2's critic only touched the action columns (resumed GEMM, action-only dX)
                if (rs0 == 0) l1p_coefs(pa.c1, l.invA, l.shA, ks);
                l1p_mfma<S>(pa.c1, X, l.bufA, LDA, 0, H1);
            }
            l2_store(pa.c2, l, H2, tid);
            lds_barrier();
            PH(1);
            if (it != 3) {
                gemm_fwd<H2, LDA, LDB, H1 / 16>(l.bufA, l.invA, l.shA, th + L.aW2, fp, aP2);
                lds_barrier();
                PH(2);
                const float z = out_layer_row(aP2, LDB, l.invB, l.shB, l.w3B, b3, H2);
                if (tid % LPR == 0) {
                    const float t = tanhf(z);
                    l.sT[tid / LPR] = t;
                    l.sA1[tid / LPR] = t * high;
                }
                lds_barrier();
                PH(3);
            }
        }
        if (it != 3) {  // ---- critic forward (agent/model.py:63-83), parameters in `pc`
            const float* th = n.th + L.actor_size;
            const float* act = (it == 1) ? l.sAct : l.sA1;
            FwdPre<H2 / (16 * NW)> fp;
            fwd_prefetch<H2>(fp, th + L.cW2, th + L.cb2, KC / 16, (it == 2) ? H1 / 16 : 0);
            __builtin_amdgcn_sched_barrier(0);
            const float b3 = pc.b3;
            if (it != 2) {  // pass 2 resumes from pass 1's state-feature sums: no state first layer
                if (rs0 == 0) l1p_coefs(pc.cs, l.invA, l.shA, ks);
                l1p_mfma<S>(pc.cs, X, l.bufA, LDA, 0, H1);
            }
            if (ra == 0) l1p_coefs(pc.ca, l.invA, l.shA, H1 + ka);
            l1p_mfma<1>(pc.ca, act, l.bufA, LDA, H1, HA);
            l2_store(pc.c2, l, H2, tid);
            __builtin_amdgcn_sched_barrier(0);
            // parameters of the next forwards (critic(net) of passes 1, 2; actor(net) of passes 2, 3)
            pc = load_critic<S, H1, H2, HA>(L, net, ks, ka, tid);
            pa = load_actor<S, H1, H2>(L, net, ks, tid);
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();
            PH(4);
            if (it == 2)
                gemm_fwd<H2, LDA, LDB, KC / 16, H1 / 16, 0>(l.bufA, l.invA, l.shA, th + L.cW2, fp, cP2, l.bufC, cs_snap);
            else
                gemm_fwd<H2, LDA, LDB, KC / 16, 0, H1 / 16>(l.bufA, l.invA, l.shA, th + L.cW2, fp, cP2,
                                                            it == 1 ? l.bufC : nullptr, cs_snap);
            lds_barrier();
            PH(5);
            const float q = out_layer_row(cP2, LDB, l.invB, l.shB, l.w3B, b3, H2);
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
        const float* wth = crit ? net.th + L.actor_size : net.th;
        float* gout = crit ? gc : ga;
        BnSet lo, hi;
        if (crit) {
            lo = {wth + L.cgs, net.st + L.cmms, net.st + L.cmvs, gc + L.cgs, gc + L.cbes, 0};
        } else {
            lo = {wth + L.ag1, net.st + L.amm1, net.st + L.amv1, ga + L.ag1, ga + L.abe1, 0};
        }
        hi = {net.th + L.actor_size + L.cga, net.st + L.cmma, net.st + L.cmva, gc + L.cga, gc + L.cbea, H1};
        const float* wdx = wth + (crit ? L.cW2 : L.aW2);
        const int dx_begin = (it == 2) ? H1 : 0, dx_end = crit ? KC : H1;
        // Fused form: weight gradient (+ update) and input gradient interleaved per 64-feature block, so that the
        // input-gradient GEMM finds its W2 rows in L2 (gemm_dw_dx; PMC FETCH_SIZE -16 %). Gradients-out form: the two
        // GEMMs back to back, the input-gradient one with its own tile prefetch (faster when nothing streams in between).
        constexpr bool kInterleave = FUSED;
        DxPre<H2> dxp;
        if (it == 2 || !kInterleave) dx_prefetch<H2>(dxp, wdx, dx_begin, dx_end, lo, hi, H1);  // consumed by gemm_dx
        __builtin_amdgcn_sched_barrier(0);
        out_layer_backward(bP2, LDB, l.invB, l.shB, l.sD, l.w3B, l.rsB, l.mmB, H2, bDZ, LDB, l.scr,
                           wg ? gout + (crit ? L.cW3 : L.aW3) : nullptr, gout + (crit ? L.cg3 : L.ag2),
                           gout + (crit ? L.cbe3 : L.abe2), sink);
        PH(it == 1 ? 7 : (it == 2 ? 12 : 15));
        if (wg) {
            col_sums(bDZ, LDB, H2, l.db, gout + (crit ? L.cb2 : L.ab2), sink);
            lds_barrier();
            PH(it == 1 ? 8 : 16);
            if constexpr (kInterleave) {
                gemm_dw_dx<H2, LDA, LDB>(l.bufA, l.invA, l.shA, crit ? KC : H1, bDZ, l.db,
                                         gw2 + (crit ? L.actor_size + L.cW2 : L.aW2), bulk, wdx, lo, hi, H1, sink);
                PH(it == 1 ? 9 : 17);
            } else {
                gemm_dw<H2, LDA, LDB>(l.bufA, l.invA, l.shA, crit ? KC : H1, bDZ, l.db,
                                      gw2 + (crit ? L.actor_size + L.cW2 : L.aW2), bulk);
                lds_barrier();
                PH(it == 1 ? 9 : 17);
                gemm_dx<H2, LDB, LDA>(bDZ, wdx, dx_begin, dx_end, l.bufA, lo, hi, H1, true, dxp, sink);
            }
        } else {
            gemm_dx<H2, LDB, LDA>(bDZ, wdx, dx_begin, dx_end, l.bufA, lo, hi, H1, false, dxp, sink);
        }
        lds_barrier();
        PH(it == 1 ? 10 : (it == 2 ? 13 : 18));
        if (it == 1) {
            dense_in_grads_k<S>(l.sS, S, l.bufA, LDA, 0, H1, gc + L.cWs, gc + L.cbs, sink);
            dense_in_grads_k<1>(l.sAct, 1, l.bufA, LDA, H1, HA, gc + L.cWa, gc + L.cba, sink);
            lds_barrier();
            PH(11);
        } else if (it == 2) {  // da1[r] = sum_j dza[r][j] * Wa[0][j]
            const float* cth = net.th + L.actor_size;
            const int rr = tid / LPR, part = tid % LPR;
            float acc = 0.f;
            for (int j = part; j < HA; j += LPR) acc = fmaf(l.bufA[rr * LDA + H1 + j], cth[L.cWa + j], acc);
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if (LPR == 8) acc += __shfl_xor(acc, 4);
            if (part == 0) l.sDa[rr] = acc;
            lds_barrier();
            PH(14);
        } else {
            dense_in_grads_k<S>(l.sS, S, l.bufA, LDA, 0, H1, ga + L.aW1, ga + L.ab1, sink);
            PH(19);
        }
    }
}

template <int S, int H1, int H2, int HA, bool FUSED>
static int launch(const avd_mlp_layout* lay, int n_agents, int set_mod, const float* theta, const float* stats,
                  float* theta_t, float* stats_t, const float* s, const float* a, const float* r, const float* s2,
                  float gamma, float high, float* grads, float* losses, UpdArgs upd, void* stream) {
    const size_t lds = sizeof(float) * learn_lds_floats(*lay, FT);
    hipError_t e = hipFuncSetAttribute((const void*)learn_kernel_t<S, H1, H2, HA, FUSED>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        set_error("avd_learn_f32: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        return AVD_E_LAUNCH;
    }
    hipLaunchKernelGGL((learn_kernel_t<S, H1, H2, HA, FUSED>), dim3(n_agents), dim3(FT), lds, (hipStream_t)stream, *lay,
                       set_mod, theta, stats, theta_t, stats_t, s, a, r, s2, gamma, high, grads, losses, upd);
    return check_launch(FUSED ? "avd_learn_update_f32" : "avd_learn_f32");
}

}  // namespace fast

// ------------------------------------------------------------------------------------------
// batch-1 forward per agent (act / Q read-out): one workgroup per agent row, weights streamed once
// ------------------------------------------------------------------------------------------
// mode 0: actor, out[agent][a] = tanh(.)*high ; mode 1: critic, out[agent][a] = q      (A = num_actions outputs)
__global__ __launch_bounds__(NTHREADS) void mlp_rows_kernel(avd_mlp_layout L, int mode, int set_mod,
                                                             const float* __restrict__ theta,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ state, int x_stride,
                                                             const float* __restrict__ action, float high,
                                                             float* __restrict__ out,
                                                             const int32_t* __restrict__ run_if_nonzero) {
    if (run_if_nonzero && *run_if_nonzero == 0) return;  // uniform across the grid: `out` already holds the result
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* h1 = smem;                    // H1 + Ha
    float* h2 = h1 + L.H1 + L.Ha;        // H2
    float* part = h2 + L.H2;             // 256
    float* xin = part + NTHREADS;        // S (<= 64) then A (<= 16)
    float* ain = xin + gen::MAX_S;
    const int agent = blockIdx.x;
    const int set = set_mod > 0 ? agent % set_mod : agent;
    const float* th = theta + (long)set * L.theta_size;
    const float* st = stats + (long)set * L.stats_size;
    if (threadIdx.x < L.S) xin[threadIdx.x] = state[(long)agent * x_stride + threadIdx.x];
    if (mode == 1 && threadIdx.x < L.A) ain[threadIdx.x] = action[(long)agent * L.A + threadIdx.x];
    __syncthreads();
    if (mode == 0) {
        gemv_relu(xin, L.S, th + L.aW1, th + L.ab1, L.H1, part, h1);
        bn_apply(h1, L.H1, th + L.ag1, th + L.abe1, st + L.amm1, st + L.amv1);
        __syncthreads();
        gemv_relu(h1, L.H1, th + L.aW2, th + L.ab2, L.H2, part, h2);
        bn_apply(h2, L.H2, th + L.ag2, th + L.abe2, st + L.amm2, st + L.amv2);
        __syncthreads();
        for (int a = 0; a < L.A; ++a) {
            const float z = block_dot(h2, th + L.aW3 + a, L.A, L.H2, part) + th[L.ab3 + a];
            if (threadIdx.x == 0) out[(long)agent * L.A + a] = tanhf(z) * high;
        }
    } else {
        const float* c = th + L.actor_size;
        gemv_relu(xin, L.S, c + L.cWs, c + L.cbs, L.H1, part, h1);
        gemv_relu(ain, L.A, c + L.cWa, c + L.cba, L.Ha, part, h1 + L.H1);
        bn_apply(h1, L.H1, c + L.cgs, c + L.cbes, st + L.cmms, st + L.cmvs);
        bn_apply(h1 + L.H1, L.Ha, c + L.cga, c + L.cbea, st + L.cmma, st + L.cmva);
        __syncthreads();
        gemv_relu(h1, L.H1 + L.Ha, c + L.cW2, c + L.cb2, L.H2, part, h2);
        bn_apply(h2, L.H2, c + L.cg3, c + L.cbe3, st + L.cmm3, st + L.cmv3);
        __syncthreads();
        for (int a = 0; a < L.A; ++a) {
            const float q = block_dot(h2, c + L.cW3 + a, L.A, L.H2, part) + c[L.cb3 + a];
            if (threadIdx.x == 0) out[(long)agent * L.A + a] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The actor forward for agents that SHARE weight sets (set_mod > 0: interfrl with shared sets, trainer.py:121-128): one
// workgroup evaluates R agents of the same set, so the set's 143 KB of weights are read once per R rows instead of once
// per row (20 480 agents x 143 KB = 2.9 GB of L2 reads per step as batch-1 GEMVs). Every row goes through EXACTLY
// mlp_rows_kernel's sequence of operations (same k split, same partial-sum order, same reduction tree), so the result
// is bit-identical to the batch-1 kernel: tests/test_gpu_trainer.py::test_shared_sets_equal_per_agent_sets_under_interfrl.
// ------------------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void gemv_relu_rows(const float* x, int ldx, int K, const float* __restrict__ W,
                                               const float* __restrict__ b, int N, float* part, float* y, int ldy) {
    const int cols = N < NTHREADS ? N : NTHREADS;
    const int ksplit = NTHREADS / cols;
    for (int n0 = 0; n0 < N; n0 += cols) {
        const int n = n0 + (threadIdx.x % cols);
        const int kh = threadIdx.x / cols;
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        if (kh < ksplit && n < N) {
            const int kb = (K * kh) / ksplit, ke = (K * (kh + 1)) / ksplit;
#pragma unroll 4
            for (int k = kb; k < ke; ++k) {
                const float w = W[(long)k * N + n];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = fmaf(x[r * ldx + k], w, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) part[r * NTHREADS + threadIdx.x] = acc[r];
        __syncthreads();
        if (threadIdx.x < cols && n < N) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float sum = b[n];
                for (int h = 0; h < ksplit; ++h) sum += part[r * NTHREADS + h * cols + threadIdx.x];
                y[r * ldy + n] = fmaxf(sum, 0.f);
            }
        }
        __syncthreads();
    }
}

template <int R>
__global__ __launch_bounds__(NTHREADS) void actor_rows_shared_kernel(avd_mlp_layout L, int set_mod, int n_agents,
                                                                      const float* __restrict__ theta,
                                                                      const float* __restrict__ stats,
                                                                      const float* __restrict__ state, int x_stride, float high,
                                                                      float* __restrict__ out,
                                                                      const int32_t* __restrict__ run_if_nonzero) {
    if (run_if_nonzero && *run_if_nonzero == 0) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* h1 = smem;                        // [R][H1]
    float* h2 = h1 + R * L.H1;               // [R][H2]
    float* part = h2 + R * L.H2;             // [R][NTHREADS]
    float* xin = part + R * NTHREADS;        // [R][MAX_S]
    const int set = blockIdx.x % set_mod, g = blockIdx.x / set_mod, P = n_agents / set_mod;
    const float* th = theta + (long)set * L.theta_size;
    const float* st = stats + (long)set * L.stats_size;
    // row r = agent (R g + r) * set_mod + set; rows past the last platoon repeat the last one and are not stored
    for (int i = threadIdx.x; i < R * L.S; i += NTHREADS) {
        const int r = i / L.S, k = i - r * L.S, pl = min(R * g + r, P - 1);
        xin[r * gen::MAX_S + k] = state[((long)pl * set_mod + set) * x_stride + k];
    }
    __syncthreads();
    gemv_relu_rows<R>(xin, gen::MAX_S, L.S, th + L.aW1, th + L.ab1, L.H1, part, h1, L.H1);
#pragma unroll
    for (int r = 0; r < R; ++r) bn_apply(h1 + r * L.H1, L.H1, th + L.ag1, th + L.abe1, st + L.amm1, st + L.amv1);
    __syncthreads();
    gemv_relu_rows<R>(h1, L.H1, L.H1, th + L.aW2, th + L.ab2, L.H2, part, h2, L.H2);
#pragma unroll
    for (int r = 0; r < R; ++r) bn_apply(h2 + r * L.H2, L.H2, th + L.ag2, th + L.abe2, st + L.amm2, st + L.amv2);
    __syncthreads();
    for (int a = 0; a < L.A; ++a) {
        for (int r = 0; r < R; ++r) {
            const float z = block_dot(h2 + r * L.H2, th + L.aW3 + a, L.A, L.H2, part) + th[L.ab3 + a];
            const int pl = R * g + r;
            if (threadIdx.x == 0 && pl < P) out[((long)pl * set_mod + set) * L.A + a] = tanhf(z) * high;
        }
    }
}

#ifdef AVD_PHASE_TIMING
}  // namespace avd
extern "C" __attribute__((visibility("default"))) int avd_debug_phase_cycles(unsigned long long* h_out, int reset) {
    if (h_out) (void)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(avd::g_phase_cycles), sizeof(unsigned long long) * 32);
    if (reset) {
        unsigned long long z[32] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(avd::g_phase_cycles), z, sizeof(z));
    }
    return 0;
}
namespace avd {
#endif

static inline int round4(int x) { return (x + 3) & ~3; }

}  // namespace avd

using namespace avd;

extern "C" int avd_mlp_layout_init(avd_mlp_layout* o, int S, int A, int H1, int H2, int Ha, int B) {
    AVD_REQUIRE(o, "avd_mlp_layout_init: null");
    AVD_REQUIRE(S > 0 && A > 0 && H1 > 0 && H2 > 0 && Ha > 0 && B > 0, "avd_mlp_layout_init: non-positive dimension");
    memset(o, 0, sizeof(*o));
    o->S = S, o->A = A, o->H1 = H1, o->H2 = H2, o->Ha = Ha, o->B = B;
    int p = 0;
    auto take = [&](int n) {
        const int at = p;
        p += round4(n);
        return at;
    };
    o->aW1 = take(S * H1), o->ab1 = take(H1), o->ag1 = take(H1), o->abe1 = take(H1);
    o->aW2 = take(H1 * H2), o->ab2 = take(H2), o->ag2 = take(H2), o->abe2 = take(H2);
    o->aW3 = take(H2 * A), o->ab3 = take(A);
    o->actor_size = p;
    p = 0;  // critic offsets are relative to the critic block
    o->cWs = take(S * H1), o->cbs = take(H1), o->cgs = take(H1), o->cbes = take(H1);
    o->cWa = take(A * Ha), o->cba = take(Ha), o->cga = take(Ha), o->cbea = take(Ha);
    o->cW2 = take((H1 + Ha) * H2), o->cb2 = take(H2), o->cg3 = take(H2), o->cbe3 = take(H2);
    o->cW3 = take(H2 * A), o->cb3 = take(A);
    o->theta_size = o->actor_size + p;
    p = 0;
    o->amm1 = take(H1), o->amv1 = take(H1), o->amm2 = take(H2), o->amv2 = take(H2);
    o->cmms = take(H1), o->cmvs = take(H1), o->cmma = take(Ha), o->cmva = take(Ha);
    o->cmm3 = take(H2), o->cmv3 = take(H2);
    o->stats_size = p;
    return AVD_OK;
}

// Which kernel serves the reference widths: learn_kernel_l (lean.hip, two workgroups per CU) unless
// AVD_LEARN_KERNEL=fast asks for learn_kernel_t (one workgroup per CU, first-layer activations in LDS).
static bool use_lean_kernel() {
    const char* k = AVD_DIAG_ENV("LEARN_KERNEL");
    return !(k && !strcmp(k, "fast"));
}

static int check_mlp_dims(const avd_mlp_layout* L, const char* who, bool rows_only = false) {
    AVD_REQUIRE(L, "%s: null layout", who);
    if (rows_only) {  // batch-1 forward: plain GEMVs, any widths
        if (L->A > gen::MAX_A || L->S > gen::MAX_S) {
            set_error("%s: need S <= %d and A <= %d (got S=%d A=%d)", who, gen::MAX_S, gen::MAX_A, L->S, L->A);
            return AVD_E_UNSUPPORTED;
        }
        return AVD_OK;
    }
    if (L->A > gen::MAX_A || L->S > gen::MAX_S || (L->H1 % 16) || (L->Ha % 16) || (L->H2 % 32) || L->H2 > 16 * DX_NB) {
        set_error("%s: need S <= %d, A <= %d, H1 and Ha multiples of 16, H2 a multiple of 32 and <= %d (got S=%d A=%d "
                  "H1=%d H2=%d Ha=%d); pad the widths with zero units",
                  who, gen::MAX_S, gen::MAX_A, 16 * DX_NB, L->S, L->A, L->H1, L->H2, L->Ha);
        return AVD_E_UNSUPPORTED;
    }
    return AVD_OK;
}

static int launch_rows(const avd_mlp_layout* lay, int mode, int n_agents, int set_mod, const float* theta,
                       const float* stats, const float* state, int x_stride, const float* action, float high,
                       float* out, void* stream, const char* who, const int32_t* run_if_nonzero = nullptr) {
    int rc = check_mlp_dims(lay, who, true);
    if (rc) return rc;
    AVD_REQUIRE(n_agents > 0 && set_mod >= 0 && x_stride >= lay->S, "%s: n_agents=%d set_mod=%d x_stride=%d", who,
                n_agents, set_mod, x_stride);
    AVD_REQUIRE(theta && stats && state && out && (mode == 0 || action), "%s: null pointer", who);
    constexpr int R = 8;  // shared sets: R agents of a set per workgroup (weights read once per R rows, same bits per row)
    if (mode == 0 && set_mod > 0 && n_agents % set_mod == 0 && n_agents / set_mod >= R) {
        const size_t ldr = sizeof(float) * (size_t)R * (lay->H1 + lay->H2 + NTHREADS + gen::MAX_S);
        if (ldr <= 64 * 1024) {
            const int P = n_agents / set_mod;
            hipLaunchKernelGGL(actor_rows_shared_kernel<R>, dim3((unsigned)((P + R - 1) / R) * set_mod), dim3(NTHREADS), ldr,
                               (hipStream_t)stream, *lay, set_mod, n_agents, theta, stats, state, x_stride, high, out,
                               run_if_nonzero);
            return check_launch(who);
        }
    }
    const size_t lds = sizeof(float) * (size_t)(lay->H1 + lay->Ha + lay->H2 + NTHREADS + gen::MAX_S + gen::MAX_A);
    if (lds > 160 * 1024) {
        set_error("%s: hidden sizes need %zu B of LDS (> 160 KiB)", who, lds);
        return AVD_E_UNSUPPORTED;
    }
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)mlp_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mlp_rows_kernel, dim3(n_agents), dim3(NTHREADS), lds, (hipStream_t)stream, *lay, mode, set_mod,
                       theta, stats, state, x_stride, action, high, out, run_if_nonzero);
    return check_launch(who);
}

extern "C" int avd_actor_forward_f32(const avd_mlp_layout* lay, int n_agents, int set_mod, const float* theta,
                                     const float* stats, const float* state, int x_stride, float high, float* out,
                                     void* stream) {
    return launch_rows(lay, 0, n_agents, set_mod, theta, stats, state, x_stride, nullptr, high, out, stream,
                       "avd_actor_forward_f32");
}

extern "C" int avd_actor_forward_cond_f32(const avd_mlp_layout* lay, int n_agents, int set_mod, const float* theta,
                                          const float* stats, const float* state, int x_stride, float high, float* out,
                                          const int32_t* run_if_nonzero, void* stream) {
    AVD_REQUIRE(run_if_nonzero, "avd_actor_forward_cond_f32: null flag");
    return launch_rows(lay, 0, n_agents, set_mod, theta, stats, state, x_stride, nullptr, high, out, stream,
                       "avd_actor_forward_cond_f32", run_if_nonzero);
}

extern "C" int avd_critic_forward_f32(const avd_mlp_layout* lay, int n_agents, int set_mod, const float* theta,
                                      const float* stats, const float* state, int x_stride, const float* action,
                                      float* q, void* stream) {
    return launch_rows(lay, 1, n_agents, set_mod, theta, stats, state, x_stride, action, 0.f, q, stream,
                       "avd_critic_forward_f32");
}

extern "C" int avd_learn_f32(const avd_mlp_layout* lay, int n_agents, int set_mod, const float* theta,
                             const float* stats, const float* theta_t, const float* stats_t, const float* s,
                             const float* a, const float* r, const float* s2, float gamma, float high, float* grads,
                             float* losses, void* stream) {
    int rc = check_mlp_dims(lay, "avd_learn_f32");
    if (rc) return rc;
    if (lay->B != TILE) {
        set_error("avd_learn_f32: batch_size=%d; the tile kernel implements B == %d", lay->B, TILE);
        return AVD_E_UNSUPPORTED;
    }
    AVD_REQUIRE(n_agents > 0 && set_mod >= 0, "avd_learn_f32: n_agents=%d set_mod=%d", n_agents, set_mod);
    AVD_REQUIRE(theta && stats && theta_t && stats_t && s && a && r && s2 && grads, "avd_learn_f32: null pointer");
    // reference widths (src/config.py:112-117) take the dimension-specialised kernel; anything else the general one
    if (lay->A == 1 && lay->H1 == 256 && lay->H2 == 128 && lay->Ha == 48 && (lay->S == 3 || lay->S == 4) &&
        !AVD_DIAG_ENV("LEARN_GENERAL")) {
        const UpdArgs none = {};
        if (use_lean_kernel())
            return lean_launch(lay, false, n_agents, set_mod, theta, stats, (float*)theta_t, (float*)stats_t, s, a, r, s2,
                               gamma, high, grads, losses, none, stream);
        if (lay->S == 4)
            return fast::launch<4, 256, 128, 48, false>(lay, n_agents, set_mod, theta, stats, (float*)theta_t,
                                                        (float*)stats_t, s, a, r, s2, gamma, high, grads, losses, none,
                                                        stream);
        return fast::launch<3, 256, 128, 48, false>(lay, n_agents, set_mod, theta, stats, (float*)theta_t,
                                                    (float*)stats_t, s, a, r, s2, gamma, high, grads, losses, none,
                                                    stream);
    }
    // the centralized framework's shapes (S = 4 L, A = L, widths x 1.2) at L = 3 / 5: their own eight-wave kernel (cen.hip)
    if (cen_supports(lay) && !AVD_DIAG_ENV("LEARN_GENERAL"))
        return cen_launch(lay, n_agents, set_mod, theta, stats, theta_t, stats_t, s, a, r, s2, gamma, high, grads, losses, stream);
    const size_t lds = sizeof(float) * gen::lds_floats(*lay);
    if (lds > 160 * 1024) {
        set_error("avd_learn_f32: S=%d A=%d H1=%d H2=%d Ha=%d need %zu B of LDS per 64-row tile (> 160 KiB)", lay->S,
                  lay->A, lay->H1, lay->H2, lay->Ha, lds);
        return AVD_E_UNSUPPORTED;
    }
    hipError_t e = hipFuncSetAttribute((const void*)gen::learn_kernel_g<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) {
        set_error("avd_learn_f32: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        return AVD_E_LAUNCH;
    }
    hipLaunchKernelGGL(gen::learn_kernel_g<false>, dim3(n_agents), dim3(NTHREADS), lds, (hipStream_t)stream, *lay, set_mod,
                       theta, stats, (float*)theta_t, (float*)stats_t, s, a, r, s2, gamma, high, grads, losses, UpdArgs{});
    return check_launch("avd_learn_f32");
}

// optim.hip: Adam+Polyak over the elements of each slab that lie OUTSIDE [skip_a0, skip_a1) and [skip_c0, skip_c1)
int launch_adam_polyak_ranges(const avd_mlp_layout* lay, int n_sets, const float* theta_in, float* theta_out,
                              float* theta_t, float* m, float* v, const float* grads, const int32_t* step,
                              float actor_lr, float critic_lr, double tau, int skip_a0, int skip_a1, int skip_c0,
                              int skip_c1, void* stream);

static int learn_update_impl(const avd_mlp_layout* lay, int n_agents, const float* theta, const float* stats,
                             float* theta_out, float* theta_t, float* stats_t, float* m, float* v, const int32_t* step,
                             const float* s, const float* a, const float* r, const float* s2, float gamma, float high,
                             float actor_lr, float critic_lr, double tau, float* grads_scratch, float* losses,
                             const float* next_state, int x_stride, float* next_action, void* stream);

extern "C" int avd_learn_update_f32(const avd_mlp_layout* lay, int n_agents, const float* theta, const float* stats,
                                    float* theta_out, float* theta_t, float* stats_t, float* m, float* v,
                                    const int32_t* step, const float* s, const float* a, const float* r,
                                    const float* s2, float gamma, float high, float actor_lr, float critic_lr,
                                    double tau, float* grads_scratch, float* losses, void* stream) {
    return learn_update_impl(lay, n_agents, theta, stats, theta_out, theta_t, stats_t, m, v, step, s, a, r, s2, gamma, high,
                             actor_lr, critic_lr, tau, grads_scratch, losses, nullptr, 0, nullptr, stream);
}

extern "C" int avd_learn_update_plan(const avd_mlp_layout* lay, int n_agents, int* chunk_agents, int* n_chunks, int* update_groups) {
    AVD_REQUIRE(lay && n_agents > 0 && chunk_agents && n_chunks && update_groups, "avd_learn_update_plan: null / n_agents=%d", n_agents);
    if (cen_supports(lay) && !AVD_DIAG_ENV("LEARN_GENERAL")) {
        cen_update_plan(n_agents, chunk_agents, update_groups);
        *n_chunks = (n_agents + *chunk_agents - 1) / *chunk_agents;
    } else {  // one launch over all agents, the update applied inside it
        *chunk_agents = n_agents, *n_chunks = 1, *update_groups = 0;
    }
    return AVD_OK;
}

extern "C" int avd_learn_update_act_f32(const avd_mlp_layout* lay, int n_agents, const float* theta, const float* stats,
                                        float* theta_out, float* theta_t, float* stats_t, float* m, float* v,
                                        const int32_t* step, const float* s, const float* a, const float* r,
                                        const float* s2, float gamma, float high, float actor_lr, float critic_lr,
                                        double tau, float* grads_scratch, float* losses, const float* next_state,
                                        int x_stride, float* next_action, void* stream) {
    AVD_REQUIRE(next_state && next_action && lay && x_stride >= lay->S, "avd_learn_update_act_f32: next_state / x_stride");
    return learn_update_impl(lay, n_agents, theta, stats, theta_out, theta_t, stats_t, m, v, step, s, a, r, s2, gamma, high,
                             actor_lr, critic_lr, tau, grads_scratch, losses, next_state, x_stride, next_action, stream);
}

static int learn_update_impl(const avd_mlp_layout* lay, int n_agents, const float* theta, const float* stats,
                             float* theta_out, float* theta_t, float* stats_t, float* m, float* v, const int32_t* step,
                             const float* s, const float* a, const float* r, const float* s2, float gamma, float high,
                             float actor_lr, float critic_lr, double tau, float* grads_scratch, float* losses,
                             const float* next_state, int x_stride, float* next_action, void* stream) {
    int rc = check_mlp_dims(lay, "avd_learn_update_f32");
    if (rc) return rc;
    AVD_REQUIRE(n_agents > 0, "avd_learn_update_f32: n_agents=%d", n_agents);
    AVD_REQUIRE(theta && stats && theta_out && theta_t && stats_t && m && v && step && s && a && r && s2 && grads_scratch,
                "avd_learn_update_f32: null pointer");
    AVD_REQUIRE(theta_out != theta, "avd_learn_update_f32: theta_out must not alias theta (every pass reads pre-update weights)");
    const int a0 = lay->aW2, a1 = lay->aW2 + lay->H1 * lay->H2;  // the two W2 matrices: updated inside the learn kernels
    const int c0 = lay->actor_size + lay->cW2, c1 = c0 + (lay->H1 + lay->Ha) * lay->H2;
    if (lay->B != TILE) {
        set_error("avd_learn_update_f32: batch_size=%d; the tile kernels implement B == %d", lay->B, TILE);
        return AVD_E_UNSUPPORTED;
    }
    if (!(lay->A == 1 && lay->H1 == 256 && lay->H2 == 128 && lay->Ha == 48 && (lay->S == 3 || lay->S == 4)) ||
        AVD_DIAG_ENV("LEARN_GENERAL")) {
        // any other shape the general kernel serves (centralized framework, non-default widths): its fused form
        const size_t lds = sizeof(float) * gen::lds_floats(*lay);
        if (lds > 160 * 1024) {
            set_error("avd_learn_update_f32: S=%d A=%d H1=%d H2=%d Ha=%d need %zu B of LDS per 64-row tile (> 160 KiB)", lay->S,
                      lay->A, lay->H1, lay->H2, lay->Ha, lds);
            return AVD_E_UNSUPPORTED;
        }
        hipError_t e = hipFuncSetAttribute((const void*)gen::learn_kernel_g<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) {
            set_error("avd_learn_update_f32: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
            return AVD_E_LAUNCH;
        }
        const UpdArgs updg = {theta_out, m, v, step, actor_lr, critic_lr, (float)tau, (float)(1.0 - tau), nullptr, 0, nullptr};
        if (cen_supports(lay) && !AVD_DIAG_ENV("LEARN_GENERAL")) {
            // the centralized shapes: chunked learn + whole-row Adam / Polyak passes on two streams (cen.hip)
            rc = cen_launch_update(lay, n_agents, theta, stats, theta_out, theta_t, stats_t, m, v, step, s, a, r, s2, gamma, high, actor_lr,
                                   critic_lr, tau, grads_scratch, losses, stream);
            if (rc || !next_action) return rc;
            return launch_rows(lay, 0, n_agents, 0, theta_out, stats, next_state, x_stride, nullptr, high, next_action, stream,
                               "avd_learn_update_act_f32(actor)");
        }
        hipLaunchKernelGGL(gen::learn_kernel_g<true>, dim3(n_agents), dim3(NTHREADS), lds, (hipStream_t)stream, *lay, 0, theta,
                           stats, theta_t, stats_t, s, a, r, s2, gamma, high, grads_scratch, losses, updg);
        rc = check_launch("avd_learn_update_f32 (general)");
        if (rc) return rc;
        rc = launch_adam_polyak_ranges(lay, n_agents, theta, theta_out, theta_t, m, v, grads_scratch, step, actor_lr,
                                       critic_lr, tau, a0, a1, c0, c1, stream);
        if (rc || !next_action) return rc;
        return launch_rows(lay, 0, n_agents, 0, theta_out, stats, next_state, x_stride, nullptr, high, next_action, stream,
                           "avd_learn_update_act_f32(actor)");
    }
    UpdArgs upd = {theta_out, m, v, step, actor_lr, critic_lr, (float)tau, (float)(1.0 - tau), nullptr, 0, nullptr};
    if (use_lean_kernel()) {
        // learn_kernel_l applies the small tensors' update itself and, on request, evaluates the updated actor on the
        // agent's next state while its weights are still in L2: one launch for the whole update
        upd.act_x = next_state, upd.act_x_stride = x_stride, upd.act_out = next_action;
        return lean_launch(lay, true, n_agents, 0, theta, stats, theta_t, stats_t, s, a, r, s2, gamma, high, grads_scratch,
                           losses, upd, stream);
    }
    if (lay->S == 4)
        rc = fast::launch<4, 256, 128, 48, true>(lay, n_agents, 0, theta, stats, theta_t, stats_t, s, a, r, s2, gamma,
                                                 high, grads_scratch, losses, upd, stream);
    else
        rc = fast::launch<3, 256, 128, 48, true>(lay, n_agents, 0, theta, stats, theta_t, stats_t, s, a, r, s2, gamma,
                                                 high, grads_scratch, losses, upd, stream);
    if (rc) return rc;
    // the small tensors: everything outside the two W2 matrices (which the learn kernel has already updated)
    rc = launch_adam_polyak_ranges(lay, n_agents, theta, theta_out, theta_t, m, v, grads_scratch, step, actor_lr, critic_lr,
                                   tau, a0, a1, c0, c1, stream);
    if (rc || !next_action) return rc;
    return launch_rows(lay, 0, n_agents, 0, theta_out, stats, next_state, x_stride, nullptr, high, next_action, stream,
                       "avd_learn_update_act_f32(actor)");
}
