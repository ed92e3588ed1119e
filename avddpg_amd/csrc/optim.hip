// Optimiser / target-network / federation kernels for gfx950 (K11, K12, K13 local part).
//
// All of these are elementwise, HBM-bound streams over flat float32 slabs (16-byte accesses,
// grid-stride).  Adam follows TF 2.4.1's ApplyAdam functor exactly and is compiled without FMA
// contraction so the result is bit-identical to the float32 oracle (oracle/mlp.py:adam_update).
#include <stdlib.h>

#include "common.h"

namespace avd {

constexpr float ADAM_B1 = 0.9f, ADAM_B2 = 0.999f, ADAM_EPS = 1e-7f;

// A set whose gradient slab is not finite at the head of its actor block or of its critic block (the set learners of fset.hip /
// fsplit.hip turn a non-finite input or an fp16 overflow into an ALL-NaN block: finalize_*) -- the guarded update leaves such a set
// untouched.
// CONTRACT (finalize_* of fset.hip / fsplit.hip): element 0 of a set's actor block and of its critic block is non-finite iff the
// block is invalid -- every failure the learners detect goes through their `bad` flag, which makes finalize write NaN over the whole
// slab. An Inf head (an overflow that reached the sums without tripping a watch) is treated the same.
__device__ __forceinline__ bool slab_is_nan(const float* g, int actor_size) {
    const unsigned a = __float_as_uint(g[0]) & 0x7fffffffu, c = __float_as_uint(g[actor_size]) & 0x7fffffffu;
    return a >= 0x7f800000u || c >= 0x7f800000u;
}

// grid: (blocks over theta_size/4, n_sets). GUARD: avd_adam_polyak_guarded_f32 -- a set with a NaN gradient slab takes no step at all
// (weights, moments, targets untouched), its Adam iteration count (already advanced by the caller) is put back and *skipped counts it.
template <bool GUARD>
__global__ __launch_bounds__(256) void adam_polyak_kernel(int theta_size, int actor_size, float4* __restrict__ theta,
                                                          float4* __restrict__ theta_t, float4* __restrict__ m,
                                                          float4* __restrict__ v, const float4* __restrict__ grads,
                                                          int32_t* __restrict__ step, float actor_lr,
                                                          float critic_lr, float tau, float omt, int32_t* __restrict__ skipped) {
#pragma clang fp contract(off)
    const int set = blockIdx.y;
    const int t = step[set];
    if (GUARD && slab_is_nan((const float*)grads + (long)set * theta_size, actor_size)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            step[set] = t - 1;  // (every thread of the set returns here: nobody uses the count)
            if (skipped) atomicAdd(skipped, 1);
        }
        return;
    }
    // beta^t as float32(pow) like the oracle / TF (math_ops.pow on float32 scalars)
    const float b1p = (float)pow((double)ADAM_B1, (double)t);
    const float b2p = (float)pow((double)ADAM_B2, (double)t);
    const float root = sqrtf(1.0f - b2p);
    const float alpha_a = (actor_lr * root) / (1.0f - b1p);
    const float alpha_c = (critic_lr * root) / (1.0f - b1p);
    const int n4 = theta_size / 4;
    const long base = (long)set * n4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const float alpha = (i * 4 < actor_size) ? alpha_a : alpha_c;  // blocks are 4-float aligned
        float4 w = theta[base + i], wt = theta_t[base + i], mm = m[base + i], vv = v[base + i];
        const float4 g = grads[base + i];
        float* wp = &w.x;
        float* tp = &wt.x;
        float* mp = &mm.x;
        float* vp = &vv.x;
        const float* gp = &g.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mp[k] = mp[k] + (gp[k] - mp[k]) * (1.0f - ADAM_B1);
            vp[k] = vp[k] + (gp[k] * gp[k] - vp[k]) * (1.0f - ADAM_B2);
            wp[k] = wp[k] - (mp[k] * alpha) / (sqrtf(vp[k]) + ADAM_EPS);
            tp[k] = wp[k] * tau + tp[k] * omt;  // update_target on the freshly updated weight
        }
        theta[base + i] = w;
        theta_t[base + i] = wt;
        m[base + i] = mm;
        v[base + i] = vv;
    }
}

// Same arithmetic for the float4 groups of each slab OUTSIDE two skip ranges, reading the pre-update weights from
// theta_in and writing theta_out (second half of avd_learn_update_f32: the learn kernel itself updates the two W2
// matrices in its weight-gradient epilogues; this pass covers the ~6 % of small tensors).
__global__ __launch_bounds__(256) void adam_polyak_ranges_kernel(int theta_size, int actor_size,
                                                                 const float4* __restrict__ theta_in,
                                                                 float4* __restrict__ theta_out,
                                                                 float4* __restrict__ theta_t, float4* __restrict__ m,
                                                                 float4* __restrict__ v,
                                                                 const float4* __restrict__ grads,
                                                                 const int32_t* __restrict__ step, float actor_lr,
                                                                 float critic_lr, float tau, float omt, int a0, int a1,
                                                                 int c0, int c1) {
#pragma clang fp contract(off)
    const int set = blockIdx.y;
    const int t = step[set];
    const float b1p = (float)pow((double)ADAM_B1, (double)t);
    const float b2p = (float)pow((double)ADAM_B2, (double)t);
    const float root = sqrtf(1.0f - b2p);
    const float alpha_a = (actor_lr * root) / (1.0f - b1p);
    const float alpha_c = (critic_lr * root) / (1.0f - b1p);
    // compact index over the kept float4 groups: [0,a0) [a1,c0) [c1,theta_size)
    const int n0 = a0 / 4, n1 = (c0 - a1) / 4, n2 = (theta_size - c1) / 4;
    const long base = (long)set * (theta_size / 4);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n0 + n1 + n2; j += gridDim.x * blockDim.x) {
        const int i = j < n0 ? j : (j < n0 + n1 ? a1 / 4 + (j - n0) : c1 / 4 + (j - n0 - n1));
        const float alpha = (i * 4 < actor_size) ? alpha_a : alpha_c;
        float4 w = theta_in[base + i], wt = theta_t[base + i], mm = m[base + i], vv = v[base + i];
        const float4 g = grads[base + i];
        float* wp = &w.x;
        float* tp = &wt.x;
        float* mp = &mm.x;
        float* vp = &vv.x;
        const float* gp = &g.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mp[k] = mp[k] + (gp[k] - mp[k]) * (1.0f - ADAM_B1);
            vp[k] = vp[k] + (gp[k] * gp[k] - vp[k]) * (1.0f - ADAM_B2);
            wp[k] = wp[k] - (mp[k] * alpha) / (sqrtf(vp[k]) + ADAM_EPS);
            tp[k] = wp[k] * tau + tp[k] * omt;
        }
        theta_out[base + i] = w;
        theta_t[base + i] = wt;
        m[base + i] = mm;
        v[base + i] = vv;
    }
}

// The same update of WHOLE slab rows, theta_in -> theta_out, as a small persistent grid: workgroup b walks rows b, b + gridDim.x, ..
// with AR_UNR float4 groups of each of the five arrays in flight per thread. Made to run BESIDE a compute kernel that holds every CU's
// LDS (cen.hip: the centralized learn kernel's chunks): a few hundred long-lived workgroups -- one or two per CU, ~100 registers -- keep
// HBM busy without flooding each CU's wave slots and memory queue the way the row-per-blockIdx.y grid above does. Every access is
// non-temporal: the rows stream through once, and without the hint they evict the weights the learn workgroups keep re-reading
// (4096 x 5 centralized, same box: 7.29 -> 6.68 ms per step).
#ifndef AR_UNR_N
#define AR_UNR_N 4
#endif
constexpr int AR_UNR = AR_UNR_N;
typedef float v4f __attribute__((ext_vector_type(4)));
#define NTL(p) ({ const v4f t_ = __builtin_nontemporal_load((const v4f*)(p)); make_float4(t_[0], t_[1], t_[2], t_[3]); })
#define NTS(val_, p) __builtin_nontemporal_store((v4f){(val_).x, (val_).y, (val_).z, (val_).w}, (v4f*)(p))
__global__ __launch_bounds__(256) void adam_polyak_rows_kernel(int theta_size, int actor_size, int n_sets,
                                                               const float4* __restrict__ theta_in, float4* __restrict__ theta_out,
                                                               float4* __restrict__ theta_t, float4* __restrict__ m,
                                                               float4* __restrict__ v, const float4* __restrict__ grads,
                                                               const int32_t* __restrict__ step, float actor_lr, float critic_lr,
                                                               float tau, float omt) {
#pragma clang fp contract(off)
    const int n4 = theta_size / 4;
    for (int set = blockIdx.x; set < n_sets; set += gridDim.x) {
        const int t = step[set];
        const float b1p = (float)pow((double)ADAM_B1, (double)t);
        const float b2p = (float)pow((double)ADAM_B2, (double)t);
        const float root = sqrtf(1.0f - b2p);
        const float alpha_a = (actor_lr * root) / (1.0f - b1p);
        const float alpha_c = (critic_lr * root) / (1.0f - b1p);
        const long base = (long)set * n4;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 256 * AR_UNR) {
            float4 w[AR_UNR], wt[AR_UNR], mm[AR_UNR], vv[AR_UNR], g[AR_UNR];
#pragma unroll
            for (int u = 0; u < AR_UNR; ++u) {
                const int i = min(i0 + 256 * u, n4 - 1);
                w[u] = NTL(theta_in + base + i), wt[u] = NTL(theta_t + base + i), mm[u] = NTL(m + base + i), vv[u] = NTL(v + base + i);
                g[u] = NTL(grads + base + i);
            }
#pragma unroll
            for (int u = 0; u < AR_UNR; ++u) {
                const int i = i0 + 256 * u;
                if (i >= n4) break;
                const float alpha = (i * 4 < actor_size) ? alpha_a : alpha_c;
                float* wp = &w[u].x;
                float* tp = &wt[u].x;
                float* mp = &mm[u].x;
                float* vp = &vv[u].x;
                const float* gp = &g[u].x;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    mp[k] = mp[k] + (gp[k] - mp[k]) * (1.0f - ADAM_B1);
                    vp[k] = vp[k] + (gp[k] * gp[k] - vp[k]) * (1.0f - ADAM_B2);
                    wp[k] = wp[k] - (mp[k] * alpha) / (sqrtf(vp[k]) + ADAM_EPS);
                    tp[k] = wp[k] * tau + tp[k] * omt;
                }
                NTS(w[u], theta_out + base + i), NTS(wt[u], theta_t + base + i), NTS(mm[u], m + base + i), NTS(vv[u], v + base + i);
            }
        }
    }
}

#undef NTL
#undef NTS

// intrafrl (workers/trainer.py:189-190, 341-342, 417-431): the M agents of platoon p all step with the MEAN of their M gradients
// (src/server/federated.py:47-63; weighted :99-118) -- averaged HERE, where the gradients are consumed: grid (blocks over theta_size / 4,
// P); a thread loads its float4 group of the platoon's M gradient rows once, forms the mean in fed_sum_kernel's order (four partial
// sums over the members i = ph, ph + 4, .., combined in phase order) and fed_finalize_kernel's scaling, and applies Adam + Polyak
// to the M agents' weights with it. Replaces fed_sum + fed_finalize + fed_scatter + adam_polyak (r05: the 6.3 GB gradient slab of
// 4096 x 5 agents was read twice and written twice; now it is read once). lead_skip: the lead vehicle of every platoon takes no
// step at all under intra_directional_averaging (:417-418) -- a predicate, its gradient still enters the mean.
constexpr int INTRA_MAX_M = 16;
__global__ __launch_bounds__(256) void adam_polyak_intra_kernel(int theta_size, int actor_size, int M, int lead_skip,
                                                                float4* __restrict__ theta, float4* __restrict__ theta_t,
                                                                float4* __restrict__ m, float4* __restrict__ v,
                                                                const float4* __restrict__ grads, const int32_t* __restrict__ step,
                                                                const float* __restrict__ weights, float actor_lr, float critic_lr,
                                                                float tau, float omt) {
#pragma clang fp contract(off)
    __shared__ float s_aa[INTRA_MAX_M], s_ac[INTRA_MAX_M], s_w[INTRA_MAX_M], s_scale;
    const int p = blockIdx.y, n4 = theta_size / 4;
    const long set0 = (long)p * M;
    if (threadIdx.x < M) {
        const int t = step[set0 + threadIdx.x];
        const float b1p = (float)pow((double)ADAM_B1, (double)t), b2p = (float)pow((double)ADAM_B2, (double)t);
        const float root = sqrtf(1.0f - b2p);
        s_aa[threadIdx.x] = (actor_lr * root) / (1.0f - b1p);
        s_ac[threadIdx.x] = (critic_lr * root) / (1.0f - b1p);
        s_w[threadIdx.x] = weights ? weights[set0 + threadIdx.x] : 1.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ws = 0.f;
        for (int i = 0; i < M; ++i) ws += s_w[i];  // (fed_sum_kernel's wsum: ascending)
        s_scale = 1.0f / ws;
    }
    __syncthreads();
    for (int i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += gridDim.x * blockDim.x) {
        float4 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < M; ++i) {
            const float4 g = grads[(set0 + i) * n4 + i4];
            const float w = s_w[i];
            float4& a = acc[i & 3];
            a.x = __builtin_fmaf(w, g.x, a.x), a.y = __builtin_fmaf(w, g.y, a.y), a.z = __builtin_fmaf(w, g.z, a.z), a.w = __builtin_fmaf(w, g.w, a.w);
        }
        float4 r = acc[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) r.x += acc[k].x, r.y += acc[k].y, r.z += acc[k].z, r.w += acc[k].w;
        float gm[4];
        if (weights)
            gm[0] = s_scale * r.x, gm[1] = s_scale * r.y, gm[2] = s_scale * r.z, gm[3] = s_scale * r.w;  // federated.py:110
        else
            gm[0] = r.x / (float)M, gm[1] = r.y / (float)M, gm[2] = r.z / (float)M, gm[3] = r.w / (float)M;  // reduce_mean (:62)
        const bool actor = i4 * 4 < actor_size;  // blocks are 4-float aligned
        for (int i = lead_skip ? 1 : 0; i < M; ++i) {
            const long o = (set0 + i) * n4 + i4;
            const float alpha = actor ? s_aa[i] : s_ac[i];
            float4 w = theta[o], wt = theta_t[o], mm = m[o], vv = v[o];
            float* wp = &w.x;
            float* tp = &wt.x;
            float* mp = &mm.x;
            float* vp = &vv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mp[k] = mp[k] + (gm[k] - mp[k]) * (1.0f - ADAM_B1);
                vp[k] = vp[k] + (gm[k] * gm[k] - vp[k]) * (1.0f - ADAM_B2);
                wp[k] = wp[k] - (mp[k] * alpha) / (sqrtf(vp[k]) + ADAM_EPS);
                tp[k] = wp[k] * tau + tp[k] * omt;
            }
            theta[o] = w, theta_t[o] = wt, m[o] = mm, v[o] = vv;
        }
    }
}
// the BN statistics' soft update for the same agents (set % M == 0 skipped under lead_skip)
__global__ void polyak_intra_kernel(int stats_size, int M, int lead_skip, const float* __restrict__ w, float* __restrict__ t, float tau,
                                    float omt) {
#pragma clang fp contract(off)
    const int set = blockIdx.y;
    if (lead_skip && set % M == 0) return;
    const long base = (long)set * stats_size;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < stats_size; i += gridDim.x * blockDim.x)
        t[base + i] = w[base + i] * tau + t[base + i] * omt;
}

__global__ void polyak_kernel(long n, const float* __restrict__ w, float* __restrict__ t, float tau, float omt) {
#pragma clang fp contract(off)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        t[i] = w[i] * tau + t[i] * omt;
}
// the BN statistics' soft update of the guarded form: set by set, skipping the sets adam_polyak_kernel<true> skips
__global__ void polyak_guarded_kernel(int stats_size, const float* __restrict__ w, float* __restrict__ t, float tau, float omt,
                                      const float* __restrict__ grads, int theta_size, int actor_size) {
#pragma clang fp contract(off)
    const int set = blockIdx.y;
    if (slab_is_nan(grads + (long)set * theta_size, actor_size)) return;
    const long base = (long)set * stats_size;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < stats_size; i += gridDim.x * blockDim.x)
        t[base + i] = w[base + i] * tau + t[base + i] * omt;
}

// out[o][j] = sum_i w[row(o,i)] * g[row(o,i)][j], reproducible: a workgroup owns 64 float4 columns of one output row and
// splits the members i over 4 phases (thread = column + 64 * phase, members phase, phase + 4, ...), 8 loads in flight per
// thread; the four partial sums meet in LDS in phase order. (One thread per column walking all members serially kept
// 1.5 workgroups per CU busy: 3.0 TB/s on the 6.3 GB gradient slab of 4096 x 5 agents; this form streams it at HBM rate.)
constexpr int FS_COLS = 64, FS_PH = 4, FS_UNR = 8;
__global__ __launch_bounds__(256) void fed_sum_kernel(int n_in, int so, int si, int n, const float4* __restrict__ g,
                                                      const float* __restrict__ weights, float4* __restrict__ out,
                                                      float* __restrict__ wsum) {
    __shared__ float4 part[FS_PH][FS_COLS];
    const int n4 = n / 4;
    const int o = blockIdx.y;
    if (wsum && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n_in; ++i) s += weights[(long)o * so + (long)i * si];
        wsum[o] = s;
    }
    const int col = threadIdx.x % FS_COLS, ph = threadIdx.x / FS_COLS;
    const int j = blockIdx.x * FS_COLS + col;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < n4) {
        for (int i0 = ph; i0 < n_in; i0 += FS_PH * FS_UNR) {
            float4 x[FS_UNR];
            float w[FS_UNR];
#pragma unroll
            for (int u = 0; u < FS_UNR; ++u) {
                const int i = i0 + FS_PH * u;
                const long row = (long)o * so + (long)(i < n_in ? i : ph) * si;  // past the end: a valid row, weight 0
                x[u] = g[row * n4 + j];
                w[u] = i < n_in ? (weights ? weights[row] : 1.0f) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < FS_UNR; ++u)
                if (i0 + FS_PH * u < n_in) acc.x += w[u] * x[u].x, acc.y += w[u] * x[u].y, acc.z += w[u] * x[u].z, acc.w += w[u] * x[u].w;
        }
    }
    part[ph][col] = acc;
    __syncthreads();
    if (ph == 0 && j < n4) {
        float4 r = part[0][col];
#pragma unroll
        for (int p = 1; p < FS_PH; ++p) r.x += part[p][col].x, r.y += part[p][col].y, r.z += part[p][col].z, r.w += part[p][col].w;
        out[(long)o * n4 + j] = r;
    }
}

__global__ void fed_finalize_kernel(int n_out, int n, float* __restrict__ out, float count,
                                    const float* __restrict__ wsum) {
#pragma clang fp contract(off)
    const long total = (long)n_out * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (wsum)
            out[i] = (1.0f / wsum[i / n]) * out[i];  // federated.py:110
        else
            out[i] = out[i] / count;  // reduce_mean (federated.py:62)
    }
}

// grid: (blocks over n/4, n_in - i_begin, n_out)
__global__ void fed_scatter_kernel(int so, int si, int i_begin, int n, const float4* __restrict__ src,
                                   float4* __restrict__ dst) {
    const int n4 = n / 4;
    const int o = blockIdx.z, i = blockIdx.y + i_begin;
    const long row = (long)o * so + (long)i * si;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n4; j += gridDim.x * blockDim.x)
        dst[row * n4 + j] = src[(long)o * n4 + j];
}

}  // namespace avd

using namespace avd;

static int adam_polyak_launch(const char* who, bool guard, const avd_mlp_layout* lay, int n_sets, float* theta, float* stats, float* theta_t,
                              float* stats_t, float* m, float* v, const float* grads, int32_t* step, float actor_lr, float critic_lr,
                              double tau, int32_t* skipped, void* stream) {
    AVD_REQUIRE(lay && n_sets > 0, "%s: n_sets=%d", who, n_sets);
    AVD_REQUIRE(theta && stats && theta_t && stats_t && m && v && grads && step, "%s: null pointer", who);
    AVD_REQUIRE(lay->theta_size % 4 == 0 && lay->actor_size % 4 == 0 && lay->stats_size % 4 == 0, "%s: layout not 4-float aligned", who);
    const float tauf = (float)tau, omt = (float)(1.0 - tau);  // Python doubles rounded to f32 (ddpgagent.py:47,53)
    const int n4 = lay->theta_size / 4;
    int gx = (n4 + 255) / 256;
    if (n_sets >= 256 && gx > 8) gx = 8;  // many sets: fewer, longer-lived blocks per set
    if (const char* e = AVD_DIAG_ENV("ADAM_GX")) gx = atoi(e);  // tuning knob (tools/adam_sweep.sh @ tag r06-pre-prune)
    if (guard)
        hipLaunchKernelGGL(adam_polyak_kernel<true>, dim3(gx, n_sets), dim3(256), 0, (hipStream_t)stream, lay->theta_size, lay->actor_size,
                           (float4*)theta, (float4*)theta_t, (float4*)m, (float4*)v, (const float4*)grads, step, actor_lr, critic_lr, tauf,
                           omt, skipped);
    else
        hipLaunchKernelGGL(adam_polyak_kernel<false>, dim3(gx, n_sets), dim3(256), 0, (hipStream_t)stream, lay->theta_size, lay->actor_size,
                           (float4*)theta, (float4*)theta_t, (float4*)m, (float4*)v, (const float4*)grads, step, actor_lr, critic_lr, tauf,
                           omt, (int32_t*)nullptr);
    int rc = check_launch(who);
    if (rc) return rc;
    // BN moving stats take part in the soft update too (ddpgagent.py:44-53 iterates .weights)
    if (guard) {
        hipLaunchKernelGGL(polyak_guarded_kernel, dim3((unsigned)((lay->stats_size + 255) / 256), n_sets), dim3(256), 0, (hipStream_t)stream,
                           lay->stats_size, stats, stats_t, tauf, omt, grads, lay->theta_size, lay->actor_size);
        return check_launch(who);
    }
    const long ns = (long)n_sets * lay->stats_size;
    long blocks = (ns + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(polyak_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ns, stats, stats_t,
                       tauf, omt);
    return check_launch(who);
}

// ---- federated weights on the device (workers/trainer.py:385-398; src/server/federated.py:99-118) ----------------------
// The reference weights agent (p, m) by |1 / mean(its last `weighted_window` episodic rewards)|. In the throughput modes no
// episodic reward ever reaches the host, so the history lives here: ring[P*M][W] of closed-episode rewards + hist_cnt[P].
// fed_history_push_kernel: one thread per platoon, once per step; the platoon's episode closes when
//   force != 0 (the caller's step limit), or *cond != 0 (the any-terminal flag: ALL platoons close, trainer.py:268-269), or
//   done[p] / ep_len[p] + 1 >= limit (per-platoon episodes: the condition avd_episode_end_f32 is about to apply).
__global__ void fed_history_push_kernel(int P, int M, int W, float* __restrict__ ep_reward, const uint8_t* __restrict__ done,
                                        const int32_t* __restrict__ ep_len, int limit, const int32_t* __restrict__ cond, int force,
                                        int zero_after, float* __restrict__ ring, int32_t* __restrict__ hist_cnt) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    bool end = force != 0 || (cond && *cond != 0);
    if (!end && done) end = done[p] != 0 || (ep_len && ep_len[p] + 1 >= limit);
    if (!end) return;
    const int c = hist_cnt[p], slot = c % W;
    for (int m = 0; m < M; ++m) {
        const long v = (long)p * M + m;
        ring[v * W + slot] = ep_reward[v];
        if (zero_after) ep_reward[v] = 0.f;
    }
    hist_cnt[p] = c + 1;
}

// fed_weights: two launches. (1) fed_weights_w_kernel, one thread per agent: the candidate weight |1 / mean(ring row)| (the W entries
// summed in slot order, float32). (2) fed_weights_kernel, one block of 1024 threads per vehicle index m: enabled = host_enabled (0 / 1),
// or, when host_enabled < 0, "every platoon has closed at least W episodes" (the reference's `training_episode >= weighted_window`,
// trainer.py:694: with the all-platoons episode rule every count is the episode number). Enabled: w = the candidate, wsum[m] = sum_p w,
// agent_weight = w P / wsum[m] (the factor the set learners take); disabled: w = agent_weight = 1, wsum[m] = P -- the weighted formulas
// then give the plain mean. (As ONE kernel of M blocks walking P x W ring entries per block it was latency-bound: 60 us per step.)
__global__ void fed_weights_w_kernel(long n, int W, const float* __restrict__ ring, float* __restrict__ w_cand) {
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    float acc = 0.f;
    for (int k = 0; k < W; ++k) acc += ring[v * W + k];
    w_cand[v] = fabsf(1.0f / (acc / (float)W));
}
__global__ __launch_bounds__(1024) void fed_weights_kernel(int P, int M, int W, const int32_t* __restrict__ hist_cnt, int host_enabled,
                                                           float* __restrict__ w_raw, float* __restrict__ aw, float* __restrict__ wsum) {
    __shared__ float part[1024];
    __shared__ int cmin[1024];
    const int m = blockIdx.x, tid = threadIdx.x;
    int enabled = host_enabled;
    if (host_enabled < 0) {
        int c = 0x7fffffff;
        for (int p = tid; p < P; p += 1024) c = min(c, hist_cnt[p]);
        cmin[tid] = c;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (tid < o) cmin[tid] = min(cmin[tid], cmin[tid + o]);
            __syncthreads();
        }
        enabled = cmin[0] >= W ? 1 : 0;
    }
    float s = 0.f;
    for (int p = tid; p < P; p += 1024) {
        const long v = (long)p * M + m;
        const float w = enabled ? w_raw[v] : 1.0f;  // (w_raw holds the candidates)
        w_raw[v] = w;
        s += w;
    }
    part[tid] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {  // fixed tree: the same bits on every launch
        if (tid < o) part[tid] += part[tid + o];
        __syncthreads();
    }
    const float total = part[0];
    if (tid == 0) wsum[m] = total;
    const float f = (float)P / total;
    for (int p = tid; p < P; p += 1024) {
        const long v = (long)p * M + m;
        aw[v] = enabled ? w_raw[v] * f : 1.0f;
    }
}

extern "C" int avd_adam_polyak_f32(const avd_mlp_layout* lay, int n_sets, float* theta, float* stats, float* theta_t,
                                   float* stats_t, float* m, float* v, const float* grads, const int32_t* step,
                                   float actor_lr, float critic_lr, double tau, void* stream) {
    return adam_polyak_launch("avd_adam_polyak_f32", false, lay, n_sets, theta, stats, theta_t, stats_t, m, v, grads, (int32_t*)step, actor_lr,
                              critic_lr, tau, nullptr, stream);
}

extern "C" int avd_adam_polyak_guarded_f32(const avd_mlp_layout* lay, int n_sets, float* theta, float* stats, float* theta_t,
                                           float* stats_t, float* m, float* v, const float* grads, int32_t* step, float actor_lr,
                                           float critic_lr, double tau, int32_t* skipped, void* stream) {
    return adam_polyak_launch("avd_adam_polyak_guarded_f32", true, lay, n_sets, theta, stats, theta_t, stats_t, m, v, grads, step, actor_lr,
                              critic_lr, tau, skipped, stream);
}

int launch_adam_polyak_ranges(const avd_mlp_layout* lay, int n_sets, const float* theta_in, float* theta_out,
                              float* theta_t, float* m, float* v, const float* grads, const int32_t* step,
                              float actor_lr, float critic_lr, double tau, int skip_a0, int skip_a1, int skip_c0,
                              int skip_c1, void* stream) {
    AVD_REQUIRE(skip_a0 % 4 == 0 && skip_a1 % 4 == 0 && skip_c0 % 4 == 0 && skip_c1 % 4 == 0 && skip_a1 <= skip_c0,
                "launch_adam_polyak_ranges: skip ranges must be 4-float aligned and ordered");
    const int kept4 = (lay->theta_size - (skip_a1 - skip_a0) - (skip_c1 - skip_c0)) / 4;
    int gx = (kept4 + 255) / 256;
    hipLaunchKernelGGL(adam_polyak_ranges_kernel, dim3(gx, n_sets), dim3(256), 0, (hipStream_t)stream, lay->theta_size,
                       lay->actor_size, (const float4*)theta_in, (float4*)theta_out, (float4*)theta_t, (float4*)m,
                       (float4*)v, (const float4*)grads, step, actor_lr, critic_lr, (float)tau, (float)(1.0 - tau),
                       skip_a0, skip_a1, skip_c0, skip_c1);
    return check_launch("avd_learn_update_f32(small tensors)");
}

int launch_adam_polyak_rows(const avd_mlp_layout* lay, int n_sets, int n_groups, const float* theta_in, float* theta_out, float* theta_t,
                            float* m, float* v, const float* grads, const int32_t* step, float actor_lr, float critic_lr, double tau,
                            void* stream) {
    AVD_REQUIRE(lay->theta_size % 4 == 0 && lay->actor_size % 4 == 0 && n_groups > 0, "launch_adam_polyak_rows: layout not 4-float aligned");
    hipLaunchKernelGGL(adam_polyak_rows_kernel, dim3(n_groups < n_sets ? n_groups : n_sets), dim3(256), 0, (hipStream_t)stream,
                       lay->theta_size, lay->actor_size, n_sets, (const float4*)theta_in, (float4*)theta_out, (float4*)theta_t, (float4*)m,
                       (float4*)v, (const float4*)grads, step, actor_lr, critic_lr, (float)tau, (float)(1.0 - tau));
    return check_launch("avd_learn_update_f32(update pass)");
}

extern "C" int avd_polyak_f32(int64_t n, const float* w, float* t, double tau, void* stream) {
    AVD_REQUIRE(n > 0 && w && t, "avd_polyak_f32: n=%ld", (long)n);
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(polyak_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long)n, w, t,
                       (float)tau, (float)(1.0 - tau));
    return check_launch("avd_polyak_f32");
}

extern "C" int avd_fed_sum_f32(int n_out, int n_in, int stride_out, int stride_in, int n, const float* g,
                               const float* weights, float* out, float* wsum, void* stream) {
    AVD_REQUIRE(n_out > 0 && n_in > 0 && n > 0 && n % 4 == 0 && g && out, "avd_fed_sum_f32: n_out=%d n_in=%d n=%d",
                n_out, n_in, n);
    AVD_REQUIRE(n_out <= 65535, "avd_fed_sum_f32: n_out=%d exceeds the grid limit", n_out);
    AVD_REQUIRE(!wsum || weights, "avd_fed_sum_f32: wsum requested without weights");
    const int gx = (n / 4 + FS_COLS - 1) / FS_COLS;
    hipLaunchKernelGGL(fed_sum_kernel, dim3(gx, n_out), dim3(256), 0, (hipStream_t)stream, n_in, stride_out,
                       stride_in, n, (const float4*)g, weights, (float4*)out, wsum);
    return check_launch("avd_fed_sum_f32");
}

extern "C" int avd_fed_finalize_f32(int n_out, int n, float* out, float count, const float* wsum, void* stream) {
    AVD_REQUIRE(n_out > 0 && n > 0 && out && (wsum || count > 0.f), "avd_fed_finalize_f32: n_out=%d n=%d", n_out, n);
    long blocks = ((long)n_out * n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fed_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_out, n, out,
                       count, wsum);
    return check_launch("avd_fed_finalize_f32");
}

extern "C" int avd_fed_scatter_f32(int n_out, int n_in, int stride_out, int stride_in, int i_begin, int n,
                                   const float* src, float* dst, void* stream) {
    AVD_REQUIRE(n_out > 0 && n_in > 0 && i_begin >= 0 && i_begin < n_in && n > 0 && n % 4 == 0 && src && dst,
                "avd_fed_scatter_f32: n_out=%d n_in=%d i_begin=%d n=%d", n_out, n_in, i_begin, n);
    AVD_REQUIRE(n_out <= 65535 && n_in <= 65535, "avd_fed_scatter_f32: group counts exceed the grid limit");
    int gx = (n / 4 + 255) / 256;
    if (gx > 16) gx = 16;
    hipLaunchKernelGGL(fed_scatter_kernel, dim3(gx, n_in - i_begin, n_out), dim3(256), 0, (hipStream_t)stream,
                       stride_out, stride_in, i_begin, n, (const float4*)src, (float4*)dst);
    return check_launch("avd_fed_scatter_f32");
}

extern "C" int avd_fed_history_push_f32(int P, int M, int W, float* ep_reward, const uint8_t* done, const int32_t* ep_len, int limit,
                                        const int32_t* cond, int force, int zero_after, float* ring, int32_t* hist_cnt, void* stream) {
    AVD_REQUIRE(P > 0 && M > 0 && W > 0 && ep_reward && ring && hist_cnt, "avd_fed_history_push_f32: P=%d M=%d W=%d", P, M, W);
    AVD_REQUIRE(force || cond || (done && (!ep_len || limit >= 1)), "avd_fed_history_push_f32: no closing condition given");
    hipLaunchKernelGGL(fed_history_push_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, M, W, ep_reward, done,
                       ep_len, limit, cond, force, zero_after, ring, hist_cnt);
    return check_launch("avd_fed_history_push_f32");
}

extern "C" int avd_fed_weights_f32(int P, int M, int W, const float* ring, const int32_t* hist_cnt, int host_enabled, float* w_raw,
                                   float* agent_weight, float* wsum, void* stream) {
    AVD_REQUIRE(P > 0 && M > 0 && M <= 65535 && W > 0 && ring && hist_cnt && w_raw && agent_weight && wsum,
                "avd_fed_weights_f32: P=%d M=%d W=%d", P, M, W);
    const long n = (long)P * M;
    hipLaunchKernelGGL(fed_weights_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, W, ring, w_raw);
    hipLaunchKernelGGL(fed_weights_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, P, M, W, hist_cnt, host_enabled, w_raw,
                       agent_weight, wsum);
    return check_launch("avd_fed_weights_f32");
}

extern "C" int avd_adam_polyak_intra_f32(const avd_mlp_layout* lay, int P, int M, int lead_skip, float* theta, float* stats, float* theta_t,
                                         float* stats_t, float* m, float* v, const float* grads, const int32_t* step, const float* weights,
                                         float actor_lr, float critic_lr, double tau, void* stream) {
    AVD_REQUIRE(lay && P > 0 && M > 0 && M <= INTRA_MAX_M && P <= 65535, "avd_adam_polyak_intra_f32: P=%d M=%d (M <= %d)", P, M, INTRA_MAX_M);
    AVD_REQUIRE(theta && stats && theta_t && stats_t && m && v && grads && step, "avd_adam_polyak_intra_f32: null pointer");
    AVD_REQUIRE(lay->theta_size % 4 == 0 && lay->actor_size % 4 == 0, "avd_adam_polyak_intra_f32: layout not 4-float aligned");
    const float tauf = (float)tau, omt = (float)(1.0 - tau);
    const int n4 = lay->theta_size / 4;
    int gx = (n4 + 255) / 256;
    if (P >= 64 && gx > 8) gx = 8;  // many platoons: fewer, longer-lived blocks per platoon (as adam_polyak_launch)
    hipLaunchKernelGGL(adam_polyak_intra_kernel, dim3(gx, P), dim3(256), 0, (hipStream_t)stream, lay->theta_size, lay->actor_size, M,
                       lead_skip ? 1 : 0, (float4*)theta, (float4*)theta_t, (float4*)m, (float4*)v, (const float4*)grads, step, weights,
                       actor_lr, critic_lr, tauf, omt);
    int rc = check_launch("avd_adam_polyak_intra_f32");
    if (rc) return rc;
    hipLaunchKernelGGL(polyak_intra_kernel, dim3((unsigned)((lay->stats_size + 255) / 256), (unsigned)(P * M)), dim3(256), 0,
                       (hipStream_t)stream, lay->stats_size, M, lead_skip ? 1 : 0, stats, stats_t, tauf, omt);
    return check_launch("avd_adam_polyak_intra_f32");
}
