// Platoon environment kernels for gfx950: step (K1-K3), reset (K4), OU noise (K5), policy epilogue.
//
// HBM-bound byte work (48 B per vehicle-step + 5 B per platoon): one thread per vehicle, one
// 16-byte load of x per lane, whole platoons per 256-thread block so the predecessor chain and
// the per-platoon any-terminal / mean-reward reductions stay inside LDS.  All arithmetic is
// written without FMA contraction in the reference's operation order so the float32 result is
// bit-identical to the float32 oracle (oracle/platoon.py:batched_step).
#include "common.h"

namespace avd {

constexpr int ENV_THREADS = 256;

struct EnvLds {
    float A[AVD_MAX_L][16];
    float B[AVD_MAX_L][4];
    float C[AVD_MAX_L][4];
    float chain[ENV_THREADS];  // Model B: this step's action; Model A: post-step acceleration
    float negr[ENV_THREADS];
    int term[ENV_THREADS];
};

__global__ __launch_bounds__(ENV_THREADS) void env_step_kernel(const avd_env_consts* __restrict__ cst, int P, int L,
                                                               const float4* __restrict__ x_in,
                                                               float4* __restrict__ x_out, float* __restrict__ prev_a,
                                                               float* __restrict__ cum_accel,
                                                               const float* __restrict__ u,
                                                               const float* __restrict__ leader_exog,
                                                               float* __restrict__ reward, uint8_t* __restrict__ term,
                                                               uint8_t* __restrict__ done,
                                                               float* __restrict__ reward_mean,
                                                               int32_t* __restrict__ any_done) {
#pragma clang fp contract(off)
    __shared__ EnvLds lds;
    const int tid = threadIdx.x;
    const int pb = ENV_THREADS / L;  // whole platoons per block
    const int p0 = blockIdx.x * pb;
    // stage the per-vehicle-index matrices once per block
    for (int i = tid; i < L * 16; i += ENV_THREADS) lds.A[i >> 4][i & 15] = cst->A[i >> 4][i & 15];
    for (int i = tid; i < L * 4; i += ENV_THREADS) {
        lds.B[i >> 2][i & 3] = cst->B[i >> 2][i & 3];
        lds.C[i >> 2][i & 3] = cst->C[i >> 2][i & 3];
    }
    const int lp = tid / L;
    const int i = tid - lp * L;
    const int p = p0 + lp;
    const bool active = (lp < pb) && (p < P);
    const long v = (long)p * L + i;
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    float pa = 0.f, uu = 0.f;
    if (active) {
        xv = x_in[v];
        pa = prev_a[v];
        uu = u[v];
    }
    __syncthreads();
    const float* Ai = lds.A[i];
    const float* Bi = lds.B[i];
    const float* Ci = lds.C[i];
    // A.dot(x) row by row, left to right (environment.py:513)
    float ax[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ax[r] = ((Ai[r * 4 + 0] * xv.x + Ai[r * 4 + 1] * xv.y) + Ai[r * 4 + 2] * xv.z) + Ai[r * 4 + 3] * xv.w;
    const int model_a = cst->model_a;
    // what the follower behind needs: Model B the action, Model A the post-step accel (C[2] == 0 by construction)
    lds.chain[tid] = model_a ? (ax[2] + Bi[2] * uu) : uu;
    __syncthreads();
    float exog = 0.f;
    if (active) exog = (i == 0) ? leader_exog[p] : lds.chain[tid - 1];
    // reward from the PRE-update state (environment.py:473-476, 505-510)
    const float norm_ep = fabsf(xv.x) / cst->max_ep;
    const float norm_ev = fabsf(xv.y) / cst->max_ev;
    const float norm_u = fabsf(uu) / cst->abs_action_high;
    const float n_jerk = fabsf(xv.z - pa) / cst->two_max_a;
    const bool is_term = ((fabsf(xv.x) > cst->max_ep) || (fabsf(xv.y) > cst->max_ev)) && (cst->can_terminate != 0);
    float rew = (((cst->ca * norm_ep + cst->cb * norm_ev) + cst->cc * norm_u) + cst->cd * n_jerk) * cst->re_scalar;
    if (is_term) rew = cst->terminal_reward * cst->re_scalar;
    const float negr = -rew;
    if (active) {
        float4 xn;
        xn.x = (ax[0] + Bi[0] * uu) + Ci[0] * exog;
        xn.y = (ax[1] + Bi[1] * uu) + Ci[1] * exog;
        xn.z = (ax[2] + Bi[2] * uu) + Ci[2] * exog;
        xn.w = (ax[3] + Bi[3] * uu) + Ci[3] * exog;
        x_out[v] = xn;      // state advances even when terminal (:512-513)
        prev_a[v] = xv.z;   // prev_x <- x
        if (cum_accel) cum_accel[v] = cum_accel[v] + xv.z;  // :500
        reward[v] = negr;
        if (term) term[v] = is_term ? 1 : 0;
    }
    lds.negr[tid] = negr;
    lds.term[tid] = (active && is_term) ? 1 : 0;
    __syncthreads();
    int block_any = 0;
    if (active && i == 0) {
        int any = 0;
        float s = 0.f;
        for (int k = 0; k < L; ++k) {
            any |= lds.term[tid + k];
            s = s + lds.negr[tid + k];
        }
        done[p] = (uint8_t)any;
        if (reward_mean) reward_mean[p] = (1.0f / (float)L) * s;  // environment.py:281
        block_any = any;
    }
    // any-terminal flag (trainer.py:268): one plain store per workgroup at most. Every writer stores the same value,
    // so no atomic is needed (an atomicOr per terminal platoon serialises on one address: 13x slower at P = 2^20).
    if (any_done && __syncthreads_or(block_any) && tid == 0) *any_done = 1;
}

// Fresh state of vehicle (p, i) (Vehicle.reset, environment.py:520-559; Platoon.reset :284-301) -- the draws only; the caller
// chains a_lead = predecessor's fresh x[2] (:291-294).
__device__ __forceinline__ void reset_draws(const avd_env_consts* cst, int mode, const float* draws, const float* front_accel,
                                            uint64_t seed, uint64_t counter, int p, int i, long v, float& d0, float& d1,
                                            float& d2, float& fa) {
#pragma clang fp contract(off)
    if (mode == 1) {  // evaluator constants (environment.py:534-539)
        d0 = cst->reset_ep_eval, d1 = cst->reset_ev_eval, d2 = cst->reset_a_eval;
    } else if (mode == 2) {  // rand_states=False (:552-555)
        d0 = cst->reset_ep_max, d1 = cst->reset_max_ev, d2 = cst->reset_max_a;
    } else if (draws) {  // host-RNG parity mode
        d0 = draws[v * 3 + 0], d1 = draws[v * 3 + 1], d2 = draws[v * 3 + 2];
    } else {  // device Philox (:547-549; util.py:67-70)
        const u32x4 ra = philox_at(seed, counter, (uint32_t)v, STREAM_RESET_A);
        const u32x4 rb = philox_at(seed, counter, (uint32_t)v, STREAM_RESET_B);
        if (cst->uniform_reset) {
            d0 = uniform_pm1(ra.x) * cst->reset_ep_max;
            d1 = uniform_pm1(ra.y) * cst->reset_max_ev;
            d2 = uniform_pm1(rb.x) * cst->reset_max_a;
        } else {
            float n1;
            const float n0 = box_muller(ra.x, ra.y, &n1);
            d0 = n0 * cst->reset_ep_max;
            d1 = n1 * cst->reset_max_ev;
            d2 = box_muller(rb.x, rb.y, nullptr) * cst->reset_max_a;
        }
    }
    if (i == 0) {
        if (front_accel) {
            fa = front_accel[p];
        } else {
            const u32x4 rb = philox_at(seed, counter, (uint32_t)v, STREAM_RESET_B);
            fa = (cst->uniform_reset ? uniform_pm1(rb.z) : box_muller(rb.z, rb.w, nullptr)) * cst->leader_reset_a;
        }
    }
}

__global__ __launch_bounds__(ENV_THREADS) void env_reset_kernel(const avd_env_consts* __restrict__ cst, int P, int L,
                                                                float4* __restrict__ x, float* __restrict__ prev_a,
                                                                float* __restrict__ cum_accel,
                                                                const float* __restrict__ draws,
                                                                const float* __restrict__ front_accel, int mode,
                                                                uint64_t seed, uint64_t counter,
                                                                const int32_t* __restrict__ cond) {
#pragma clang fp contract(off)
    __shared__ float x2s[ENV_THREADS];
    if (cond && *cond == 0) return;  // uniform across the grid
    const int tid = threadIdx.x;
    const int pb = ENV_THREADS / L;
    const int lp = tid / L;
    const int i = tid - lp * L;
    const int p = blockIdx.x * pb + lp;
    const bool active = (lp < pb) && (p < P);
    const long v = (long)p * L + i;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, fa = 0.f;
    if (active) reset_draws(cst, mode, draws, front_accel, seed, counter, p, i, v, d0, d1, d2, fa);
    x2s[tid] = d2;
    __syncthreads();
    if (active) {
        const float a_lead = (i == 0) ? fa : x2s[tid - 1];  // chain: predecessor's fresh x[2] (:291-294)
        x[v] = make_float4(d0, d1, d2, a_lead);
        prev_a[v] = d2;  // prev_x = x (:557)
        if (cum_accel) cum_accel[v] = 0.f;
    }
}

// Per-platoon episode end (vectorised-environment form of workers/trainer.py:232-273; device-RNG throughput mode). The reference
// ends the episode of ALL platoons when any one is terminal (:268-269) -- with thousands of platoons that cuts every episode to
// the first terminal among them. Here each platoon runs its own episode: after a step, a platoon whose step was terminal
// (done[p]) or whose episode has reached `limit` steps (config.py:89) closes its episode -- its M float32 episodic reward
// counters (:249, 321) go into the platoon's statistics (sum over finished episodes of the platoon-mean episodic reward, of the
// episode lengths, and the episode count: what trainer.py:510-517 appends per episode, kept as sums so that nothing leaves the
// device per step), the counters restart from 0 and the platoon gets fresh reset states (Platoon.reset, same draws as
// env_reset_kernel at (seed, counter, vehicle)). *any_reset is set to 1 when any platoon was reset (the caller's "states changed
// under the actor outputs" flag). One thread per vehicle, whole platoons per block.
__global__ __launch_bounds__(ENV_THREADS) void episode_end_kernel(const avd_env_consts* __restrict__ cst, int P, int L, int M,
                                                                  float4* __restrict__ x, float* __restrict__ prev_a,
                                                                  float* __restrict__ cum_accel,
                                                                  const uint8_t* __restrict__ done, int32_t* __restrict__ ep_len,
                                                                  float* __restrict__ ep_reward, int limit,
                                                                  float* __restrict__ ret_sum, float* __restrict__ len_sum,
                                                                  int32_t* __restrict__ ep_cnt, int32_t* __restrict__ any_reset,
                                                                  int mode, uint64_t seed, uint64_t counter) {
#pragma clang fp contract(off)
    __shared__ float x2s[ENV_THREADS];
    __shared__ float rs[ENV_THREADS];
    const int tid = threadIdx.x;
    const int pb = ENV_THREADS / L;
    const int lp = tid / L;
    const int i = tid - lp * L;
    const int p = blockIdx.x * pb + lp;
    const bool active = (lp < pb) && (p < P);
    const long v = (long)p * L + i;
    int len = 0;
    bool end = false;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, fa = 0.f, er = 0.f;
    if (active) {
        len = ep_len[p] + 1;
        end = (done[p] != 0) || (len >= limit);
        if (end) {
            reset_draws(cst, mode, nullptr, nullptr, seed, counter, p, i, v, d0, d1, d2, fa);
            if (i < M) er = ep_reward[(long)p * M + i];
        }
    }
    x2s[tid] = d2;
    rs[tid] = er;
    const int block_any = __syncthreads_or(end ? 1 : 0);  // also orders the ep_len reads above before the write below
    if (active && end) {
        const float a_lead = (i == 0) ? fa : x2s[tid - 1];
        x[v] = make_float4(d0, d1, d2, a_lead);
        prev_a[v] = d2;
        if (cum_accel) cum_accel[v] = 0.f;
        if (i < M) ep_reward[(long)p * M + i] = 0.f;
    }
    if (active && i == 0) {
        if (end) {
            float s = 0.f;
            for (int k = 0; k < M; ++k) s = s + rs[tid + k];  // vehicle order
            ret_sum[p] = ret_sum[p] + s / (float)M;
            len_sum[p] = len_sum[p] + (float)len;
            ep_cnt[p] = ep_cnt[p] + 1;
            ep_len[p] = 0;
        } else {
            ep_len[p] = len;
        }
    }
    if (any_reset && block_any && tid == 0) *any_reset = 1;  // every writer stores the same value
}

// ---- one launch per training step: OU noise -> policy clip -> leader exog -> platoon step -> replay add (+ reward sums) -----
// workers/trainer.py:282-322 for every platoon at once (device-RNG mode, decentralized agents): what ou_step_kernel,
// policy_kernel, normal_kernel / uniform_kernel, env_step_kernel, replay_add_kernel and the episodic-reward update do as
// seven launches, with the same Philox draws (stream, call counter, index) and the same unfused float arithmetic -- the
// results are bit-identical to the separate kernels (tests/test_gpu_trainer.py). Whole platoons sit inside ONE wavefront
// (64 / L platoons per wave, the remaining lanes idle), so the predecessor chain is a lane shift and the per-platoon
// any-terminal / reward reductions are wavefront operations (ballot, shuffles in vehicle order), not LDS loops.
struct StepArgs {
    const avd_env_consts* cst;
    int P, L, S;
    const float4* x_in;
    float4* x_out;
    float *prev_a, *cum_accel, *reward;
    uint8_t *term, *done;
    int32_t *any_done, *any_done_other;  // this step's flag (zero on entry); the other step parity's flag, zeroed here
    const float* actor_out;              // [P*L] tanh(.) * high
    float *ou_state, *action, *leader_exog;
    float theta, mean, dt, scale, lo, hi, exog_scale;
    int exog_uniform;
    uint64_t seed, ou_counter, exog_counter;
    float* ring;  // [P*L][cap][2S+2] or NULL (no replay add)
    int cap, slot;
    float* ep_reward;  // [P*L] += reward, or NULL
};

__global__ __launch_bounds__(ENV_THREADS) void step_fused_kernel(const StepArgs a) {
#pragma clang fp contract(off)
    __shared__ float sA[AVD_MAX_L][16], sB[AVD_MAX_L][4], sC[AVD_MAX_L][4];
    const avd_env_consts* cst = a.cst;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, L = a.L;
    for (int i = tid; i < L * 16; i += ENV_THREADS) sA[i >> 4][i & 15] = cst->A[i >> 4][i & 15];
    for (int i = tid; i < L * 4; i += ENV_THREADS) sB[i >> 2][i & 3] = cst->B[i >> 2][i & 3], sC[i >> 2][i & 3] = cst->C[i >> 2][i & 3];
    if (blockIdx.x == 0 && tid == 0 && a.any_done_other) *a.any_done_other = 0;
    const int pw = 64 / L;                      // whole platoons per wave
    const int lp = lane / L, i = lane - lp * L;  // platoon of the wave, vehicle
    const int p = (blockIdx.x * (ENV_THREADS / 64) + wv) * pw + lp;
    const bool active = (lp < pw) && (p < a.P);
    const long v = (long)p * L + i;
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    float pa = 0.f, uu = 0.f, exog_own = 0.f;
    if (active) {
        xv = a.x_in[v];
        pa = a.prev_a[v];
        // OUActionNoise.__call__ (src/noise.py:15-19) and policy (agent/ddpgagent.py:22-27)
        const u32x4 rn = philox_at(a.seed, a.ou_counter, (uint32_t)v, STREAM_OU);
        const float nrm = box_muller(rn.x, rn.y, nullptr);
        const float st = a.ou_state[v];
        const float noise = (st + (a.theta * (a.mean - st)) * a.dt) + a.scale * nrm;
        a.ou_state[v] = noise;
        uu = fminf(fmaxf(a.actor_out[v] + noise, a.lo), a.hi);
        a.action[v] = uu;
        if (i == 0) {  // leader exog, redrawn every step (workers/trainer.py:291-295; util.get_random_val)
            const u32x4 re = philox_at(a.seed, a.exog_counter, (uint32_t)p, STREAM_NORMAL);
            exog_own = (a.exog_uniform ? uniform_pm1(re.x) : box_muller(re.x, re.y, nullptr)) * a.exog_scale;
            a.leader_exog[p] = exog_own;
        }
    }
    __syncthreads();
    const float* Ai = sA[i];
    const float* Bi = sB[i];
    const float* Ci = sC[i];
    float ax[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ax[r] = ((Ai[r * 4 + 0] * xv.x + Ai[r * 4 + 1] * xv.y) + Ai[r * 4 + 2] * xv.z) + Ai[r * 4 + 3] * xv.w;
    // what the follower behind needs: Model B the action, Model A the post-step accel -- one lane up
    const float chain = cst->model_a ? (ax[2] + Bi[2] * uu) : uu;
    const float from_pred = __shfl_up(chain, 1);
    const float exog = (i == 0) ? exog_own : from_pred;
    const float norm_ep = fabsf(xv.x) / cst->max_ep;
    const float norm_ev = fabsf(xv.y) / cst->max_ev;
    const float norm_u = fabsf(uu) / cst->abs_action_high;
    const float n_jerk = fabsf(xv.z - pa) / cst->two_max_a;
    const bool is_term = active && ((fabsf(xv.x) > cst->max_ep) || (fabsf(xv.y) > cst->max_ev)) && (cst->can_terminate != 0);
    float rew = (((cst->ca * norm_ep + cst->cb * norm_ev) + cst->cc * norm_u) + cst->cd * n_jerk) * cst->re_scalar;
    if (is_term) rew = cst->terminal_reward * cst->re_scalar;
    const float negr = -rew;
    const unsigned long long tmask = __ballot(is_term);
    if (active) {
        float4 xn;
        xn.x = (ax[0] + Bi[0] * uu) + Ci[0] * exog;
        xn.y = (ax[1] + Bi[1] * uu) + Ci[1] * exog;
        xn.z = (ax[2] + Bi[2] * uu) + Ci[2] * exog;
        xn.w = (ax[3] + Bi[3] * uu) + Ci[3] * exog;
        a.x_out[v] = xn;
        a.prev_a[v] = xv.z;
        if (a.cum_accel) a.cum_accel[v] = a.cum_accel[v] + xv.z;
        a.reward[v] = negr;
        if (a.term) a.term[v] = is_term ? 1 : 0;
        if (a.ep_reward) a.ep_reward[v] = a.ep_reward[v] + negr;  // float32 counters (workers/trainer.py:249, 321)
        if (i == 0) a.done[p] = (uint8_t)(((tmask >> (lp * L)) & ((1ull << L) - 1ull)) != 0ull);
        if (a.ring) {  // ReplayBuffer.add (src/replaybuffer.py:36-47): row [s a r s'] at slot counter % capacity
            const int S = a.S, row = 2 * S + 2;
            float* dst = a.ring + ((long)v * a.cap + a.slot) * row;
            const float xo[4] = {xv.x, xv.y, xv.z, xv.w}, xw[4] = {xn.x, xn.y, xn.z, xn.w};
            if (S == 4) {  // 40-byte rows, 8-byte aligned
                ((float2*)dst)[0] = make_float2(xo[0], xo[1]);
                ((float2*)dst)[1] = make_float2(xo[2], xo[3]);
                ((float2*)dst)[2] = make_float2(uu, negr);
                ((float2*)dst)[3] = make_float2(xw[0], xw[1]);
                ((float2*)dst)[4] = make_float2(xw[2], xw[3]);
            } else {
                for (int k = 0; k < S; ++k) dst[k] = xo[k], dst[S + 2 + k] = xw[k];
                dst[S] = uu, dst[S + 1] = negr;
            }
        }
    }
    // any-terminal flag (trainer.py:268): one plain store per workgroup at most, every writer stores the same value
    if (a.any_done && __syncthreads_or(tmask != 0ull) && tid == 0) *a.any_done = 1;
}

__global__ void ou_step_kernel(int n, float* __restrict__ st, const float* __restrict__ normals, float theta,
                               float mean, float dt, float scale, uint64_t seed, uint64_t counter) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float nrm;
    if (normals) {
        nrm = normals[i];
    } else {
        const u32x4 r = philox_at(seed, counter, (uint32_t)i, STREAM_OU);
        nrm = box_muller(r.x, r.y, nullptr);
    }
    const float x = st[i];
    st[i] = (x + (theta * (mean - x)) * dt) + scale * nrm;  // noise.py:15-19
}

__global__ void policy_kernel(int n, const float* __restrict__ actor_out, const float* __restrict__ noise, float lo,
                              float hi, float* __restrict__ action) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = actor_out[i];
    if (noise) a = a + noise[i];
    action[i] = fminf(fmaxf(a, lo), hi);  // np.clip (ddpgagent.py:27)
}

__global__ void normal_kernel(int n, float* __restrict__ out, float std_dev, uint64_t seed, uint64_t counter) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32x4 r = philox_at(seed, counter, (uint32_t)i, STREAM_NORMAL);
    out[i] = box_muller(r.x, r.y, nullptr) * std_dev;
}

__global__ void uniform_kernel(int n, float* __restrict__ out, float half_width, uint64_t seed, uint64_t counter) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32x4 r = philox_at(seed, counter, (uint32_t)i, STREAM_NORMAL);  // same stream slot as normal_kernel: one or the other
    out[i] = uniform_pm1(r.x) * half_width;
}

}  // namespace avd

using namespace avd;

extern "C" int avd_env_step_f32(const avd_env_consts* d_consts, int P, int L, const float* x_in, float* x_out,
                                float* prev_a, float* cum_accel, const float* u, const float* leader_exog,
                                float* reward, uint8_t* term, uint8_t* done, float* reward_mean, int32_t* any_done,
                                void* stream) {
    AVD_REQUIRE(P > 0 && L > 0 && L <= AVD_MAX_L, "avd_env_step_f32: P=%d L=%d (L must be 1..%d)", P, L, AVD_MAX_L);
    AVD_REQUIRE(d_consts && x_in && x_out && prev_a && u && leader_exog && reward && done,
                "avd_env_step_f32: null pointer");
    const int pb = ENV_THREADS / L;
    const int grid = (P + pb - 1) / pb;
    hipLaunchKernelGGL(env_step_kernel, dim3(grid), dim3(ENV_THREADS), 0, (hipStream_t)stream, d_consts, P, L,
                       (const float4*)x_in, (float4*)x_out, prev_a, cum_accel, u, leader_exog, reward, term, done,
                       reward_mean, any_done);
    return check_launch("avd_env_step_f32");
}

extern "C" int avd_env_reset_f32(const avd_env_consts* d_consts, int P, int L, float* x, float* prev_a,
                                 float* cum_accel, const float* draws, const float* front_accel, int mode,
                                 uint64_t seed, uint64_t counter, const int32_t* cond, void* stream) {
    AVD_REQUIRE(P > 0 && L > 0 && L <= AVD_MAX_L, "avd_env_reset_f32: P=%d L=%d", P, L);
    AVD_REQUIRE(d_consts && x && prev_a, "avd_env_reset_f32: null pointer");
    AVD_REQUIRE(mode >= 0 && mode <= 2, "avd_env_reset_f32: mode %d", mode);
    const int pb = ENV_THREADS / L;
    const int grid = (P + pb - 1) / pb;
    hipLaunchKernelGGL(env_reset_kernel, dim3(grid), dim3(ENV_THREADS), 0, (hipStream_t)stream, d_consts, P, L,
                       (float4*)x, prev_a, cum_accel, draws, front_accel, mode, seed, counter, cond);
    return check_launch("avd_env_reset_f32");
}

extern "C" int avd_episode_end_f32(const avd_env_consts* d_consts, int P, int L, int M, float* x, float* prev_a, float* cum_accel,
                                   const uint8_t* done, int32_t* ep_len, float* ep_reward, int limit, float* ret_sum,
                                   float* len_sum, int32_t* ep_cnt, int32_t* any_reset, int mode, uint64_t seed,
                                   uint64_t counter, void* stream) {
    AVD_REQUIRE(P > 0 && L > 0 && L <= AVD_MAX_L && M >= 1 && M <= L, "avd_episode_end_f32: P=%d L=%d M=%d", P, L, M);
    AVD_REQUIRE(d_consts && x && prev_a && done && ep_len && ep_reward && ret_sum && len_sum && ep_cnt,
                "avd_episode_end_f32: null pointer");
    AVD_REQUIRE(limit >= 1 && mode >= 0 && mode <= 2, "avd_episode_end_f32: limit=%d mode=%d", limit, mode);
    const int pb = ENV_THREADS / L;
    hipLaunchKernelGGL(episode_end_kernel, dim3((P + pb - 1) / pb), dim3(ENV_THREADS), 0, (hipStream_t)stream, d_consts, P, L, M,
                       (float4*)x, prev_a, cum_accel, done, ep_len, ep_reward, limit, ret_sum, len_sum, ep_cnt, any_reset, mode,
                       seed, counter);
    return check_launch("avd_episode_end_f32");
}

extern "C" int avd_ou_step_f32(int n, float* ou_state, const float* normals, float theta, float mean, float dt,
                               float std_dev, uint64_t seed, uint64_t counter, void* stream) {
    AVD_REQUIRE(n > 0 && ou_state, "avd_ou_step_f32: n=%d", n);
    const float scale = std_dev * (float)sqrt((double)dt);  // float32(std_dev) * float32(sqrt(dt)), as the f32 oracle
    hipLaunchKernelGGL(ou_step_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, ou_state, normals,
                       theta, mean, dt, scale, seed, counter);
    return check_launch("avd_ou_step_f32");
}

extern "C" int avd_policy_f32(int n, const float* actor_out, const float* noise, float lo, float hi, float* action,
                              void* stream) {
    AVD_REQUIRE(n > 0 && actor_out && action, "avd_policy_f32: n=%d", n);
    hipLaunchKernelGGL(policy_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, actor_out, noise,
                       lo, hi, action);
    return check_launch("avd_policy_f32");
}

extern "C" int avd_normal_f32(int n, float* out, float std_dev, uint64_t seed, uint64_t counter, void* stream) {
    AVD_REQUIRE(n > 0 && out, "avd_normal_f32: n=%d", n);
    hipLaunchKernelGGL(normal_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, out, std_dev, seed,
                       counter);
    return check_launch("avd_normal_f32");
}

extern "C" int avd_uniform_f32(int n, float* out, float half_width, uint64_t seed, uint64_t counter, void* stream) {
    AVD_REQUIRE(n > 0 && out, "avd_uniform_f32: n=%d", n);
    hipLaunchKernelGGL(uniform_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, out, half_width, seed,
                       counter);
    return check_launch("avd_uniform_f32");
}

extern "C" int avd_step_fused_f32(const avd_env_consts* d_consts, int P, int L, int S, const float* x_in, float* x_out,
                                  float* prev_a, float* cum_accel, float* reward, uint8_t* term, uint8_t* done,
                                  int32_t* any_done, int32_t* any_done_other, const float* actor_out, float* ou_state,
                                  float* action, float* leader_exog, float ou_theta, float ou_mean, float ou_dt, float ou_std_dev,
                                  float action_low, float action_high, float exog_scale, int exog_uniform, uint64_t seed,
                                  uint64_t ou_counter, uint64_t exog_counter, float* ring, int cap, int64_t replay_counter,
                                  float* ep_reward, void* stream) {
    AVD_REQUIRE(P > 0 && L > 0 && L <= AVD_MAX_L && (S == 3 || S == 4), "avd_step_fused_f32: P=%d L=%d S=%d", P, L, S);
    AVD_REQUIRE(d_consts && x_in && x_out && prev_a && reward && done && actor_out && ou_state && action && leader_exog,
                "avd_step_fused_f32: null pointer");
    AVD_REQUIRE(!ring || (cap > 0 && replay_counter >= 0), "avd_step_fused_f32: cap=%d counter=%ld", cap, (long)replay_counter);
    StepArgs a;
    a.cst = d_consts, a.P = P, a.L = L, a.S = S, a.x_in = (const float4*)x_in, a.x_out = (float4*)x_out, a.prev_a = prev_a;
    a.cum_accel = cum_accel, a.reward = reward, a.term = term, a.done = done, a.any_done = any_done, a.any_done_other = any_done_other;
    a.actor_out = actor_out, a.ou_state = ou_state, a.action = action, a.leader_exog = leader_exog;
    a.theta = ou_theta, a.mean = ou_mean, a.dt = ou_dt, a.scale = ou_std_dev * (float)sqrt((double)ou_dt);  // as avd_ou_step_f32
    a.lo = action_low, a.hi = action_high, a.exog_scale = exog_scale, a.exog_uniform = exog_uniform;
    a.seed = seed, a.ou_counter = ou_counter, a.exog_counter = exog_counter;
    a.ring = ring, a.cap = cap, a.slot = ring ? (int)(replay_counter % cap) : 0, a.ep_reward = ep_reward;
    const int per_block = (ENV_THREADS / 64) * (64 / L);
    hipLaunchKernelGGL(step_fused_kernel, dim3((P + per_block - 1) / per_block), dim3(ENV_THREADS), 0, (hipStream_t)stream, a);
    return check_launch("avd_step_fused_f32");
}
