// Replay buffer kernels for gfx950: ring write (K7), index draw and row gather (K8).
//
// ring[n_agents][cap][row] float32 with row = [s(S) a(A) r(1) s2(S)].  Integer work (slot, sample
// range, indices) is bit-exact with the reference / oracle; rows are copied, never recomputed.
#include "common.h"

namespace avd {

__global__ void replay_add_kernel(int n_agents, int cap, int S, int A, float* __restrict__ ring, int slot,
                                  const float* __restrict__ s_prev, const float* __restrict__ s_next, int x_stride,
                                  const float* __restrict__ action, const float* __restrict__ reward) {
    const int row = 2 * S + A + 1;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)n_agents * row) return;
    const int a = (int)(t / row);
    const int j = (int)(t - (long)a * row);
    float val;
    if (j < S)
        val = s_prev[(long)a * x_stride + j];
    else if (j < S + A)
        val = action[(long)a * A + (j - S)];
    else if (j == S + A)
        val = reward[a];
    else
        val = s_next[(long)a * x_stride + (j - S - A - 1)];
    ring[((long)a * cap + slot) * row + j] = val;
}

__global__ void replay_indices_kernel(int n_agents, int B, uint32_t range, uint64_t seed, uint64_t counter,
                                      int32_t* __restrict__ idx) {
    // one Philox call yields 4 indices
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n_agents * B;
    const long base = t * 4;
    if (base >= total) return;
    const u32x4 r = philox_at(seed, counter, (uint32_t)t, STREAM_REPLAY);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < total) idx[base + k] = (int32_t)(((uint64_t)w[k] * range) >> 32);
}

__global__ void replay_gather_kernel(int n_agents, int cap, int S, int A, int B, const float* __restrict__ ring,
                                     const int32_t* __restrict__ idx, float* __restrict__ s, float* __restrict__ a,
                                     float* __restrict__ r, float* __restrict__ s2) {
    const int row = 2 * S + A + 1;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n_agents * B * row;
    if (t >= total) return;
    const long ab = t / row;  // (agent, b)
    const int j = (int)(t - ab * row);
    const int ag = (int)(ab / B);
    const int ix = idx[ab];
    const float val = ring[((long)ag * cap + ix) * row + j];
    if (j < S)
        s[ab * S + j] = val;
    else if (j < S + A)
        a[ab * A + (j - S)] = val;
    else if (j == S + A)
        r[ab] = val;
    else
        s2[ab * S + (j - S - A - 1)] = val;
}

// ReplayBuffer.sample (src/replaybuffer.py:49-63) in one launch: a thread draws FOUR batch indices of an agent with one Philox
// call -- the draws of replay_indices_kernel, bit for bit -- and moves their rows whole (8-byte pieces of the 40-byte rows,
// 16-byte pieces out), instead of one thread per float behind a separate index kernel.
template <int S>
__global__ __launch_bounds__(256) void replay_sample_kernel(int n_agents, int cap, int B, const float* __restrict__ ring,
                                                            uint32_t range, uint64_t seed, uint64_t counter,
                                                            int32_t* __restrict__ idx, float* __restrict__ s,
                                                            float* __restrict__ a, float* __restrict__ r,
                                                            float* __restrict__ s2) {
    constexpr int row = 2 * S + 2;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n_agents * B, base = t * 4;
    if (base >= total) return;
    const u32x4 rr = philox_at(seed, counter, (uint32_t)t, STREAM_REPLAY);
    const uint32_t w[4] = {rr.x, rr.y, rr.z, rr.w};
    const int ag = (int)(base / B);  // B is a multiple of 4: the four rows belong to one agent
    int ix[4];
    float v[4][row];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ix[k] = (int32_t)(((uint64_t)w[k] * range) >> 32);
        const float2* src = (const float2*)(ring + ((long)ag * cap + ix[k]) * row);
#pragma unroll
        for (int j = 0; j < row / 2; ++j) {
            const float2 q = src[j];
            v[k][2 * j] = q.x, v[k][2 * j + 1] = q.y;
        }
    }
    *(int4*)(idx + base) = make_int4(ix[0], ix[1], ix[2], ix[3]);
    *(float4*)(a + base) = make_float4(v[0][S], v[1][S], v[2][S], v[3][S]);
    *(float4*)(r + base) = make_float4(v[0][S + 1], v[1][S + 1], v[2][S + 1], v[3][S + 1]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (S == 4) {
            *(float4*)(s + (base + k) * 4) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
            *(float4*)(s2 + (base + k) * 4) = make_float4(v[k][6], v[k][7], v[k][8], v[k][9]);
        } else {
#pragma unroll
            for (int j = 0; j < S; ++j) s[(base + k) * S + j] = v[k][j], s2[(base + k) * S + j] = v[k][S + 2 + j];
        }
    }
}

}  // namespace avd

using namespace avd;

extern "C" int avd_replay_sample_f32(int n_agents, int cap, int S, int A, int B, const float* ring, int range, uint64_t seed,
                                     uint64_t counter, int32_t* idx, float* s, float* a, float* r, float* s2, void* stream) {
    AVD_REQUIRE(n_agents > 0 && cap > 0 && (S == 3 || S == 4) && A == 1 && B > 0 && B % 4 == 0 && range > 0 && range <= cap,
                "avd_replay_sample_f32: n=%d cap=%d S=%d A=%d B=%d range=%d (S in {3, 4}, A = 1, B a multiple of 4)", n_agents, cap, S,
                A, B, range);
    AVD_REQUIRE(ring && idx && s && a && r && s2, "avd_replay_sample_f32: null pointer");
    const long calls = (long)n_agents * B / 4;
    const dim3 grid((unsigned)((calls + 255) / 256));
    if (S == 4)
        hipLaunchKernelGGL(replay_sample_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, n_agents, cap, B, ring, (uint32_t)range,
                           seed, counter, idx, s, a, r, s2);
    else
        hipLaunchKernelGGL(replay_sample_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, n_agents, cap, B, ring, (uint32_t)range,
                           seed, counter, idx, s, a, r, s2);
    return check_launch("avd_replay_sample_f32");
}

extern "C" int avd_replay_add_f32(int n_agents, int cap, int S, int A, float* ring, int64_t counter,
                                  const float* s_prev, const float* s_next, int x_stride, const float* action,
                                  const float* reward, void* stream) {
    AVD_REQUIRE(n_agents > 0 && cap > 0 && S > 0 && A > 0 && counter >= 0 && x_stride >= S,
                "avd_replay_add_f32: n=%d cap=%d S=%d A=%d counter=%ld x_stride=%d", n_agents, cap, S, A, (long)counter,
                x_stride);
    AVD_REQUIRE(ring && s_prev && s_next && action && reward, "avd_replay_add_f32: null pointer");
    const int slot = (int)(counter % cap);  // replaybuffer.py:40
    const long total = (long)n_agents * (2 * S + A + 1);
    hipLaunchKernelGGL(replay_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       n_agents, cap, S, A, ring, slot, s_prev, s_next, x_stride, action, reward);
    return check_launch("avd_replay_add_f32");
}

extern "C" int avd_replay_indices(int n_agents, int B, int range, uint64_t seed, uint64_t counter, int32_t* idx,
                                  void* stream) {
    AVD_REQUIRE(n_agents > 0 && B > 0 && range > 0 && idx, "avd_replay_indices: n=%d B=%d range=%d", n_agents, B,
                range);
    const long calls = ((long)n_agents * B + 3) / 4;
    hipLaunchKernelGGL(replay_indices_kernel, dim3((unsigned)((calls + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, n_agents, B, (uint32_t)range, seed, counter, idx);
    return check_launch("avd_replay_indices");
}

extern "C" int avd_replay_gather_f32(int n_agents, int cap, int S, int A, int B, const float* ring,
                                     const int32_t* idx, float* s, float* a, float* r, float* s2, void* stream) {
    AVD_REQUIRE(n_agents > 0 && cap > 0 && S > 0 && A > 0 && B > 0, "avd_replay_gather_f32: bad sizes");
    AVD_REQUIRE(ring && idx && s && a && r && s2, "avd_replay_gather_f32: null pointer");
    const long total = (long)n_agents * B * (2 * S + A + 1);
    hipLaunchKernelGGL(replay_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, n_agents, cap, S, A, B, ring, idx, s, a, r, s2);
    return check_launch("avd_replay_gather_f32");
}
