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

}  // namespace avd

using namespace avd;

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
