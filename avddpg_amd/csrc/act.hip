// Acting with SHARED weight sets on the matrix cores: mu(s) = tanh(actor(s)) * high (agent/model.py:26-36;
// workers/trainer.py:287-289) for all P platoons' vehicle-m agents of a set at once.
//
// With one weight set per vehicle index (interfrl, every step federated) the P agents of a set evaluate the same network on P
// different states: a [P x S] -> [P x 256] -> [P x 128] -> [P] chain per set. The batch-1 rows kernel (mlp.hip) streams the
// set's 143 KB of weights once per 8 agents (83 us at 4096 x 5); here a workgroup takes 32 agents of a set and its four
// waves one 32-column tile each of the second layer on v_mfma_f32_32x32x2_f32 -- exact f32 products, f32 accumulation, the
// reference's arithmetic class (no bf16 / fp16 anywhere) -- with the weights read straight from the f32 slab (L2-resident):
//   B operand: the first layer, recomputed per lane on the VALU (K = S <= 4 inputs: 4 FMAs + relu + BN per element, the
//              per-feature constants come from an LDS table built once per workgroup);
//   A operand: W2[f][n] as it lies in the slab, one dword per lane and k-step, 128 contiguous bytes per lane half;
//   epilogue:  bias, relu, BN2 and the 1-wide output layer per lane, the four waves' column sums meet in LDS.
// Not bit-identical to the rows kernel (different f32 summation order: 1e-7 relative), which stays the engine wherever
// bit-equality with the per-agent weight-set regime is asserted.
#include "common.h"

namespace avd {
namespace act {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float BN_EPS = 1e-3f;  // tf.keras BatchNormalization default epsilon (agent/model.py:28)
constexpr int H1 = 256, H2 = 128;

template <int S>
__global__ __launch_bounds__(256) void actor_set_kernel(const avd_mlp_layout L, int n_agents, int n_sets, const float* __restrict__ theta,
                                                        const float* __restrict__ stats, const float* __restrict__ states,
                                                        int x_stride, float high, float* __restrict__ out,
                                                        const int32_t* __restrict__ cond) {
    __shared__ __attribute__((aligned(16))) float t1[H1][8];  // per first-layer feature: w0..w3, b1, inv1, sh1, -
    __shared__ __attribute__((aligned(16))) float t2[H2][2];  // per second-layer column: b2, c3 = inv2 * w3
    __shared__ float red[4][32];
    __shared__ float d3s;
    if (cond && *cond == 0) return;  // uniform across the grid
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, kh = lane >> 5;
    const int set = blockIdx.x % n_sets, unit = blockIdx.x / n_sets, P = n_agents / n_sets;
    const float* th = theta + (long)set * L.theta_size;
    const float* st = stats + (long)set * L.stats_size;
    {
        const int f = tid;  // 256 threads = 256 features
        const float inv = (1.0f / sqrtf(st[L.amv1 + f] + BN_EPS)) * th[L.ag1 + f];
        t1[f][0] = th[L.aW1 + f], t1[f][1] = th[L.aW1 + H1 + f], t1[f][2] = th[L.aW1 + 2 * H1 + f];
        t1[f][3] = S > 3 ? th[L.aW1 + 3 * H1 + f] : 0.f;
        t1[f][4] = th[L.ab1 + f], t1[f][5] = inv, t1[f][6] = th[L.abe1 + f] - st[L.amm1 + f] * inv, t1[f][7] = 0.f;
        if (tid < H2) {
            const float inv2 = (1.0f / sqrtf(st[L.amv2 + tid] + BN_EPS)) * th[L.ag2 + tid];
            t2[tid][0] = th[L.ab2 + tid], t2[tid][1] = inv2 * th[L.aW3 + tid];
        }
    }
    __syncthreads();
    if (w == 0) {  // d3: 128 terms over 64 lanes, fixed order
        float v = 0.f;
        for (int n = lane; n < H2; n += 64) {
            const float inv2 = (1.0f / sqrtf(st[L.amv2 + n] + BN_EPS)) * th[L.ag2 + n];
            v += (th[L.abe2 + n] - st[L.amm2 + n] * inv2) * th[L.aW3 + n];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) d3s = th[L.ab3] + v;
    }
    const int prow = 32 * unit + r;  // platoon of this lane's batch row
    const bool ok = prow < P;
    const long agent = (long)(ok ? prow : 0) * n_sets + set;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < S; ++k) x[k] = states[agent * x_stride + k];
    const float* W2 = th + L.aW2 + 32 * w + r;  // + f * H2
    f32x16 acc = {};
#pragma unroll 8
    for (int step = 0; step < H1 / 2; ++step) {
        const int f = 2 * step + kh;
        const float4 c0 = *(const float4*)(&t1[f][0]), c1 = *(const float4*)(&t1[f][4]);
        float z = fmaf(x[0], c0.x, c1.x);
        z = fmaf(x[1], c0.y, z), z = fmaf(x[2], c0.z, z), z = fmaf(x[3], c0.w, z);
        const float p1 = fmaf(fmaxf(z, 0.f), c1.y, c1.z);  // BatchNorm (inference form) of relu(z1)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W2[(long)f * H2], p1, acc, 0, 0, 0);
    }
    // acc[i] = z2[column 32 w + (i & 3) + 8 (i >> 2) + 4 kh][row r] without its bias
    float zp = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = 32 * w + (i & 3) + 8 * (i >> 2) + 4 * kh;
        zp = fmaf(fmaxf(acc[i] + t2[n][0], 0.f), t2[n][1], zp);
    }
    zp += __shfl_xor(zp, 32);
    if (kh == 0) red[w][r] = zp;
    __syncthreads();
    if (w == 0 && kh == 0 && ok) out[agent] = tanhf(d3s + ((red[0][r] + red[1][r]) + (red[2][r] + red[3][r]))) * high;
}

}  // namespace act
}  // namespace avd

using namespace avd;

extern "C" int avd_actor_forward_set_f32(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                                         const float* states, int x_stride, float high, float* out, const int32_t* run_if_nonzero,
                                         void* stream) {
    AVD_REQUIRE(lay && theta && stats && states && out, "avd_actor_forward_set_f32: null pointer");
    if (lay->H1 != act::H1 || lay->H2 != act::H2 || lay->A != 1 || (lay->S != 3 && lay->S != 4)) {
        set_error("avd_actor_forward_set_f32: serves the reference widths only (layer1 256, layer2 128, A = 1, S in {3, 4}); got H1=%d "
                  "H2=%d A=%d S=%d (avd_actor_forward_f32 takes any)", lay->H1, lay->H2, lay->A, lay->S);
        return AVD_E_UNSUPPORTED;
    }
    AVD_REQUIRE(n_sets > 0 && n_agents > 0 && n_agents % n_sets == 0 && x_stride >= lay->S, "avd_actor_forward_set_f32: n_agents=%d n_sets=%d x_stride=%d",
                n_agents, n_sets, x_stride);
    const int P = n_agents / n_sets;
    const dim3 grid((unsigned)(((P + 31) / 32) * n_sets)), block(256);
    if (lay->S == 4)
        hipLaunchKernelGGL(act::actor_set_kernel<4>, grid, block, 0, (hipStream_t)stream, *lay, n_agents, n_sets, theta, stats, states,
                           x_stride, high, out, run_if_nonzero);
    else
        hipLaunchKernelGGL(act::actor_set_kernel<3>, grid, block, 0, (hipStream_t)stream, *lay, n_agents, n_sets, theta, stats, states,
                           x_stride, high, out, run_if_nonzero);
    return check_launch("avd_actor_forward_set_f32");
}
