/*
 * avddpg_hip.h -- C ABI of the MI355X (gfx950) hot-path library libavddpg_hip.so.
 *
 * The reference (cboin1996/avddpg) has no FFI: its hot path sits behind plain
 * Python objects.  Each entry point below is the batched, device-resident
 * replacement of one of those objects' methods; the reference interface it
 * replaces is cited per function (file:line under the reference tree).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc'ed / torch CUDA tensor
 *    data_ptr) unless it is named h_* or is an avd_* struct passed by pointer
 *    from the host (those are read on the host at call time);
 *  - sizes are explicit; nothing is allocated, nothing synchronises: the call
 *    enqueues kernels on `stream` (a hipStream_t passed as void*) and returns;
 *  - return value: 0 = ok, < 0 = error (AVD_E_*), message in avd_last_error();
 *  - layouts: platoon-major; vehicle state x[P][L][4] float32 (one 16-byte
 *    load per vehicle); per-agent vectors [P][L]; agent id v = p*L + m.
 *  - thread-compatible: calls on different streams may run concurrently.
 */
#ifndef AVDDPG_HIP_H
#define AVDDPG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
/* The library is built with -fvisibility=hidden: these declarations are its whole dynamic symbol table. */
#pragma GCC visibility push(default)
#endif

#define AVD_MAX_L 16 /* vehicles per platoon supported by the env kernels */

#define AVD_OK 0
#define AVD_E_INVALID (-1)     /* bad argument / unsupported dimension */
#define AVD_E_LAUNCH (-2)      /* HIP launch or runtime error */
#define AVD_E_UNSUPPORTED (-3) /* shape outside what the kernels implement */

/* ---- environment constants (reference src/config.py:39-107 and the matrices of
 *      src/environment.py:390-451, built on the host in float64 and rounded once) ---- */
typedef struct avd_env_consts {
    int32_t L;             /* pl_size */
    int32_t model_a;       /* 1: Model A exogenous chain (predecessor's post-step accel, environment.py:263-267);
                              0: Model B (predecessor's action this step, :257-261) */
    int32_t can_terminate; /* config.py:75 */
    int32_t uniform_reset; /* rand_gen == uniform (util.py:67-68), device-RNG reset only */
    float max_ep, max_ev;  /* config.py:57-58 */
    float abs_action_high; /* |action_high|, environment.py:475 */
    float two_max_a;       /* 2*action_high, environment.py:476 */
    float T;               /* sample_rate */
    float ca, cb, cc, cd;  /* reward coefficients a,b,c,d config.py:52-55 */
    float re_scalar, terminal_reward;
    float stand_still, timegap;                              /* environment.py:343, 502 */
    float reset_ep_max, reset_max_ev, reset_max_a;           /* config.py:60-62 */
    float reset_ep_eval, reset_ev_eval, reset_a_eval;        /* config.py:64-66 */
    float leader_reset_a;                                    /* config.py:41 */
    float A[AVD_MAX_L][16]; /* row-major 4x4 per vehicle index */
    float B[AVD_MAX_L][4];
    float C[AVD_MAX_L][4];
} avd_env_consts;

/* ---- actor/critic parameter slab layout (agent/model.py:4-85) ----
 * One weight set = theta[theta_size] (trainable: actor block then critic block,
 * every tensor starting on a 4-float boundary, padding kept at zero) plus
 * stats[stats_size] (the non-trainable BatchNormalization moving mean/var).
 * Dense kernels are stored like Keras: W[in][out], row-major. */
typedef struct avd_mlp_layout {
    int32_t S, A, H1, H2, Ha, B;
    /* actor trainables */
    int32_t aW1, ab1, ag1, abe1, aW2, ab2, ag2, abe2, aW3, ab3;
    int32_t actor_size; /* critic block starts here */
    /* critic trainables */
    int32_t cWs, cbs, cgs, cbes, cWa, cba, cga, cbea, cW2, cb2, cg3, cbe3, cW3, cb3;
    int32_t theta_size;
    /* non-trainable stats */
    int32_t amm1, amv1, amm2, amv2, cmms, cmvs, cmma, cmva, cmm3, cmv3;
    int32_t stats_size;
} avd_mlp_layout;

const char* avd_last_error(void);
int avd_version(void);
/* 0 for the shipped library: it reads NO environment variable. 1 for the diagnostic build (`make diag`,
 * libavddpg_hip_diag.so, -DAVD_DIAG), in which AVD_* switches select kernel variants for A/B runs and cross-checks
 * (csrc/common.h AVD_DIAG_ENV). bench.py refuses a diagnostic library unless --allow-diagnostics. */
int avd_diagnostics_enabled(void);

/* Fills `out` for the given network widths. S = num_states (<= 64), A = num_actions (<= 16; the centralized
 * framework has S = 4L, A = L), H1/H2 = layer1/layer2 size, Ha = critic action layer size (config.py:112-117),
 * B = batch_size (config.py:106). The MLP kernels need H1 and Ha to be multiples of 16 and H2 a multiple of 32:
 * other widths (e.g. the centralized 307/153/57) are zero-padded by the caller -- a padded unit has zero
 * weights/bias and BN beta = mean = 0, outputs 0, receives zero gradients and stays zero under Adam. */
int avd_mlp_layout_init(avd_mlp_layout* out, int S, int A, int H1, int H2, int Ha, int B);

/* ---- environment ------------------------------------------------------------
 * Replaces Platoon.step (src/environment.py:209-241) + Vehicle.step (:460-518)
 * + get_exogenous_info (:253-269) + get_reward (:271-282) for P platoons at once.
 *   x_in  [P][L][4]  state before the step      x_out [P][L][4] state after (may alias x_in)
 *   prev_a[P][L]     prev_x[2], in/out           cum_accel [P][L] in/out or NULL (velocity/headway aux)
 *   u     [P][L]     actions                     leader_exog [P]
 *   reward[P][L]     NEGATED reward (as returned by Vehicle.step)   term [P][L] (0/1) or NULL
 *   done  [P]        any(term) per platoon       reward_mean [P] centralized mean or NULL
 *   any_done         single int32 OR-accumulated over all platoons (trainer.py:268) or NULL;
 *                    the caller zeroes it. */
int avd_env_step_f32(const avd_env_consts* d_consts, int P, int L, const float* x_in, float* x_out, float* prev_a,
                     float* cum_accel, const float* u, const float* leader_exog, float* reward, uint8_t* term,
                     uint8_t* done, float* reward_mean, int32_t* any_done, void* stream);

/* Replaces Platoon.reset / Vehicle.reset (src/environment.py:284-301, 520-559).
 *   mode 0 = training (random states), 1 = evaluator constants, 2 = rand_states=False constants.
 *   draws [P][L][3] pre-drawn, already scaled values (host-RNG parity mode) or NULL = device Philox
 *   front_accel [P] pre-drawn leader accel or NULL (device RNG: N(0, leader_reset_a)).
 *   cond        device int32 or NULL: when given, the reset happens only if *cond != 0 (the
 *               any-terminal episode break of workers/trainer.py:268-269 without a host sync). */
int avd_env_reset_f32(const avd_env_consts* d_consts, int P, int L, float* x, float* prev_a, float* cum_accel,
                      const float* draws, const float* front_accel, int mode, uint64_t seed, uint64_t counter,
                      const int32_t* cond, void* stream);

/* Per-platoon episode end -- the vectorised-environment form of the episode loop (workers/trainer.py:232-273) for the
 * device-RNG throughput mode. The reference closes the episode of ALL platoons when any is terminal (:268-269; that form is
 * avd_env_reset_f32 with `cond`). Here, called once after every step, each platoon closes its OWN episode when its step was
 * terminal (done[p], src/environment.py:237-241) or its episode has reached `limit` steps (src/config.py:89):
 *   ep_len    [P] int32  steps of the running episode (+1 per call; 0 after a close)
 *   ep_reward [P*M] f32  the episodic reward counters of :249, 321 (M = L decentralized, 1 centralized); zeroed on close
 *   ret_sum / len_sum [P] f32, ep_cnt [P] int32: per platoon, over its closed episodes, the sum of the platoon-mean episodic
 *             reward (what :510-517 appends per episode), of the episode lengths, and the episode count -- running sums the
 *             caller reads and clears whenever it reports a curve point; nothing leaves the device per step
 *   any_reset int32 or NULL: set to 1 when any platoon was closed (states changed under already-computed actor outputs)
 * A closed platoon gets fresh states exactly like avd_env_reset_f32 (mode, seed, counter; device Philox draws). */
int avd_episode_end_f32(const avd_env_consts* d_consts, int P, int L, int M, float* x, float* prev_a, float* cum_accel,
                        const uint8_t* done, int32_t* ep_len, float* ep_reward, int limit, float* ret_sum, float* len_sum,
                        int32_t* ep_cnt, int32_t* any_reset, int mode, uint64_t seed, uint64_t counter, void* stream);

/* Aux read-outs used by the evaluator/renderer (environment.py:243-251, 477, 500-503), from the
 * PRE-step state that produced them: jerk = (x2 - prev_a)/T is computed by the caller from the
 * buffers; no kernel needed. */

/* ---- exploration noise + policy ----------------------------------------------
 * OUActionNoise.__call__ (src/noise.py:14-23) for n independent scalar processes.
 *   normals [n] pre-drawn N(0,1) or NULL = device Philox (seed, counter). */
int avd_ou_step_f32(int n, float* ou_state, const float* normals, float theta, float mean, float dt, float std_dev,
                    uint64_t seed, uint64_t counter, void* stream);

/* ddpgagent.policy (agent/ddpgagent.py:6-29): action = clip(actor_out + noise, lo, hi).
 *   noise [n] or NULL (evaluator: no noise). */
int avd_policy_f32(int n, const float* actor_out, const float* noise, float lo, float hi, float* action,
                   void* stream);

/* Leader exogenous input N(0, std) per platoon per step (workers/trainer.py:291-295), device Philox. */
int avd_normal_f32(int n, float* out, float std_dev, uint64_t seed, uint64_t counter, void* stream);

/* The same draw when conf.rand_gen == 'uniform': U(-half_width, half_width) (src/util.py:55-70 get_random_val,
 * workers/trainer.py:291-295), device Philox, same (seed, counter, index) addressing as avd_normal_f32. */
int avd_uniform_f32(int n, float* out, float half_width, uint64_t seed, uint64_t counter, void* stream);

/* ---- replay buffer (src/replaybuffer.py:5-63) ----------------------------------
 * ring [n_agents][cap][row] float32, row = [s(S) a(A) r(1) s2(S)], row = 2S+A+1.
 * add: writes slot (counter % cap) of every agent (:40-47).
 *   s_prev/s_next: [n_agents] rows of `x_stride` floats, first S used (Model A hides x[3], environment.py:518). */
int avd_replay_add_f32(int n_agents, int cap, int S, int A, float* ring, int64_t counter, const float* s_prev,
                       const float* s_next, int x_stride, const float* action, const float* reward, void* stream);

/* Uniform indices with replacement in [0, range) (np.random.choice(range, B), :52-54) from device
 * Philox: idx[a][b] = mulhi32(philox(seed, counter, a, b), range). Host-RNG parity mode uploads
 * numpy's own indices instead. */
int avd_replay_indices(int n_agents, int B, int range, uint64_t seed, uint64_t counter, int32_t* idx, void* stream);

/* gather (:57-61): s[n][B][S], a[n][B][A], r[n][B], s2[n][B][S] <- ring rows idx[n][B]. */
int avd_replay_gather_f32(int n_agents, int cap, int S, int A, int B, const float* ring, const int32_t* idx, float* s,
                          float* a, float* r, float* s2, void* stream);

/* ---- actor / critic -------------------------------------------------------------
 * Weight-set addressing shared by the calls below: agent v uses set (set_mod > 0 ? v % set_mod : v)
 * -- set_mod = 0: one weight set per agent (reference `nofrl`, workers/trainer.py:100-149);
 *    set_mod = M: one set per vehicle index shared by all platoons (interfrl+gradients keeps
 *    them bit-identical, workers/trainer.py:415-425). */

/* actor(state) (agent/model.py:26-36, called at workers/trainer.py:287-289): out[v] = tanh(.)*high
 * for n_agents rows of `x_stride` floats (first S used); out [n_agents][A]. */
int avd_actor_forward_f32(const avd_mlp_layout* lay, int n_agents, int set_mod, const float* theta, const float* stats,
                          const float* state, int x_stride, float high, float* out, void* stream);

/* The same, executed only when *run_if_nonzero != 0 (a device flag, e.g. the any-terminal flag of the previous step):
 * when a fused update has already left actor(next state) in `out` (avd_learn_update_act_f32) and no reset has changed
 * the states since, the launch returns at once and `out` stays. */
int avd_actor_forward_cond_f32(const avd_mlp_layout* lay, int n_agents, int set_mod, const float* theta,
                               const float* stats, const float* state, int x_stride, float high, float* out,
                               const int32_t* run_if_nonzero, void* stream);

/* critic([state, action]) -> q[n_agents][A] (agent/model.py:63-83; the output width is num_actions, :80);
 * action [n_agents][A]; rows as above, batch 1 per agent. */
int avd_critic_forward_f32(const avd_mlp_layout* lay, int n_agents, int set_mod, const float* theta,
                           const float* stats, const float* state, int x_stride, const float* action, float* q,
                           void* stream);

/* Trainer.learn (workers/trainer.py:472-508) for n_agents batches of B rows:
 *   y = r + gamma*Q'(s2, mu'(s2)); Lc = mean((y-Q(s,a))^2); La = -mean(Q(s, mu(s)))
 *   grads [n_agents][theta_size]: actor block = dLa/dactor, critic block = dLc/dcritic (pre-update weights)
 *   losses [n_agents][2] = (Lc, La) or NULL. */
int avd_learn_f32(const avd_mlp_layout* lay, int n_agents, int set_mod, const float* theta, const float* stats,
                  const float* theta_t, const float* stats_t, const float* s, const float* a, const float* r,
                  const float* s2, float gamma, float high, float* grads, float* losses, void* stream);

/* Adam x2 (critic then actor; tf.keras.optimizers.Adam defaults, workers/trainer.py:138-139, 348-349)
 * followed by ddpgagent.update_target over ALL weights incl. BN stats (agent/ddpgagent.py:31-55;
 * workers/trainer.py:352-356), fused per element, for n_sets weight sets.
 *   grads [n_sets][theta_size]; step [n_sets] = Adam iteration count AFTER this update (>= 1). */
int avd_adam_polyak_f32(const avd_mlp_layout* lay, int n_sets, float* theta, float* stats, float* theta_t,
                        float* stats_t, float* m, float* v, const float* grads, const int32_t* step, float actor_lr,
                        float critic_lr, double tau, void* stream);

/* avd_adam_polyak_f32 behind a finiteness guard, for gradient slabs that come from the 16-bit set learners
 * (avd_learn_set_split_f16x3 / avd_learn_set_fused_bf16 turn a non-finite input, or a value fp16 cannot hold, into an ALL-NaN block of
 * the slab -- loud, but one such batch fed to Adam would poison the shared weight set for good, and through the all-reduce every rank's):
 * a set whose slab starts with a NaN in its actor block or in its critic block takes NO step -- weights, moments, targets and target
 * statistics untouched, step[set] (advanced by the caller, as for avd_adam_polyak_f32) put back by one, *skipped (optional, device
 * int32) incremented. Every other set: bit-identical to avd_adam_polyak_f32. (The reference's float32 Trainer.learn has no such range
 * limit, workers/trainer.py:472-508; states are bounded by max_ep / max_ev = 20, src/config.py:56-57, far below the limits.) */
int avd_adam_polyak_guarded_f32(const avd_mlp_layout* lay, int n_sets, float* theta, float* stats, float* theta_t, float* stats_t,
                                float* m, float* v, const float* grads, int32_t* step, float actor_lr, float critic_lr, double tau,
                                int32_t* skipped, void* stream);

/* Fused form of avd_learn_f32 + avd_adam_polyak_f32 for one weight set per agent (reference nofrl:
 * workers/trainer.py:325-356 learn, apply_gradients x2, update_target per agent): Adam and the soft update are
 * applied where each gradient is produced, so no gradient slab is written or read back.
 *   theta      [n_agents][theta_size] pre-update weights, READ ONLY for the whole call (both gradients are taken
 *              at the pre-update actor and critic, trainer.py:492-506)
 *   theta_out  [n_agents][theta_size] receives the updated weights; must not alias theta (callers ping-pong)
 *   theta_t, stats_t, m, v  updated in place;  step [n_agents] = Adam iteration AFTER this update;
 *   grads_scratch [n_agents][theta_size] workspace: receives only the gradients of the small tensors (biases, BN
 *              gamma/beta, first/last layers, ~6 % of the slab), consumed at the end of the same launch (learn_kernel_l)
 *              or by a second, range-restricted launch (learn_kernel_t);
 *   losses [n_agents][2] or NULL.  Same result as the two separate calls, bit for bit. Any shape avd_learn_f32 serves; B = 64.
 *   By shape: the reference widths take learn_kernel_l / learn_kernel_t (update applied where each gradient is produced); the
 *   centralized framework at L = 3 / 5 (S = 4 L, A = L, widths 320 / 160 / 64) runs as a PIPELINE -- learn kernels over chunks of
 *   256 agents in `stream`, each chunk's whole-row Adam + Polyak pass on a library-owned side stream under the next chunk's learn
 *   kernel, forked off and joined back into `stream` by events (work queued on `stream` afterwards sees every result; the call is
 *   capturable); there grads_scratch receives ALL gradients. Every other shape: the general kernel's fused form. */
int avd_learn_update_f32(const avd_mlp_layout* lay, int n_agents, const float* theta, const float* stats,
                         float* theta_out, float* theta_t, float* stats_t, float* m, float* v, const int32_t* step,
                         const float* s, const float* a, const float* r, const float* s2, float gamma, float high,
                         float actor_lr, float critic_lr, double tau, float* grads_scratch, float* losses,
                         void* stream);

/* How avd_learn_update_f32 will cut `n_agents` models of this shape into launches on the current device: the centralized
 * pipeline's chunk size (one learn workgroup per CU and chunk, at most 32 chunks), chunk count and the workgroups of its
 * Adam + Polyak pass; every other shape: one launch (chunk = n_agents, 1 chunk, 0). For callers that describe what they measured. */
int avd_learn_update_plan(const avd_mlp_layout* lay, int n_agents, int* chunk_agents, int* n_chunks, int* update_groups);

/* avd_learn_update_f32 plus the agents' NEXT actions (workers/trainer.py:287-289 of the following step, before noise
 * and clipping): next_action[v] = actor(next_state[v * x_stride ..]) with the UPDATED weights, evaluated by the
 * workgroup that has just written them (A = 1). Bit-identical to avd_actor_forward_f32 on theta_out afterwards. */
int avd_learn_update_act_f32(const avd_mlp_layout* lay, int n_agents, const float* theta, const float* stats,
                             float* theta_out, float* theta_t, float* stats_t, float* m, float* v, const int32_t* step,
                             const float* s, const float* a, const float* r, const float* s2, float gamma, float high,
                             float actor_lr, float critic_lr, double tau, float* grads_scratch, float* losses,
                             const float* next_state, int x_stride, float* next_action, void* stream);

/* update_target alone (agent/ddpgagent.py:31-55): t = w*tau + t*(1-tau) over n floats. */
int avd_polyak_f32(int64_t n, const float* w, float* t, double tau, void* stream);

/* ---- federated averaging (src/server/federated.py:18-122; workers/trainer.py:400-456) ----
 * Agents are rows of g[n_out*n_in][n]; row of (o, i) = o*stride_out + i*stride_in.
 *   interfrl (average vehicle m over platoons, trainer.py:186-187): n_out=M, n_in=P, stride_out=1, stride_in=M
 *   intrafrl (average the vehicles inside platoon p, :189-190):    n_out=P, n_in=M, stride_out=M, stride_in=1
 * out[o][j] = sum_i (w[row(o,i)] or 1) * g[row(o,i)][j], i ascending (reproducible) -- the local partial
 * sum a rank contributes to the RCCL all-reduce. weights [n_out*n_in] or NULL; wsum [n_out] or NULL. */
int avd_fed_sum_f32(int n_out, int n_in, int stride_out, int stride_in, int n, const float* g, const float* weights,
                    float* out, float* wsum, void* stream);
/* finalize: unweighted (wsum == NULL): out[o][j] /= count (tf.reduce_mean, federated.py:62);
 * weighted: out[o][j] *= (1/wsum[o]) (federated.py:110). */
int avd_fed_finalize_f32(int n_out, int n, float* out, float count, const float* wsum, void* stream);
/* scatter group rows back to agents: dst[row(o,i)][:] = src[o][:] for i >= i_begin
 * (each agent receives its group's average, trainer.py:415-425, 448-456; i_begin = 1 skips the lead
 * vehicle under intra_directional_averaging, :417-418). */
int avd_fed_scatter_f32(int n_out, int n_in, int stride_out, int stride_in, int i_begin, int n, const float* src,
                        float* dst, void* stream);

/* intrafrl + gradients in ONE pass over the gradient slab (workers/trainer.py:189-190, 341-342, 417-431): the M agents of
 * platoon p (rows p*M .. p*M + M - 1 of every slab) all take the Adam + Polyak step of avd_adam_polyak_f32 with the MEAN of
 * their M gradient rows -- avd_fed_sum_f32 + avd_fed_finalize_f32 (same summation order and scaling) + avd_fed_scatter_f32 +
 * avd_adam_polyak_f32 without the averaged slab and its copy back. weights [P*M] or NULL: the weighted mean
 * (federated.py:99-118). lead_skip != 0: intra_directional_averaging -- vehicle 0 of every platoon takes no step (no Adam, no
 * soft update, :417-418; the caller must not advance its step count either) while its gradient still enters the mean.
 * step [P*M]: the agents' Adam iteration counts, already advanced by the caller for the agents that step. */
int avd_adam_polyak_intra_f32(const avd_mlp_layout* lay, int P, int M, int lead_skip, float* theta, float* stats, float* theta_t,
                              float* stats_t, float* m, float* v, const float* grads, const int32_t* step, const float* weights,
                              float actor_lr, float critic_lr, double tau, void* stream);

/* ---- federated weights in the throughput modes (workers/trainer.py:385-398, 694; src/server/federated.py:99-118) ----
 * The reference weights agent (p, m) by |1 / mean(all_ep_reward_lists[p][m][-weighted_window:])| from `training_episode >=
 * weighted_window` on. With the episode bookkeeping on the device the history lives there too:
 *   ring [P*M][W] f32  the last W closed episodes' rewards of every agent (slot = episode number % W); hist_cnt [P] int32.
 * avd_fed_history_push_f32, once per step BEFORE the episode end / conditional reset: platoon p closes when `force` (the caller's
 * step limit), or *cond != 0 (the any-terminal flag: every platoon closes, :268-269), or -- `done` given -- done[p] or
 * ep_len[p] + 1 >= limit (per-platoon episodes: what avd_episode_end_f32 is about to apply; ep_len may be NULL). A closing
 * platoon's M counters ep_reward[p*M ..] go into its ring row (and are zeroed when zero_after: the all-platoons rule has no other
 * kernel doing it). */
int avd_fed_history_push_f32(int P, int M, int W, float* ep_reward, const uint8_t* done, const int32_t* ep_len, int limit,
                             const int32_t* cond, int force, int zero_after, float* ring, int32_t* hist_cnt, void* stream);
/* avd_fed_weights_f32: w_raw [P*M] = |1 / mean(ring row)| (get_weight, trainer.py:385-398), wsum [M] = sum over platoons
 * (fed_weight_sums), agent_weight [P*M] = w P / wsum -- the per-agent factor of avd_learn_set_split_f16x3 /
 * avd_learn_set_fused_bf16 (federated.py:99-118 as a weighted mean). host_enabled 1 / 0: weighted / plain mean (all ones,
 * wsum = P); < 0: decided here -- weighted once every platoon has closed W episodes (is_weighted_fed_enabled, :694). Deterministic
 * (fixed reduction tree). A zero episodic-reward mean gives an infinite weight, as in the reference. */
int avd_fed_weights_f32(int P, int M, int W, const float* ring, const int32_t* hist_cnt, int host_enabled, float* w_raw,
                        float* agent_weight, float* wsum, void* stream);

/* ---- one launch per training step (device-RNG mode, decentralized agents) ----------------------------------------
 * advance_environment + the replay add of train_all_models (workers/trainer.py:282-322) for all P platoons:
 *   noise  = OUActionNoise.__call__()                       src/noise.py:15-19        (= avd_ou_step_f32, Philox stream OU)
 *   action = clip(actor_out + noise, low, high)             agent/ddpgagent.py:22-27  (= avd_policy_f32)
 *   exog   = get_random_val(rand_gen, reset_max_u)          workers/trainer.py:291-295 (= avd_normal_f32 / avd_uniform_f32)
 *   states, rewards, terminal = Platoon.step(action, exog)  src/environment.py:209-241 (= avd_env_step_f32)
 *   ReplayBuffer.add((prev_state, action, reward, state))   src/replaybuffer.py:36-47  (= avd_replay_add_f32; ring may be NULL)
 *   episodic reward counters += reward                      workers/trainer.py:321     (ep_reward may be NULL)
 * Same draws (seed, call counters, indices) and the same unfused float32 arithmetic as the separate entry points: results are
 * bit-identical to calling them one after the other. any_done: this step's any-terminal flag (must be 0 on entry);
 * any_done_other (optional): a second flag word that is set to 0 (two flags used alternately need no clearing launch). */
int avd_step_fused_f32(const avd_env_consts* d_consts, int P, int L, int S, const float* x_in, float* x_out, float* prev_a,
                       float* cum_accel, float* reward, uint8_t* term, uint8_t* done, int32_t* any_done, int32_t* any_done_other,
                       const float* actor_out, float* ou_state, float* action, float* leader_exog, float ou_theta, float ou_mean,
                       float ou_dt, float ou_std_dev, float action_low, float action_high, float exog_scale, int exog_uniform,
                       uint64_t seed, uint64_t ou_counter, uint64_t exog_counter, float* ring, int cap, int64_t replay_counter,
                       float* ep_reward, void* stream);

/* ReplayBuffer.sample (src/replaybuffer.py:49-63) in one launch: avd_replay_indices (same Philox draws, bit for bit) + 
 * avd_replay_gather_f32, rows moved whole. range = min(buffer_counter, capacity). S in {3, 4}, A = 1, B a multiple of 4. */
int avd_replay_sample_f32(int n_agents, int cap, int S, int A, int B, const float* ring, int range, uint64_t seed, uint64_t counter,
                          int32_t* idx, float* s, float* a, float* r, float* s2, void* stream);

/* actor(state) for agents that SHARE n_sets weight sets (agent v uses set v % n_sets), reference widths, on the f32 matrix
 * cores (csrc/act.hip: v_mfma_f32_32x32x2_f32, exact f32 products -- agent/model.py:26-36 in the reference's arithmetic
 * class). Same values as avd_actor_forward_f32 with set_mod = n_sets up to the f32 summation order (1e-7 relative).
 * run_if_nonzero (optional): device flag; the launch does nothing when it reads 0. */
int avd_actor_forward_set_f32(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                              const float* states, int x_stride, float high, float* out, const int32_t* run_if_nonzero, void* stream);

/* ---- shared-weight-set learner (interfrl with every step federated; BASELINE config 5) -----------------
 * When the P platoons' vehicle-m agents share one weight set (workers/trainer.py:121-128, 400-431: identical initial
 * weights + identical averaged gradients every step), Trainer.learn (:472-508) followed by the federated mean
 * (src/server/federated.py:69-92) equals ONE learn over the set's P x B rows. This entry point computes that as
 * layer-wise bf16 MFMA GEMMs (f32 accumulation, f32 parameters and gradients):
 *   theta/stats/theta_t/stats_t [n_sets][...]   weight sets (same slabs as avd_learn_f32 with set_mod = n_sets)
 *   s, s2 [n_sets][rows][S], a [n_sets][rows][1], r [n_sets][rows]   SET-MAJOR batches, rows = (n_agents / n_sets) * B
 *   row_weight [n_sets][rows] or NULL   per-row factor on both loss seeds: w_p * P / sum_p w_p on platoon p's rows gives
 *                                Server.get_weighted_avg_params (src/server/federated.py:99-118); losses stay unweighted
 *   grads [n_sets][theta_size]   mean gradient per set (what avd_fed_sum + avd_fed_finalize give for per-agent gradients)
 *   losses [n_sets][2] or NULL   mean critic / actor loss per set
 *   workspace: device scratch of at least avd_learn_shared_workspace() bytes.
 * Widths: layer1/layer2 sizes multiples of 64, action layer multiple of 16, A == 1, S in {3, 4}, B multiple of 64. */
int avd_learn_shared_workspace(const avd_mlp_layout* lay, int n_agents, int n_sets, size_t* bytes);
int avd_learn_shared_bf16(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                          const float* theta_t, const float* stats_t, const float* s, const float* a, const float* r,
                          const float* s2, const float* row_weight, float gamma, float high, float* grads, float* losses,
                          void* workspace, size_t workspace_bytes, void* stream);

/* The same quantity -- Trainer.learn (workers/trainer.py:472-508) + federated mean over the platoons
 * (src/server/federated.py:47-63, 99-118; workers/trainer.py:400-431) for agents sharing n_sets weight sets -- at the
 * REFERENCE widths (layer1 256, layer2 128, action layer 48; src/config.py:112-117), as persistent kernels that keep the
 * second-layer weights in registers and stream the agents' batches through them (csrc/fset.hip). Differences from
 * avd_learn_shared_bf16: batches are AGENT-MAJOR, exactly what avd_replay_gather_f32 writes
 *   s, s2 [n_agents][B][S], a [n_agents][B], r [n_agents][B]; agent v uses weight set v % n_sets;
 *   agent_weight [n_agents] or NULL: factor on both loss seeds of the agent's rows (w_p * P / sum_p w_p = the weighted mean);
 * the result is deterministic (per-workgroup partial sums combined in a fixed order, no float atomics on the gradients).
 * bf16 GEMM operands, f32 accumulation / parameters / gradients. Other widths: AVD_E_UNSUPPORTED. */
int avd_learn_set_fused_workspace(const avd_mlp_layout* lay, int n_agents, int n_sets, size_t* bytes);
int avd_learn_set_fused_bf16(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                             const float* theta_t, const float* stats_t, const float* s, const float* a, const float* r,
                             const float* s2, const float* agent_weight, float gamma, float high, float* grads, float* losses,
                             void* workspace, size_t workspace_bytes, void* stream);

/* avd_learn_set_fused_bf16 with f32-class results (csrc/fsplit.hip): same arguments, same workspace protocol, same
 * deterministic reduction, but every matrix-product operand is carried as an fp16 PAIR hi + lo (|x - hi - lo| <= 2^-22 |x|,
 * measured worst 2^-23; static operands and row factors scaled by exact powers of two into fp16's range) and each product runs
 * as A_hi B_hi + A_lo B_hi + A_hi B_lo on v_mfma_f32_32x32x16_f16 with f32 accumulation -- two MFMAs where one operand is the
 * exact relu mask: dZ2 = g3[row] c3[n] [z2 > 0] is rank one times a mask (the output layers are one unit wide). (avd_learn_set_split_bf16x3,
 * r03's name from when the pairs were bf16, remains as a deprecated alias of avd_learn_set_split_f16x3.) The reference computes Trainer.learn in float32
 * (agent/model.py:26-36, 63-83; workers/trainer.py:472-508): this entry point is tested at 2e-5 of each gradient tensor's max
 * against the float64 oracle (tests/test_gpu_fsplit.py: measured <= 1.1e-5 at 4096 x 5; the exact-f32 kernels are asserted at
 * 1e-4), which the single-rounded bf16 operands of avd_learn_set_fused_bf16 miss by three orders of magnitude on the actor
 * gradients. A non-finite input (weights, BN statistics, s, a, r, s2) or a value that fp16 cannot hold -- a first-layer
 * activation relu(z1) >= 1023.75, |S1 w1| or |x| >= 65520 -- gives an all-NaN gradient slab (tested), never silently wrong
 * finite gradients. avd_learn_set_split_mfma_count: the wave-level v_mfma_f32_32x32x16_f16 instructions (32 768 FLOP each) one
 * call issues, from the kernels' loop structure (bench.py prices the executed matrix work from it). */
int avd_learn_set_split_mfma_count(const avd_mlp_layout* lay, int n_agents, int n_sets, unsigned long long* mfma_32x32x16);
/* The same call in two phases over ONE workspace, for callers that exchange gradients between processes
 * (workers/trainer.py:400-431 averages the critic and the actor gradient lists independently, src/server/federated.py:47-63):
 *   avd_learn_set_split_critic  operand preparation, targets, mu, critic loss and gradients, d q / d mu; writes the CRITIC block
 *                               of every set of `grads` (zeroes the slab first) and both losses;
 *   avd_learn_set_split_actor   the actor gradients from what the critic phase left in the workspace; writes the ACTOR block.
 * critic, then actor, on the same stream with the same workspace and nothing else touching it in between, is bit-identical to
 * avd_learn_set_split_f16x3 (tested). Between the two the critic block is final: a multi-GPU caller MAY start its all-reduce on a
 * side stream there (avddpg_amd/dist.py exchange_two_phase; opt-in: measured on a one-rank RCCL communicator the persistent kernels of
 * the actor phase leave the collective no CU to run beside them). */
int avd_learn_set_split_critic(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                               const float* theta_t, const float* stats_t, const float* s, const float* a, const float* r,
                               const float* s2, const float* agent_weight, float gamma, float high, float* grads, float* losses,
                               void* workspace, size_t workspace_bytes, void* stream);
int avd_learn_set_split_actor(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                              const float* s, float high, float* grads, void* workspace, size_t workspace_bytes, void* stream);
int avd_learn_set_split_workspace(const avd_mlp_layout* lay, int n_agents, int n_sets, size_t* bytes);
int avd_learn_set_split_f16x3(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                               const float* theta_t, const float* stats_t, const float* s, const float* a, const float* r,
                               const float* s2, const float* agent_weight, float gamma, float high, float* grads, float* losses,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Deprecated alias of avd_learn_set_split_f16x3 (same arguments, same results). */
int avd_learn_set_split_bf16x3(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta, const float* stats,
                               const float* theta_t, const float* stats_t, const float* s, const float* a, const float* r,
                               const float* s2, const float* agent_weight, float gamma, float high, float* grads, float* losses,
                               void* workspace, size_t workspace_bytes, void* stream);

/* actor(state) (agent/model.py:26-36, workers/trainer.py:286-289) for agents that share n_sets weight sets, as the same
 * bf16 GEMM chain: state [n_sets][rows][S] SET-MAJOR (rows = n_agents / n_sets, tightly packed S floats per row),
 * out [n_sets][rows] = tanh(.) * high. */
int avd_actor_forward_shared_workspace(const avd_mlp_layout* lay, int n_agents, int n_sets, size_t* bytes);
int avd_actor_forward_shared_bf16(const avd_mlp_layout* lay, int n_agents, int n_sets, const float* theta,
                                  const float* stats, const float* state, float high, float* out, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* D[M][Nc] (f32, ldd) = A[M][K] . B[Nc][K]^T with bf16 operands (K contiguous, K % 64 == 0) and f32 accumulation: the
 * GEMM under avd_learn_shared_bf16, exposed for parity tests. A and B must be readable up to the next multiple of 256
 * rows. */
int avd_gemm_bt_bf16(int M, int Nc, int K, const void* A, long lda, const void* B, long ldb, float* D, long ldd,
                     void* stream);

#ifdef __cplusplus
#pragma GCC visibility pop
}
#endif
#endif /* AVDDPG_HIP_H */
