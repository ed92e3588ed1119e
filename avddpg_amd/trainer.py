"""Training loop over the HIP kernels: counterpart of reference ``workers/trainer.py``.

``VecTrainer`` reproduces the control flow of ``Trainer.run`` (:223-280), ``advance_environment``
(:282-302), ``train_all_models`` (:304-359) and the federated branches (:400-456) for P platoons at
once: all state lives in HBM, one kernel launch per stage per step, no per-platoon Python.

``Trainer`` keeps the reference's constructor / ``initialize()`` / ``run()`` / ``learn()`` names on top of
it (reporting -- CSV, plots, model files -- is out of scope of this hot path).

Weight-set regimes (the reference always holds P x M separate agents):
  * per-agent  : one weight set per (platoon, vehicle)  -- nofrl, intrafrl, and any interfrl schedule
                 in which agents can diverge between federated steps;
  * shared     : one weight set per vehicle index -- interfrl + gradients when EVERY update is a
                 federated one (fed_update_delay_steps == 1, fed_update_count == 1, no cutoff): the
                 reference's P copies then stay bit-identical (same init :121-128, same averaged
                 gradients through identical Adam states :415-425), so a single copy is exact.
"""
import logging

import numpy as np
import torch

from . import vec
from ._hip import call, ptr, stream_handle

log = logging.getLogger(__name__)


# ---- schedule predicates, same names and semantics as workers/trainer.py:631-695 ------------------
def is_fed_enabled(conf):
    return (conf.fed_method == conf.interfrl or conf.fed_method == conf.intrafrl) and (conf.framework == conf.dcntrl)


def is_gradient_updates_enabled(conf):
    return conf.aggregation_method == conf.gradients


def is_model_weight_updates_enabled(conf):
    return conf.aggregation_method == conf.weights


def is_weighted_fed_enabled(conf, training_episode):
    return conf.weighted_average_enabled and training_episode >= conf.weighted_window


def is_valid_update_episode(conf, training_episode):
    return conf.fed_enabled and (training_episode % conf.fed_update_count) == 0 and \
        training_episode <= conf.fed_cutoff_episode


def is_valid_update_step(conf, training_step):
    return (training_step % conf.fed_update_delay_steps) == 0


def is_valid_step_for_federated_training_with_gradients(conf, training_episode, training_step):
    return is_fed_enabled(conf) and is_valid_update_episode(conf, training_episode) and \
        is_valid_update_step(conf, training_step) and is_gradient_updates_enabled(conf)


def is_valid_step_for_federated_training_with_weights(conf, training_episode, training_step):
    return is_fed_enabled(conf) and is_valid_update_episode(conf, training_episode) and \
        is_valid_update_step(conf, training_step) and is_model_weight_updates_enabled(conf)


class VecTrainer:
    def __init__(self, conf, device=None, rng="device", group=None, shared_sets=None, seed=None, auto_reset=False,
                 pipeline_chunks=1, fused_update=False, shared_engine=None, init_seed=None, fused_step=None,
                 replay_ring=None, overlap_allreduce=None):
        """group: torch.distributed process group whose ranks each hold ``conf.num_platoons`` platoons
        (interfrl gradients are all-reduced over it). auto_reset: end episodes on the device (no host
        sync per step); needs rng='device'. True: the reference's rule -- any terminal platoon (or the step limit) ends
        the episode of ALL platoons (workers/trainer.py:268-269). "platoon": every platoon runs its own episodes
        (vec.VecPlatoon.episode_end: closed on its own terminal step or at its own step limit, episodic rewards kept as
        per-platoon running sums in ``env.ep_stats``) -- the vectorised-environment form for thousands of platoons, where
        the any-terminal rule cuts every episode to the first terminal among them.
        seed: this rank's env / noise / replay stream seed (give every rank its own). init_seed: seed of the initial
        weights, ``conf.random_seed`` by default -- rank-INVARIANT: every agent on every rank starts from the same
        weights (workers/trainer.py:121-131); with a group they are broadcast from rank 0 as well."""
        conf.refresh()
        self.conf, self.rng, self.group = conf, rng, group
        self.device = torch.device(device if device is not None else "cuda")
        self.P, self.L = conf.num_platoons, conf.pl_size
        seed = conf.random_seed if seed is None else seed
        self.env = vec.VecPlatoon(self.P, self.L, conf, self.device, rand_states=conf.rand_states, rng=rng, seed=seed)
        # models per platoon: L decentralized, 1 centralized (environment.py:35-42). The reference trainer iterates
        # conf.pl_size models (trainer.py:45) and therefore only completes a centralized step when pl_size == 1; for
        # pl_size > 1 this follows the loop shape of its evaluator (workers/evaluator.py:48-91: env.num_models).
        self.centralized = conf.framework == conf.cntrl
        self.M = self.env.num_models
        self.S, self.A = self.env.num_states, self.env.num_actions
        if self.centralized and conf.model == conf.modelA and self.L > 1:
            # every Vehicle is handed the platoon's num_states = 3L >= 4 and returns x[0:3L] = all 4 entries
            # (environment.py:55-63, 518): the 4L-wide observation does not fit the 3L-wide network input.
            raise ValueError("centralized + Model A with pl_size > 1: the reference's observation is 4L wide but its "
                             "network input 3L (src/environment.py:45-63, 518); not a runnable configuration")
        self.x_stride = 4 * self.L // self.M  # floats between consecutive agents' observations in env.x
        n_agents = self.P * self.M
        self.n_agents = n_agents
        self.ou = vec.VecOUNoise(n_agents, conf, self.device, rng=rng, seed=seed)
        fed = is_fed_enabled(conf)
        can_share = (fed and conf.fed_method == conf.interfrl and is_gradient_updates_enabled(conf)
                     and conf.fed_update_delay_steps == 1 and conf.fed_update_count == 1
                     and conf.fed_cutoff_ratio >= 1.0)
        self.shared = can_share if shared_sets is None else bool(shared_sets)
        if self.shared and not can_share:
            raise ValueError("shared weight sets are only exact for interfrl+gradients with every step federated")
        self.set_mod = self.M if self.shared else 0
        self.agents = vec.AgentGroup(self.M if self.shared else n_agents, self.S, self.A, conf, self.device,
                                     seed=conf.random_seed if init_seed is None else init_seed,
                                     hidd_mult=self.env.hidden_multiplier)
        from . import dist as _dist
        _dist.broadcast_agents(self.agents, group)
        # platoons over all ranks: a constant, reduced once here (the federated mean's divisor)
        self.total_platoons = _dist.total_platoons(self.P, group, self.device)
        self.replay = vec.VecReplay(n_agents, conf.buffer_size, conf.batch_size, self.S, self.A, self.device, rng=rng,
                                    seed=seed, ring=replay_ring)
        f32 = dict(dtype=torch.float32, device=self.device)
        # Shared weight sets: "per_agent" = the f32 LDS-resident kernel per agent + fed_sum (exact f32, widths up to
        # 256); "batched" = one learn over each set's P x 64 rows as bf16 MFMA GEMMs (csrc/wide.hip; any width multiple
        # of 64, e.g. BASELINE config 5's 1024); "fused" = the same quantity at the reference widths as persistent
        # register-resident-weight kernels (csrc/fset.hip; bf16 operands, deterministic, agent-major batches); "fused3" =
        # that design with every GEMM operand an exact bf16 hi + lo pair (csrc/fsplit.hip): f32-class results.
        # Default: per_agent where it exists.
        lay = self.agents.lay
        fits = lay.H2 <= 256
        self.shared_engine = shared_engine or ("batched" if (self.shared and not fits) else "per_agent")
        if self.shared_engine not in ("per_agent", "batched", "fused", "fused3"):
            raise ValueError(f"shared_engine={shared_engine!r}")
        if self.shared_engine in ("batched", "fused", "fused3") and not self.shared:
            raise ValueError("the batched learners need shared weight sets (interfrl + gradients, every step federated)")
        if self.shared_engine in ("fused", "fused3") and (lay.H1, lay.H2, lay.Ha, lay.A, lay.B) != (256, 128, 48, 1, 64):
            raise ValueError("shared_engine='fused' / 'fused3' (csrc/fset.hip, fsplit.hip) serve the reference widths 256/128/48, A = 1, batch 64 only; "
                             f"got {lay.H1}/{lay.H2}/{lay.Ha}, A = {lay.A}, batch {lay.B}: use shared_engine='batched'")
        self.actor_out = torch.zeros(n_agents, self.A, **f32)
        self.actions = torch.zeros(self.P, self.M, self.A, **f32)  # self.actions[p][m] (trainer.py:179)
        self.leader_exog = torch.zeros(self.P, **f32)
        batched = self.shared and self.shared_engine in ("batched", "fused", "fused3")  # no per-agent gradient slab (188 GB at hidden 1024)
        self.grads = None if batched else torch.zeros(n_agents, self.agents.lay.theta_size, **f32)
        self.losses = torch.zeros(n_agents, 2, **f32)
        self.ep_reward = torch.zeros(self.P, self.M, **f32)  # float32 accumulators (trainer.py:249, 321)
        self.all_ep_reward_lists = [[[] for _ in range(self.M)] for _ in range(self.P)]
        self.all_avg_reward_lists = [[[] for _ in range(self.M)] for _ in range(self.P)]
        self.fed_weights = None
        self.exog_calls = 0
        self.seed = seed
        if auto_reset not in (False, True, "platoon"):
            raise ValueError(f"auto_reset={auto_reset!r}: False, True (any-terminal, all platoons) or 'platoon'")
        self.auto_reset = auto_reset
        if auto_reset and rng != "device":
            raise ValueError("auto_reset needs rng='device'")
        # Weighted federated averaging with the episode bookkeeping on the device (r06): the weights |1 / mean(last
        # `weighted_window` episodic rewards)| (trainer.py:385-398) come from a device ring of closed-episode rewards
        # (avd_fed_history_push_f32, filled where episodes close) through avd_fed_weights_f32, once per step, no host synchronisation.
        # Enabled like the reference (`training_episode >= weighted_window`, :694): under the all-platoons episode rule when every
        # platoon has closed `weighted_window` episodes (all counts ARE the episode number); with per-platoon episodes from step
        # weighted_window x steps_per_episode on, when every platoon has surely closed that many (a host-known moment, the same on
        # every rank).
        self._dev_weighted = bool(auto_reset and fed and conf.weighted_average_enabled)
        if self._dev_weighted:
            W = int(conf.weighted_window)
            self._hist_ring = torch.zeros(n_agents, W, **f32)
            self._hist_cnt = torch.zeros(self.P, dtype=torch.int32, device=self.device)
            self._w_raw, self._aw = torch.ones(n_agents, **f32), torch.ones(n_agents, **f32)
            self._wsum = torch.full((self.M,), float(self.P), **f32)
        self.steps_total = 0  # training steps since construction (per-platoon episodes: the schedule's episode-equivalent clock)
        # fused_step: OU noise, policy clip, leader exog, platoon step, replay add and the reward counters in ONE launch
        # (avd_step_fused_f32; bit-identical to the separate kernels). Device-RNG mode, decentralized agents. Default: on
        # where it applies.
        can_fuse_step = rng == "device" and not self.centralized and self.A == 1 and self.S in (3, 4)
        self.fused_step = can_fuse_step if fused_step is None else bool(fused_step)
        if self.fused_step and not can_fuse_step:
            raise ValueError("fused_step needs rng='device' and the decentralized framework")
        # shared sets at the reference widths with a set learner: act on the f32 matrix cores (csrc/act.hip) instead of the
        # batch-1 rows kernel (same values up to the f32 summation order; the rows kernel stays wherever bit-equality with
        # the per-agent weight-set regime is asserted, i.e. the per_agent engine)
        self.act_mfma = (self.shared and self.shared_engine in ("fused", "fused3")
                         and (lay.H1, lay.H2, lay.A) == (256, 128, 1) and lay.S in (3, 4))
        # overlapped exchange (two-phase learn call, critic block all-reduced on a side stream under the actor phase): OPT-IN. Measured
        # on a real RCCL communicator (one rank, r05: bench.py --one-rank-rccl) the overlapped form is SLOWER -- 2.31 ms per step against
        # 2.18, its collectives 0.53 ms against 0.014: the learn call's persistent kernels hold every CU (one workgroup each, 135-160 KB of
        # LDS), so RCCL's kernel does not run beside the actor phase but between its launches, and the fork / join costs on top. The
        # single all-reduce of the slab between learn and Adam is the default on every backend; bench.py times both forms at N > 1.
        can_overlap = group is not None and self.shared and self.shared_engine == "fused3"
        self.overlap_allreduce = False if overlap_allreduce is None else bool(overlap_allreduce)
        if self.overlap_allreduce and not can_overlap:
            raise ValueError("overlap_allreduce needs a process group and shared_engine='fused3' (the two-phase learn call)")
        self._side = None  # side stream + buffers of the overlapped exchange, made on first use
        # the exchange buffer of the shared-set learners ([M, theta] slab | flag | [M] weight sums: dist.set_exchange_buffer) and whether
        # this step's any-terminal flag has already travelled with it
        self._xbuf = None
        self._flag_exchanged = False
        ws = 1
        if group is not None:
            import torch.distributed as _td
            ws = _td.get_world_size(group)
        self._equal_shards = abs(self.total_platoons - self.P * ws) < 0.5
        self._step_parity = 0
        self._added = False
        self.fused_update = bool(fused_update)  # nofrl (any framework / widths the learn kernels serve): avd_learn_update_f32
        # fused_update also has the learn kernel evaluate the UPDATED actor on the state the next step acts from
        # (workers/trainer.py:287-289): self.actor_out then already holds the next step's actor outputs unless the states
        # were reset in between (device flag env.any_done / a host-side reset clears _act_ready)
        self._act_ready = False
        self.pipeline_chunks = int(pipeline_chunks)  # > 1: overlap Adam/Polyak with learn across agent slices (nofrl, intrafrl + gradients)
        # intrafrl + gradients: the platoon mean formed inside the Adam pass (avd_adam_polyak_intra_f32) instead of fed_sum / finalize /
        # scatter / apply over the gradient slab (same values; False keeps the four-kernel path for cross-checks)
        self.intra_fused = True
        self.timers = None
        self.episode, self.ep_step = 0, 0
        self.updates = 0  # agent-updates (one agent's learn + Adam x2 + Polyak)
        self.env_steps = 0  # platoon-steps

    # ------------------------------------------------------------------------------------------
    def _timed(self, name, fn, *args, **kw):
        """Run fn; when self.timers is a dict, bracket it with HIP events on the launch stream."""
        if self.timers is None:
            return fn(*args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        self.timers.setdefault(name, []).append((e0, e1))
        return out

    def reset_episode(self):
        """trainer.py:244-249"""
        self._act_ready = False  # new states: the actor outputs left by the last fused update are stale
        self.env.reset()
        self.ep_reward.zero_()
        self.ep_step = 0

    def _act(self):
        """advance_environment (trainer.py:282-302): actor -> OU noise -> clip, leader exog, env step."""
        conf, P, M = self.conf, self.P, self.M
        states = self.env.x.view(P * M, self.x_stride)
        if self.shared and self.shared_engine == "batched" and self.agents.lay.H2 > 256:
            # wide shared sets: every agent re-reading its set's megabytes of weights is the wrong shape; one GEMM
            # chain per set instead (bf16 operands, like this engine's learner)
            sm = self.env.x.view(P, M, 4)[..., :self.S].transpose(0, 1).contiguous()  # set-major [M, P, S]
            o = self.agents.actor_shared(sm, P * M)
            self.actor_out.copy_(o.transpose(0, 1).reshape(P * M, 1))
        elif self._act_ready:
            # the last fused update left actor(states) in actor_out; recompute only if the episode ended since (any platoon
            # terminal resets ALL platoons, :268-269 -- the flag of the previous step is still set at this point)
            self.agents.actor(states, self.set_mod, x_stride=self.x_stride, out=self.actor_out,
                              run_if_nonzero=self.env.any_done)
        elif self.act_mfma:
            self.agents.actor_set(states, P * M, x_stride=self.x_stride, out=self.actor_out.view(-1))
        else:
            self.agents.actor(states, self.set_mod, x_stride=self.x_stride, out=self.actor_out)
        if self.fused_step:
            self._step_fused()
            return
        if self.rng == "host":
            # reference draw order per platoon: M OU normals, then the leader exog (trainer.py:286-295)
            normals = np.empty((P, M))
            exog = np.empty(P)
            for p in range(P):
                for m in range(M):
                    normals[p, m] = np.random.normal(0, 1.0)
                exog[p] = (np.random.uniform(-conf.reset_max_u, conf.reset_max_u) if conf.rand_gen == conf.uniform
                           else np.random.normal(0, conf.reset_max_u))
            noise = self.ou(normals.reshape(-1))
            self.leader_exog.copy_(torch.from_numpy(exog.astype(np.float32)))
        else:
            noise = self.ou()
            # util.get_random_val(conf.rand_gen, reset_max_u) (trainer.py:291-295): U(-u, u) or N(0, u)
            call("avd_uniform_f32" if conf.rand_gen == conf.uniform else "avd_normal_f32", P, ptr(self.leader_exog),
                 conf.reset_max_u, self.seed, self.exog_calls, stream_handle())
            self.exog_calls += 1
        if self.A > 1:  # one scalar OU process per model, broadcast over its A actions (ddpgagent.py:22)
            noise = noise.view(-1, 1).expand(-1, self.A).contiguous()
        call("avd_policy_f32", self.n_agents * self.A, ptr(self.actor_out), ptr(noise), conf.action_low,
             conf.action_high, ptr(self.actions), stream_handle())
        self.env.any_done.zero_()
        self.env.step(self.actions.view(P, self.L), self.leader_exog)

    def _step_fused(self):
        """advance_environment (workers/trainer.py:282-302) + the replay add and reward counters of train_all_models
        (:314-321) in one launch; the host only keeps the call counters in step with the separate-kernel path."""
        conf, env, ou, rp = self.conf, self.env, self.ou, self.replay
        k = self._step_parity
        self._step_parity ^= 1
        env.any_done = env._any_flags[k:k + 1]  # this step's flag (cleared by the previous step's launch, zero at start)
        other = env._any_flags[1 - k:2 - k]
        env.x, env.x_prev = env.x_prev, env.x
        call("avd_step_fused_f32", ptr(env.d_consts), self.P, self.L, self.S, ptr(env.x_prev), ptr(env.x), ptr(env.prev_a),
             ptr(env.cum_accel), ptr(env.reward), ptr(env.term), ptr(env.done), ptr(env.any_done), ptr(other),
             ptr(self.actor_out), ptr(ou.state), ptr(self.actions), ptr(self.leader_exog), conf.theta, ou.mean, conf.ou_dt,
             conf.std_dev, conf.action_low, conf.action_high, conf.reset_max_u, 1 if conf.rand_gen == conf.uniform else 0,
             self.seed, ou.calls, self.exog_calls, ptr(rp.ring), rp.cap, rp.buffer_counter, ptr(self.ep_reward), stream_handle())
        ou.calls += 1
        self.exog_calls += 1
        env.step_count += 1
        rp.buffer_counter += 1
        self._added = True

    def _weights_for_fed(self, ep):
        """trainer.py:385-398: w = |1 / mean(last `weighted_window` episodic rewards)| per agent."""
        w = np.empty((self.P, self.M), dtype=np.float32)
        for p in range(self.P):
            for m in range(self.M):
                hist = self.all_ep_reward_lists[p][m][-self.conf.weighted_window:]
                if not hist:
                    raise RuntimeError(f"weighted federated averaging at episode {ep} needs the episodic rewards of agent "
                                       f"({p}, {m}), but none were recorded (update_reward_list is called by run())")
                w[p, m] = abs(1 / np.mean(hist))
        if not np.all(np.isfinite(w)):
            raise FloatingPointError(f"non-finite federated weights at episode {ep} (an episodic-reward mean of 0?)")
        return torch.from_numpy(w).to(self.device)

    def _train(self, ep, i):
        """train_all_models + federated branches (trainer.py:304-359, 400-456)."""
        conf, P, M = self.conf, self.P, self.M
        env = self.env

        def replay_part():
            # centralized: the platoon reward (1/L) * sum of the vehicles' (environment.py:236, 281)
            reward = env.reward_mean.view(P, 1) if self.centralized else env.reward
            xs = self.x_stride
            if self._added:  # the fused step launch has already written the row and the reward counters
                self._added = False
            else:
                self.replay.add(env.x_prev.view(P * M, xs), self.actions.view(P * M, self.A), reward.view(-1),
                                env.x.view(P * M, xs), xs)
                self.ep_reward += reward
            if not self.replay.buffer_counter > conf.batch_size:  # strict gate: first update after the 65th add (:322)
                return None
            return self.replay.sample()

        batch = self._timed("replay", replay_part)
        if batch is None:
            return
        s, a, r, s2 = batch
        fed = is_fed_enabled(conf)
        self.updates += self.n_agents
        if not fed and self.fused_update:
            # nofrl: learn + Adam x2 + Polyak of every agent in one kernel (no gradient slab round trip)
            nxt = self.A == 1 and self.auto_reset  # the episode loop's host-side resets go through reset_episode()
            self._timed("learn+update", self.agents.learn_update, s, a, r, s2, self.grads, self.losses,
                        next_states=env.x.view(P * M, self.x_stride) if nxt else None, x_stride=self.x_stride,
                        next_actions=self.actor_out.view(-1) if nxt else None)
            self._act_ready = nxt
            return
        if not fed and self.pipeline_chunks > 1:
            # nofrl: every agent learns and updates locally -> software-pipeline the two kernels over agent slices
            self.agents.learn_apply(s, a, r, s2, self.grads, self.losses, chunks=self.pipeline_chunks,
                                    timers=self.timers)
            return
        if (fed and conf.fed_method == conf.intrafrl and not self.shared and self.pipeline_chunks > 1
                and is_valid_update_step(conf, i) and is_valid_step_for_federated_training_with_gradients(conf, ep, i)):
            # intrafrl + gradients, every agent stepping with its platoon's mean gradient (:417-431): learn || mean + Adam + Polyak over
            # platoon chunks on two streams (vec.AgentGroup.learn_apply_intra)
            w = self._intra_weights(ep)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if self.timers is not None:
                e0.record()
            self.agents.learn_apply_intra(s, a, r, s2, self.grads, P, M, losses=self.losses, chunks=self.pipeline_chunks, weights=w,
                                          lead_skip=bool(conf.intra_directional_averaging), timers=self.timers)
            if self.timers is not None:
                e1.record()
                self.timers.setdefault("learn+update", []).append((e0, e1))
            return
        if self.shared and self.shared_engine in ("batched", "fused", "fused3"):
            weights = None
            if self._dev_weighted:
                weights = "device"  # (self._aw / self._wsum, refreshed at the end of every step)
            elif is_weighted_fed_enabled(conf, ep):
                if self.fed_weights is None or self.fed_weights[0] != ep:
                    self.fed_weights = (ep, self._weights_for_fed(ep))
                weights = self.fed_weights[1]
            self._timed("learn", self._learn_batched, s, a, r, s2, weights)
            # the 16-bit set learners answer a non-finite input / an fp16 overflow with an all-NaN slab: the guarded update then
            # leaves that weight set untouched and counts the event (nonfinite_updates()) instead of poisoning it for good
            self._timed("update", self.agents.apply, self.set_grads, guarded=self.shared_engine in ("fused", "fused3"))
            return
        self._timed("learn", self.agents.learn, s, a, r, s2, self.set_mod, grads=self.grads, losses=self.losses)
        self._timed("update", self._update, ep, i, fed)

    def _learn_batched(self, s, a, r, s2, weights=None):
        """interfrl with every step federated, shared sets: Trainer.learn + federated mean (trainer.py:400-431) as ONE
        learn over each set's P x B rows. The sampled batch is agent-major (agent v = p*M + m); the learner wants it
        set-major. Across ranks the per-set means are combined like the per-agent path's sums (dist.exchange_fed_sums)."""
        P, M, B = self.P, self.M, self.conf.batch_size
        sm = lambda x: x.view(P, M, *x.shape[1:]).transpose(0, 1).reshape(M, P * B, *x.shape[2:]).contiguous()
        if getattr(self, "set_grads", None) is None:
            from .dist import set_exchange_buffer
            self._xbuf, self.set_grads = set_exchange_buffer(M, self.agents.lay.theta_size, self.device)
            self.set_losses = torch.zeros(M, 2, dtype=torch.float32, device=self.device)
        # the reference's any-terminal rule across ranks (workers/trainer.py:268-269): with device-side episodes the rank's flag of
        # this step travels in the gradient exchange and comes back as the global one, which the conditional reset then reads
        carry = self.env.any_done if (self.group is not None and self.auto_reset is True) else None
        rw = wsum = aw_dev = None
        if isinstance(weights, str):  # device weights: the factors and the per-set sums exist already (avd_fed_weights_f32)
            aw_dev, wsum, weights = self._aw, self._wsum, None
        elif weights is not None:  # [P, M] -> factors w_p * P / sum_p w_p (federated.py:99-118)
            wsum = weights.sum(dim=0)  # [M]
        if self.shared_engine in ("fused", "fused3"):  # agent-major batches as sampled, one factor per agent
            aw = aw_dev if aw_dev is not None else (None if weights is None else (weights * (float(P) / wsum)).reshape(P * M).contiguous())
            if self.overlap_allreduce:
                self._learn_split_overlapped(s, a, r, s2, aw, wsum, carry)
                self._flag_exchanged = carry is not None
                return
            self.agents.learn_set_fused(s, a, r, s2, P * M, grads=self.set_grads, losses=self.set_losses, agent_weight=aw,
                                        split=self.shared_engine == "fused3")
        else:
            if aw_dev is not None:
                rw = aw_dev.view(P, M).transpose(0, 1).reshape(M, P, 1).expand(M, P, B).reshape(M, P * B).contiguous()
            elif weights is not None:  # per-row factors, set-major
                rw = (weights * (float(P) / wsum)).transpose(0, 1).reshape(M, P, 1).expand(M, P, B).reshape(M, P * B).contiguous()
            self.agents.learn_shared(sm(s), sm(a), sm(r), sm(s2), P * M, grads=self.set_grads, losses=self.set_losses,
                                     row_weight=rw)
        if self.group is not None:
            from .dist import exchange_set_slab
            # ONE all-reduce(sum) of [slab | flag | weight sums]; local (weighted) mean <-> sum scaling around it (one launch after it
            # when every rank holds the same number of platoons)
            self._timed("allreduce", exchange_set_slab, self._xbuf, M, self.agents.lay.theta_size, wsum, P, self.total_platoons,
                        self.group, flag=carry, equal_shards=self._equal_shards)
            self._flag_exchanged = carry is not None

    def _learn_split_overlapped(self, s, a, r, s2, aw, wsum, flag=None):
        """The split-operand learner in its two phases (avd_learn_set_split_critic / _actor) with the exchange of the critic block
        overlapped: as soon as the critic phase has written its block of the [M, theta] slab, a side stream turns it into the
        local sum and all-reduces it while the actor phase -- a third of the learn call -- still computes on the main stream; the
        actor block (with the [M] weight sums of a weighted mean riding in the same buffer) follows on the main stream, the two
        streams join, and both blocks are divided by the global platoon count / weight. The same elementwise sums as
        exchange_fed_sums on the whole slab (workers/trainer.py:400-431 averages the critic and the actor gradient lists
        independently; src/server/federated.py:47-63)."""
        from .dist import exchange_two_phase

        P, M = self.P, self.M
        lay = self.agents.lay
        A, T = lay.actor_size, lay.theta_size
        if self._side is None:
            f32 = dict(dtype=torch.float32, device=self.device)
            self._side = dict(stream=torch.cuda.Stream(device=self.device), ready=torch.cuda.Event(),
                              crit=torch.empty(M, T - A, **f32), act=torch.empty(M * A + M + 1, **f32))
        scale = float(P) if wsum is None else wsum.view(M, 1)
        learn = lambda phase: self.agents.learn_set_fused(s, a, r, s2, P * M, grads=self.set_grads, losses=self.set_losses,
                                                          agent_weight=aw, split=True, phase=phase)
        learn("critic")
        exchange_two_phase(self.set_grads, A, scale, wsum, self.total_platoons, self.group, self._side,
                           lambda: learn("actor"), timers=self.timers, flag=flag)

    def _update(self, ep, i, fed):
        conf, P, M = self.conf, self.P, self.M
        if not fed or not is_valid_update_step(conf, i):
            # local update (:345-356); note the gate tests the step only, not the episode (SURVEY 8a FRL quirk)
            if self.shared:
                raise RuntimeError("local update requested in shared-set mode")
            self.agents.apply(self.grads)
            return
        weights = None
        if self._dev_weighted:
            weights = self._w_raw.view(P, M)  # (all ones until the weighting is enabled: the weighted formulas then give the plain mean)
        elif is_weighted_fed_enabled(conf, ep):
            if self.fed_weights is None or self.fed_weights[0] != ep:
                self.fed_weights = (ep, self._weights_for_fed(ep))
            weights = self.fed_weights[1]
        method = conf.fed_method
        if is_valid_step_for_federated_training_with_gradients(conf, ep, i):
            if method == conf.intrafrl and self.intra_fused and not self.shared:
                # the platoon's mean formed where Adam consumes it: one pass over the gradient slab (avd_adam_polyak_intra_f32)
                self.agents.apply_intra(self.grads, P, M, weights=weights, lead_skip=bool(conf.intra_directional_averaging))
                return
            avg = vec.fed_mean(self.grads, P, M, weights=weights, group=self.group, method=method,
                               total=self.total_platoons)
            if self.shared:
                self.agents.apply(avg)
            elif method == conf.intrafrl and self.intra_fused:
                pass  # (handled above: never reached)
            else:
                directional = method == conf.intrafrl and conf.intra_directional_averaging
                vec.fed_scatter(avg, self.grads, P, M, method)
                if directional:
                    # the lead vehicle of every platoon is skipped entirely on a federated step (:417-418): no Adam
                    # step, no soft update. Its slabs are saved and put back around the batched apply.
                    ag = self.agents
                    lead = lambda x: x.view(P, M, -1)[:, 0]
                    keep = [lead(x).clone() for x in (ag.theta, ag.theta_t, ag.stats_t, ag.m, ag.v)]
                    keep_step = ag.step.view(P, M)[:, 0].clone()
                    ag.apply(self.grads)
                    for x, k in zip((ag.theta, ag.theta_t, ag.stats_t, ag.m, ag.v), keep):
                        lead(x).copy_(k)
                    ag.step.view(P, M)[:, 0].copy_(keep_step)
                else:
                    self.agents.apply(self.grads)
        elif is_valid_step_for_federated_training_with_weights(conf, ep, i):
            # weights aggregation (:433-456): average `.weights` (trainables AND BN stats) per group, then write the
            # average of group [0] into EVERY agent's model and target (the reference indexes `[0]` of the server's
            # result, :442-446 -- for interfrl that is vehicle 0's cross-platoon average, for intrafrl platoon 0's)
            if self.shared:
                raise RuntimeError("weights aggregation needs per-agent weight sets")
            ag = self.agents
            tp = self.total_platoons
            avg_th = vec.fed_mean(ag.theta, P, M, weights=weights, group=self.group, method=method, total=tp)[0]
            avg_st = vec.fed_mean(ag.stats, P, M, weights=weights, group=self.group, method=method, total=tp)[0]
            directional = method == conf.intrafrl and conf.intra_directional_averaging
            for dst, src in ((ag.theta, avg_th), (ag.theta_t, avg_th), (ag.stats, avg_st), (ag.stats_t, avg_st)):
                view = dst.view(P, M, -1)
                (view[:, 1:] if directional else view).copy_(src.expand_as(view[:, 1:] if directional else view))
        # else: FRL on, valid step, but not a valid update episode -> no parameter update at all

    def step(self, ep=None, i=None, sync=True):
        """One iteration of the loop at trainer.py:251-271. Returns the any-terminal flag (host bool)
        when ``sync`` (parity mode); with auto_reset the episode bookkeeping stays on the device."""
        ep = self.episode if ep is None else ep
        i = self.ep_step if i is None else i
        self._flag_exchanged = False
        self._timed("act+env", self._act)
        self._train(ep, i)
        self.env_steps += self.P
        self.ep_step += 1
        if self.group is not None and self.auto_reset is True and not self._flag_exchanged:
            # a step whose flag did not travel with a gradient exchange (the replay gate is still closed, or an engine without the
            # slab exchange): its own 1-int all-reduce(max), still without a host synchronisation
            import torch.distributed as _td
            _td.all_reduce(self.env.any_done, op=_td.ReduceOp.MAX, group=self.group)
        self.steps_total += 1
        limit = self.conf.steps_per_episode
        if self.auto_reset == "platoon":
            # per-platoon episodes: there is no global episode; the schedule predicates (fed_update_count, fed_cutoff_episode,
            # trainer.py:631-695) see the episode-EQUIVALENT clock steps / steps_per_episode -- the episode number a platoon that
            # never terminates would be in -- and the running step
            if self._dev_weighted:
                self.env.ensure_episode_state()
                self._push_history(done=self.env.done, ep_len=self.env.ep_len, limit=limit, zero_after=0)
            self.env.episode_end(self.ep_reward, self.M, limit, any_reset=self.env.any_done)
            self.episode = self.steps_total // limit
            if self._dev_weighted:
                self._refresh_weights(1 if self.steps_total >= int(self.conf.weighted_window) * limit else 0)
            return None
        if self.auto_reset:
            # any platoon terminal ends the episode for ALL platoons (:268-269); so does the step limit
            if self.ep_step >= limit:
                if self._dev_weighted:
                    self._push_history(force=1, zero_after=1)
                self.env.reset()
                self._act_ready = False
                self.ep_step = 0
                self.episode += 1
            else:
                if self._dev_weighted:
                    self._push_history(cond=self.env.any_done, zero_after=1)
                self.env.reset(cond=self.env.any_done)
            if self._dev_weighted:
                self._refresh_weights(-1)
            return None
        if not sync:
            return None
        from .dist import any_terminal
        return any_terminal(self.env.any_done, self.group)

    def _intra_weights(self, ep):
        if self._dev_weighted:
            return self._w_raw.view(self.P, self.M)
        if is_weighted_fed_enabled(self.conf, ep):
            if self.fed_weights is None or self.fed_weights[0] != ep:
                self.fed_weights = (ep, self._weights_for_fed(ep))
            return self.fed_weights[1]
        return None

    def _push_history(self, done=None, ep_len=None, limit=0, cond=None, force=0, zero_after=0):
        """Closed-episode rewards into the device ring (avd_fed_history_push_f32), BEFORE the episode end / conditional reset."""
        call("avd_fed_history_push_f32", self.P, self.M, int(self.conf.weighted_window), ptr(self.ep_reward), ptr(done), ptr(ep_len),
             int(limit), ptr(cond), int(force), int(zero_after), ptr(self._hist_ring), ptr(self._hist_cnt), stream_handle())

    def _refresh_weights(self, host_enabled):
        """trainer.py:385-398 on the device: w, the per-set sums and the learners' per-agent factors for the NEXT step's update."""
        call("avd_fed_weights_f32", self.P, self.M, int(self.conf.weighted_window), ptr(self._hist_ring), ptr(self._hist_cnt),
             int(host_enabled), ptr(self._w_raw), ptr(self._aw), ptr(self._wsum), stream_handle())

    def nonfinite_updates(self):
        """Weight-set updates skipped because the set learner returned a NaN gradient slab (host synchronisation: call it at
        reporting points). 0 unless a state, action or reward was non-finite or beyond fp16's range."""
        n = getattr(self.agents, "nonfinite_skipped", None)
        return 0 if n is None else int(n.item())

    def update_reward_list(self, ep):
        """trainer.py:510-517 (float32 counters, trailing mean over reward_averaging_window)."""
        rew = self.ep_reward.cpu().numpy()
        for p in range(self.P):
            for m in range(self.M):
                self.all_ep_reward_lists[p][m].append(rew[p, m])
                self.all_avg_reward_lists[p][m].append(
                    np.mean(self.all_ep_reward_lists[p][m][-self.conf.reward_averaging_window:]))

    def run(self, number_of_episodes=None):
        """trainer.py:232-273"""
        conf = self.conf
        n = conf.number_of_episodes if number_of_episodes is None else number_of_episodes
        for ep in range(n):
            self.episode = ep
            self.reset_episode()
            for i in range(conf.steps_per_episode):
                if self.step(ep, i):
                    break
            self.update_reward_list(ep)
        return self.all_ep_reward_lists, self.all_avg_reward_lists


class Trainer:
    """Reference-shaped facade (workers/trainer.py:18-61, 223): ``Trainer(base_dir, timestamp, debug_enabled,
    conf)``, ``initialize()``, ``run()``. Host-RNG parity mode by default."""

    def __init__(self, base_dir, timestamp, debug_enabled, conf, rng="host", device=None):
        self.base_dir, self.timestamp, self.debug_enabled, self.conf = base_dir, timestamp, debug_enabled, conf
        self.rng, self.device = rng, device
        self.engine = None

    def initialize(self):
        self.conf.timestamp = str(self.timestamp)
        self.conf.fed_enabled = is_fed_enabled(self.conf)
        self.engine = VecTrainer(self.conf, device=self.device, rng=self.rng)
        self.num_models, self.num_platoons = self.engine.M, self.engine.P
        self.num_states, self.num_actions = self.engine.S, self.engine.A
        self.all_ep_reward_lists = self.engine.all_ep_reward_lists
        self.all_avg_reward_lists = self.engine.all_avg_reward_lists

    def run(self, number_of_episodes=None):
        return self.engine.run(number_of_episodes)


def learn(rbuffer, actor_model, critic_model, target_actor, target_critic, gamma=0.99):
    """``Trainer.learn`` (workers/trainer.py:472-508) on reference-shaped objects: samples ``rbuffer`` and
    returns (critic_grad[14], actor_grad[10]) as lists in ``trainable_variables`` order. Pure w.r.t. the
    model weights."""
    from . import _hip, params

    lay = actor_model.lay
    s, a, r, s2 = rbuffer.sample()
    A = lay.actor_size
    theta = torch.cat([actor_model.theta[:, :A], critic_model.theta[:, A:]], dim=1).contiguous()
    theta_t = torch.cat([target_actor.theta[:, :A], target_critic.theta[:, A:]], dim=1).contiguous()

    def stats_of(act, cri):
        st = act.stats.clone()
        st[:, lay.cmms:] = cri.stats[:, lay.cmms:]
        return st

    stats, stats_t = stats_of(actor_model, critic_model), stats_of(target_actor, target_critic)
    grads = torch.empty(1, lay.theta_size, dtype=torch.float32, device=theta.device)
    call("avd_learn_f32", _hip.C.byref(lay), 1, 1, ptr(theta), ptr(stats), ptr(theta_t), ptr(stats_t),
         ptr(s.reshape(1, lay.B, lay.S).contiguous()), ptr(a.reshape(1, lay.B, lay.A).contiguous()),
         ptr(r.reshape(1, lay.B).contiguous()), ptr(s2.reshape(1, lay.B, lay.S).contiguous()), gamma,
         actor_model.high, ptr(grads), None, stream_handle())
    g = grads[0].cpu().numpy()
    dummy = np.zeros(lay.stats_size, dtype=np.float32)
    dims = getattr(actor_model, "dims", None)
    return (params.unpack(lay, g, dummy, "critic", trainable_only=True, dims=dims),
            params.unpack(lay, g, dummy, "actor", trainable_only=True, dims=dims))


Trainer.learn = staticmethod(learn)
