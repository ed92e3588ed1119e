"""Batched, device-resident host API over the C ABI (include/avddpg_hip.h).

One instance of each class holds the state of P platoons x L vehicles in HBM and drives the
gfx950 kernels; Python never touches per-platoon data in the loop.  The scalar classes that
mirror the reference's object API (``environment.Platoon`` ...) are thin P = 1 views of these.

Random numbers come from one of two sources:
  * ``rng="host"``   -- the global legacy ``np.random`` stream, consumed in exactly the
                        reference's order, uploaded to the kernels (fixed-seed parity mode);
  * ``rng="device"`` -- counter-based Philox inside the kernels (throughput mode; same
                        distributions, different stream).
"""
import ctypes as C

import numpy as np
import torch

from . import _hip, dynamics, params
from ._hip import call, ptr, stream_handle


def _dev(device):
    return torch.device(device if device is not None else "cuda")


class VecPlatoon:
    """P platoons of L vehicles. Batched ``Platoon`` (reference src/environment.py:8-301)."""

    def __init__(self, num_platoons, length, config, device=None, rand_states=True, evaluator_states_enabled=False,
                 rng="host", seed=1, track_aux=False):
        self.P, self.L, self.config = int(num_platoons), int(length), config
        self.length = self.L
        self.device = _dev(device)
        self.rng, self.seed = rng, int(seed)
        self.rand_states, self.evaluator_states_enabled = rand_states, evaluator_states_enabled
        centralized = config.framework == config.cntrl
        # attributes read by callers (environment.py:35-54)
        self.multiplier = self.L if centralized else 1
        self.hidden_multiplier = config.centrl_hidd_mult if centralized else 1
        self.num_models = 1 if centralized else self.L
        self.def_num_actions = 1
        self.num_actions = self.def_num_actions * self.multiplier
        self.def_num_states = 3 if config.model == config.modelA else 4
        self.num_states = self.def_num_states * self.multiplier
        self.number_of_reward_components = 4
        self.state_lbs = {0: "$e_{pi,k}$", 1: "$e_{vi,k}$", 2: "$a_{i,k}$", 3: "$a_{i-1,k}$"}
        self.jerk_lb, self.exog_lbl = "jerk", "$u_{i,k}$"
        self.obs_width = min(4, self.num_states)  # Vehicle.step returns x[0:num_states] (:518)

        self.h_consts = dynamics.env_consts(config, self.L)
        raw = np.frombuffer(bytes(self.h_consts), dtype=np.uint8).copy()
        self.d_consts = torch.from_numpy(raw).to(self.device)
        f32 = dict(dtype=torch.float32, device=self.device)
        P, L = self.P, self.L
        self.x = torch.zeros(P, L, 4, **f32)
        self.x_prev = torch.zeros(P, L, 4, **f32)  # state before the last step (= prev_states of the trainer)
        self.prev_a = torch.zeros(P, L, **f32)
        self.cum_accel = torch.zeros(P, L, **f32) if track_aux else None
        self.reward = torch.zeros(P, L, **f32)
        self.reward_mean = torch.zeros(P, **f32) if centralized else None
        self.term = torch.zeros(P, L, dtype=torch.uint8, device=self.device)
        self.done = torch.zeros(P, dtype=torch.uint8, device=self.device)
        # two any-terminal flag words used alternately by the fused step (avd_step_fused_f32 clears the other one); the
        # separate-kernel path keeps using word 0 and clears it with a fill
        self._any_flags = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.any_done = self._any_flags[0:1]
        self.reset_count = 0
        self.step_count = 0
        self.ep_len = self.ep_stats = None  # per-platoon episodes (episode_end), made on first use
        if rng == "host":
            # constructor draws of P reference Platoon objects: 2 + 3L each (environment.py:24,32,385)
            self._host_ctor_draws()

    # -- host RNG helpers: consume np.random exactly like the reference objects --------------
    def _rand(self, val):
        c = self.config
        if c.rand_gen == c.uniform:
            return np.random.uniform(-1 * val, val)
        return np.random.normal(0, val)

    def _mode(self):
        if self.evaluator_states_enabled:
            return 1 if self.rand_states else 2
        return 0 if self.rand_states else 2

    def _host_ctor_draws(self):
        """Platoon.__init__ (:24, :32) then Vehicle.__init__ -> reset (:385): front_accel, front_u, 3 per vehicle."""
        c, P, L = self.config, self.P, self.L
        self.front_accel = np.zeros(P)
        self.front_u = np.zeros(P)
        draws = np.zeros((P, L, 3), dtype=np.float64)
        for p in range(P):
            self.front_accel[p] = self._rand(c.pl_leader_reset_a)
            self.front_u[p] = self._rand(c.reset_max_u)
            if self._mode() == 0:
                for i in range(L):
                    draws[p, i, 0] = self._rand(c.reset_ep_max)
                    draws[p, i, 1] = self._rand(c.reset_max_ev)
                    draws[p, i, 2] = self._rand(c.reset_max_a)
        self._upload_reset(draws, self.front_accel)

    def _upload_reset(self, draws, fa, cond=None):
        d = torch.from_numpy(draws.astype(np.float32)).to(self.device) if draws is not None else None
        f = torch.from_numpy(np.asarray(fa, dtype=np.float32)).to(self.device) if fa is not None else None
        call("avd_env_reset_f32", ptr(self.d_consts), self.P, self.L, ptr(self.x), ptr(self.prev_a),
             ptr(self.cum_accel), ptr(d), ptr(f), self._mode(), self.seed, self.reset_count, ptr(cond),
             stream_handle())

    def _host_reset_draws(self):
        c, P, L = self.config, self.P, self.L
        draws = np.zeros((P, L, 3), dtype=np.float64)
        fa = np.zeros(P, dtype=np.float64)
        train = self._mode() == 0
        for p in range(P):  # Platoon.reset (:284-301): front_accel, then per vehicle front_u (+3 state draws)
            fa[p] = self._rand(c.pl_leader_reset_a)
            for i in range(L):
                self.front_u[p] = self._rand(c.reset_max_u)  # kept: Platoon.step(leader_exog=None) falls back to it
                if train:
                    draws[p, i, 0] = self._rand(c.reset_ep_max)
                    draws[p, i, 1] = self._rand(c.reset_max_ev)
                    draws[p, i, 2] = self._rand(c.reset_max_a)
        self.front_accel = fa.copy()
        return draws, fa

    # -- API ------------------------------------------------------------------------------------
    def observations(self, x=None):
        """[P, L, obs_width] view (decentralized) -- Model A hides x[3] (:518)."""
        x = self.x if x is None else x
        return x[..., : self.obs_width]

    def reset(self, cond=None):
        """All platoons reset together (workers/trainer.py:246-249). ``cond``: device int32 flag --
        reset only if non-zero (device-RNG mode only)."""
        draws = fa = None
        if self.rng == "host":
            if cond is not None:
                raise ValueError("conditional reset needs rng='device'")
            draws, fa = self._host_reset_draws()
        self._upload_reset(draws, fa, cond)
        self.reset_count += 1
        return self.observations()

    def episode_end(self, ep_reward, M, limit, any_reset=None):
        """Per-platoon episode end (avd_episode_end_f32; device-RNG mode): call once after every step. Platoons whose step was
        terminal or whose episode reached ``limit`` steps are closed -- their ``ep_reward`` counters [P, M] go into
        ``self.ep_stats`` (running sums over closed episodes: platoon-mean episodic reward, length, count), and they restart
        from fresh reset states. ``any_reset`` (int32[1]) is set when any platoon was closed."""
        if self.rng != "device":
            raise ValueError("per-platoon episode ends need rng='device'")
        self.ensure_episode_state()
        st = self.ep_stats
        call("avd_episode_end_f32", ptr(self.d_consts), self.P, self.L, M, ptr(self.x), ptr(self.prev_a), ptr(self.cum_accel),
             ptr(self.done), ptr(self.ep_len), ptr(ep_reward), int(limit), ptr(st["ret_sum"]), ptr(st["len_sum"]),
             ptr(st["count"]), ptr(any_reset), self._mode(), self.seed, self.reset_count, stream_handle())
        self.reset_count += 1

    def ensure_episode_state(self):
        """The per-platoon episode counters of episode_end (made on first use)."""
        if self.ep_len is None:
            z = lambda dt: torch.zeros(self.P, dtype=dt, device=self.device)
            self.ep_len = z(torch.int32)
            self.ep_stats = dict(ret_sum=z(torch.float32), len_sum=z(torch.float32), count=z(torch.int32))

    def pop_episode_stats(self):
        """(mean platoon-mean episodic reward, mean episode length, episodes closed) since the last call; clears the sums.
        Host synchronisation: call it at reporting points, not per step."""
        st = self.ep_stats
        n = int(st["count"].sum())
        ret, ln = float(st["ret_sum"].double().sum()), float(st["len_sum"].double().sum())
        for t in st.values():
            t.zero_()
        return (ret / n if n else float("nan")), (ln / n if n else float("nan")), n

    def step(self, actions, leader_exog):
        """actions [P, L] float32 (device), leader_exog [P]. Returns (obs, reward, done) device tensors;
        ``self.x_prev`` then holds the pre-step state, ``self.any_done`` the any-terminal flag."""
        self.x, self.x_prev = self.x_prev, self.x
        call("avd_env_step_f32", ptr(self.d_consts), self.P, self.L, ptr(self.x_prev), ptr(self.x), ptr(self.prev_a),
             ptr(self.cum_accel), ptr(actions), ptr(leader_exog), ptr(self.reward), ptr(self.term), ptr(self.done),
             ptr(self.reward_mean), ptr(self.any_done), stream_handle())
        self.step_count += 1
        return self.observations(), self.reward, self.done

    def get_jerk_from(self, x_before, prev_a_before):
        """jerk of the step that consumed (x_before, prev_a_before) (environment.py:477)."""
        return (x_before[..., 2] - prev_a_before) / self.config.sample_rate


class VecOUNoise:
    """n independent scalar OU processes (reference src/noise.py)."""

    def __init__(self, n, config, device=None, rng="host", seed=1, mean=0.0):
        self.n, self.config, self.device = int(n), config, _dev(device)
        self.rng, self.seed, self.calls = rng, int(seed), 0
        self.mean = float(mean)  # the level every process reverts to (src/noise.py:7, 17; the reference trainer passes zeros)
        self.state = torch.zeros(self.n, dtype=torch.float32, device=self.device)  # x_prev = 0 (noise.py:29)

    def reset(self):
        self.state.zero_()

    def __call__(self, normals=None):
        """Advance all processes; ``normals`` [n] host array of N(0,1) draws (host-RNG mode)."""
        c = self.config
        d_n = None
        if self.rng == "host":
            if normals is None:
                normals = np.random.normal(0, 1.0, size=self.n)
            d_n = torch.from_numpy(np.asarray(normals, dtype=np.float32).reshape(self.n)).to(self.device)
        call("avd_ou_step_f32", self.n, ptr(self.state), ptr(d_n), c.theta, self.mean, c.ou_dt, c.std_dev, self.seed,
             self.calls, stream_handle())
        self.calls += 1
        return self.state


class VecReplay:
    """One ring buffer per agent: ring[n_agents][cap][2S+A+1] float32 (reference src/replaybuffer.py)."""

    def __init__(self, n_agents, buffer_capacity, batch_size, num_states, num_actions, device=None, rng="host",
                 seed=1, ring=None):
        """ring: an existing [n_agents, capacity, 2S+A+1] float32 device tensor to use instead of allocating one (two trainers
        of the same shape measured in one process share the 82 GB ring of BASELINE configs[1])."""
        self.n, self.cap, self.B = int(n_agents), int(buffer_capacity), int(batch_size)
        self.S, self.A = int(num_states), int(num_actions)
        self.row = 2 * self.S + self.A + 1
        self.device, self.rng, self.seed = _dev(device), rng, int(seed)
        if ring is not None:
            if tuple(ring.shape) != (self.n, self.cap, self.row) or ring.dtype != torch.float32 or not ring.is_contiguous():
                raise _hip.AvdError(f"replay ring must be contiguous float32 {(self.n, self.cap, self.row)}, got {tuple(ring.shape)}")
            self.ring = ring
        else:
            self.ring = torch.zeros(self.n, self.cap, self.row, dtype=torch.float32, device=self.device)
        self.buffer_counter = 0  # identical for all agents: they are written in lock step
        self.samples = 0
        f32 = dict(dtype=torch.float32, device=self.device)
        self.idx = torch.zeros(self.n, self.B, dtype=torch.int32, device=self.device)
        self.s = torch.zeros(self.n, self.B, self.S, **f32)
        self.a = torch.zeros(self.n, self.B, self.A, **f32)
        self.r = torch.zeros(self.n, self.B, **f32)
        self.s2 = torch.zeros(self.n, self.B, self.S, **f32)

    def add(self, s_prev, action, reward, s_next, x_stride):
        """s_prev / s_next: [n_agents, x_stride] device tensors (first S columns used)."""
        call("avd_replay_add_f32", self.n, self.cap, self.S, self.A, ptr(self.ring), self.buffer_counter, ptr(s_prev),
             ptr(s_next), x_stride, ptr(action), ptr(reward), stream_handle())
        self.buffer_counter += 1

    def sample_range(self):
        return min(self.buffer_counter, self.cap)  # replaybuffer.py:52

    def draw_indices(self, host_idx=None):
        """host mode: np.random.choice(range, B) per agent in agent order (replaybuffer.py:54)."""
        if self.rng == "host":
            if host_idx is None:
                rr = self.sample_range()
                host_idx = np.stack([np.random.choice(rr, self.B) for _ in range(self.n)])
            self.idx.copy_(torch.from_numpy(np.asarray(host_idx).astype(np.int32)))
        else:
            call("avd_replay_indices", self.n, self.B, self.sample_range(), self.seed, self.samples, ptr(self.idx),
                 stream_handle())
        self.samples += 1
        return self.idx

    def gather(self):
        call("avd_replay_gather_f32", self.n, self.cap, self.S, self.A, self.B, ptr(self.ring), ptr(self.idx),
             ptr(self.s), ptr(self.a), ptr(self.r), ptr(self.s2), stream_handle())
        return self.s, self.a, self.r, self.s2

    def sample(self, host_idx=None):
        """ReplayBuffer.sample (src/replaybuffer.py:49-63). Device-RNG mode: index draw + row gather in ONE launch
        (avd_replay_sample_f32: the same Philox draws as draw_indices(), bit for bit, rows moved whole)."""
        if self.rng != "host" and host_idx is None and self.A == 1 and self.S in (3, 4) and self.B % 4 == 0:
            call("avd_replay_sample_f32", self.n, self.cap, self.S, self.A, self.B, ptr(self.ring), self.sample_range(), self.seed,
                 self.samples, ptr(self.idx), ptr(self.s), ptr(self.a), ptr(self.r), ptr(self.s2), stream_handle())
            self.samples += 1
            return self.s, self.a, self.r, self.s2
        self.draw_indices(host_idx)
        return self.gather()


class AgentGroup:
    """n_sets actor/critic/target/Adam weight sets in flat slabs (reference agent/model.py,
    workers/trainer.py:100-139).  ``set_mod`` maps agent v to its weight set: 0 -> set v
    (one per agent, reference nofrl), M -> set v % M (shared per vehicle index)."""

    def __init__(self, n_sets, num_states, num_actions, config, device=None, hidd_mult=1, seed=None, high_bound=None):
        c = config
        self.config, self.device, self.n_sets = c, _dev(device), int(n_sets)
        if (c.critic_layer1_size, c.critic_layer2_size) != (c.actor_layer1_size, c.actor_layer2_size):
            raise _hip.AvdError("actor and critic layer1/layer2 sizes must match (reference defaults do)")
        # logical widths (agent/model.py:27,30,65,70: int(size * hidd_mult)); the slabs use zero-padded widths
        self.dims = params.Dims(num_states, num_actions, int(c.actor_layer1_size * hidd_mult),
                                int(c.actor_layer2_size * hidd_mult), int(c.critic_act_layer_size * hidd_mult))
        H1p, H2p, Hap = params.padded_widths(self.dims.H1, self.dims.H2, self.dims.Ha)
        self.lay = _hip.make_layout(num_states, num_actions, H1p, H2p, Hap, c.batch_size)
        self.high = float(c.action_high if high_bound is None else high_bound)
        f32 = dict(dtype=torch.float32, device=self.device)
        n, T, S = self.n_sets, self.lay.theta_size, self.lay.stats_size
        self.theta = torch.zeros(n, T, **f32)
        self.stats = torch.zeros(n, S, **f32)
        self.theta_t = torch.zeros(n, T, **f32)
        self.stats_t = torch.zeros(n, S, **f32)
        self.m = torch.zeros(n, T, **f32)
        self.v = torch.zeros(n, T, **f32)
        self.step = torch.zeros(n, dtype=torch.int32, device=self.device)
        self._layp = C.byref(self.lay)
        rs = np.random.RandomState(c.random_seed if seed is None else seed)
        th, st = params.init_weights(self.lay, rs, nominal=(c.actor_layer1_size, c.actor_layer2_size), dims=self.dims)
        # every agent starts from agent (0,0)'s weights; targets copy their online nets (trainer.py:121-131)
        self.theta.copy_(torch.from_numpy(th).to(self.device).expand(n, T))
        self.stats.copy_(torch.from_numpy(st).to(self.device).expand(n, S))
        self.theta_t.copy_(self.theta)
        self.stats_t.copy_(self.stats)

    # -- forward ----------------------------------------------------------------------------------
    def actor(self, states, set_mod, x_stride=None, out=None, target=False, run_if_nonzero=None):
        """states [n_agents, x_stride] -> tanh(.)*high [n_agents] ([n_agents, A] when A > 1) (agent/model.py:26-36).
        run_if_nonzero: int32[1] device flag; the launch is a no-op when it reads 0 (`out` then keeps what a fused update
        left there, see learn_update)."""
        n_agents = states.shape[0]
        x_stride = states.shape[-1] if x_stride is None else x_stride
        shape = (n_agents,) if self.lay.A == 1 else (n_agents, self.lay.A)
        out = torch.empty(*shape, dtype=torch.float32, device=self.device) if out is None else out
        th, st = (self.theta_t, self.stats_t) if target else (self.theta, self.stats)
        if run_if_nonzero is not None:
            call("avd_actor_forward_cond_f32", self._layp, n_agents, set_mod, ptr(th), ptr(st), ptr(states), x_stride,
                 self.high, ptr(out), ptr(run_if_nonzero), stream_handle())
        else:
            call("avd_actor_forward_f32", self._layp, n_agents, set_mod, ptr(th), ptr(st), ptr(states), x_stride,
                 self.high, ptr(out), stream_handle())
        return out

    def actor_set(self, states, n_agents, x_stride=None, out=None, run_if_nonzero=None):
        """actor(state) for n_agents agents that SHARE this group's weight sets (agent v uses set v % n_sets) on the f32
        matrix cores (csrc/act.hip; reference widths). states [n_agents, x_stride] -> [n_agents]; same values as
        actor(states, set_mod=n_sets) up to the f32 summation order."""
        x_stride = states.shape[-1] if x_stride is None else x_stride
        out = torch.empty(n_agents, dtype=torch.float32, device=self.device) if out is None else out
        call("avd_actor_forward_set_f32", self._layp, n_agents, self.n_sets, ptr(self.theta), ptr(self.stats), ptr(states), x_stride,
             self.high, ptr(out), ptr(run_if_nonzero), stream_handle())
        return out

    def critic(self, states, actions, set_mod, x_stride=None, out=None, target=False):
        n_agents = states.shape[0]
        x_stride = states.shape[-1] if x_stride is None else x_stride
        shape = (n_agents,) if self.lay.A == 1 else (n_agents, self.lay.A)
        out = torch.empty(*shape, dtype=torch.float32, device=self.device) if out is None else out
        th, st = (self.theta_t, self.stats_t) if target else (self.theta, self.stats)
        call("avd_critic_forward_f32", self._layp, n_agents, set_mod, ptr(th), ptr(st), ptr(states), x_stride,
             ptr(actions), ptr(out), stream_handle())
        return out

    # -- learn / apply ----------------------------------------------------------------------------
    def learn(self, s, a, r, s2, set_mod, grads=None, losses=None):
        """Trainer.learn (workers/trainer.py:472-508) for n_agents batches -> grads [n_agents, theta_size]."""
        n_agents = s.shape[0]
        if grads is None:
            grads = torch.empty(n_agents, self.lay.theta_size, dtype=torch.float32, device=self.device)
        call("avd_learn_f32", self._layp, n_agents, set_mod, ptr(self.theta), ptr(self.stats), ptr(self.theta_t),
             ptr(self.stats_t), ptr(s), ptr(a), ptr(r), ptr(s2), self.config.gamma, self.high, ptr(grads),
             ptr(losses), stream_handle())
        return grads

    def learn_shared(self, s, a, r, s2, n_agents, grads=None, losses=None, row_weight=None):
        """Trainer.learn + federated mean for agents that SHARE this group's ``n_sets`` weight sets (interfrl with every
        step federated), as layer-wise bf16 GEMMs over all rows of a set (csrc/wide.hip). Batches are SET-MAJOR:
        s, s2 [n_sets, rows, S], a [n_sets, rows, 1], r [n_sets, rows] with rows = n_agents / n_sets * batch_size.
        row_weight [n_sets, rows] (optional): w_p * P / sum(w) on platoon p's rows = the weighted federated mean.
        Returns the mean gradient per set [n_sets, theta_size]."""
        import ctypes
        if grads is None:
            grads = torch.empty(self.n_sets, self.lay.theta_size, dtype=torch.float32, device=self.device)
        need = ctypes.c_size_t(0)
        call("avd_learn_shared_workspace", self._layp, n_agents, self.n_sets, ctypes.byref(need))
        ws = getattr(self, "_wide_ws", None)
        if ws is None or ws.numel() < need.value:
            ws = self._wide_ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        call("avd_learn_shared_bf16", self._layp, n_agents, self.n_sets, ptr(self.theta), ptr(self.stats),
             ptr(self.theta_t), ptr(self.stats_t), ptr(s), ptr(a), ptr(r), ptr(s2), ptr(row_weight), self.config.gamma,
             self.high, ptr(grads), ptr(losses), ptr(ws), ws.numel(), stream_handle())
        return grads

    def learn_set_fused(self, s, a, r, s2, n_agents, grads=None, losses=None, agent_weight=None, split=False, phase=None):
        """Trainer.learn + federated mean for agents that SHARE this group's ``n_sets`` weight sets, at the reference
        widths, as persistent resident-weight kernels. Batches are AGENT-MAJOR as sampled (agent v = p*n_sets + m uses set
        m): s, s2 [n_agents, B, S], a [n_agents, B(, 1)], r [n_agents, B].
        agent_weight [n_agents] (optional): w_p * P / sum(w) per agent = the weighted federated mean.
        split=False: bf16 GEMM operands (csrc/fset.hip, avd_learn_set_fused_bf16);
        split=True: every operand an fp16 hi + lo pair, f32-class results (csrc/fsplit.hip, avd_learn_set_split_f16x3).
        phase (split only): None = the whole call; "critic" / "actor" = its two halves over the same workspace
        (avd_learn_set_split_critic / _actor: the critic block of ``grads`` is final after the first, the actor block after the
        second; same stream, same ``grads``, nothing else in between -- VecTrainer(overlap_allreduce=True) puts the critic block's
        all-reduce between them on a side stream).
        Returns the mean gradient per set [n_sets, theta_size]."""
        import ctypes
        self._check_agent_major(s, a, r, s2, n_agents, agent_weight)
        if grads is None:
            grads = torch.empty(self.n_sets, self.lay.theta_size, dtype=torch.float32, device=self.device)
        wsf, fn, attr = (("avd_learn_set_split_workspace", "avd_learn_set_split_f16x3", "_fsplit_ws") if split else
                         ("avd_learn_set_fused_workspace", "avd_learn_set_fused_bf16", "_fset_ws"))
        need = ctypes.c_size_t(0)
        call(wsf, self._layp, n_agents, self.n_sets, ctypes.byref(need))
        ws = getattr(self, attr, None)
        if ws is None or ws.numel() < need.value:
            ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
            setattr(self, attr, ws)
        if phase is not None and not split:
            raise _hip.AvdError("phase= needs split=True (csrc/fsplit.hip)")
        if phase == "actor":
            call("avd_learn_set_split_actor", self._layp, n_agents, self.n_sets, ptr(self.theta), ptr(self.stats), ptr(s), self.high,
                 ptr(grads), ptr(ws), ws.numel(), stream_handle())
            return grads
        if phase not in (None, "critic"):
            raise _hip.AvdError(f"phase={phase!r}")
        call("avd_learn_set_split_critic" if phase == "critic" else fn, self._layp, n_agents, self.n_sets, ptr(self.theta), ptr(self.stats),
             ptr(self.theta_t), ptr(self.stats_t), ptr(s), ptr(a), ptr(r), ptr(s2), ptr(agent_weight), self.config.gamma, self.high,
             ptr(grads), ptr(losses), ptr(ws), ws.numel(), stream_handle())
        return grads

    def learn_set_split(self, s, a, r, s2, n_agents, grads=None, losses=None, agent_weight=None):
        """learn_set_fused with f32-class results (split operands, csrc/fsplit.hip)."""
        return self.learn_set_fused(s, a, r, s2, n_agents, grads=grads, losses=losses, agent_weight=agent_weight, split=True)

    def _check_agent_major(self, s, a, r, s2, n_agents, agent_weight):
        """The set learners of csrc/fset.hip / fsplit.hip take raw pointers to tightly packed AGENT-major f32 batches
        [n_agents, B, S]: a set-major or strided view (what learn_shared wants) would silently give wrong gradients."""
        lay = self.lay
        want = {"s": (n_agents, lay.B, lay.S), "s2": (n_agents, lay.B, lay.S), "r": (n_agents, lay.B)}
        for name, x in (("s", s), ("s2", s2), ("r", r)):
            if tuple(x.shape) != want[name] or x.dtype != torch.float32 or not x.is_contiguous():
                raise _hip.AvdError(f"set learner: {name} must be contiguous float32 {want[name]} (agent-major), got "
                                    f"{tuple(x.shape)} {x.dtype} contiguous={x.is_contiguous()}")
        if a.numel() != n_agents * lay.B * lay.A or a.dtype != torch.float32 or not a.is_contiguous() or a.shape[0] != n_agents:
            raise _hip.AvdError(f"set learner: a must be contiguous float32 [{n_agents}, {lay.B}(, {lay.A})], got {tuple(a.shape)} {a.dtype}")
        if n_agents <= 0 or n_agents % self.n_sets:
            raise _hip.AvdError(f"set learner: n_agents={n_agents} is not a multiple of n_sets={self.n_sets}")
        if agent_weight is not None and (agent_weight.numel() != n_agents or agent_weight.dtype != torch.float32
                                         or not agent_weight.is_contiguous()):
            raise _hip.AvdError(f"set learner: agent_weight must be contiguous float32 [{n_agents}]")

    def actor_shared(self, states_set_major, n_agents, out=None):
        """actor(state) for agents sharing this group's weight sets as one bf16 GEMM chain per set (csrc/wide.hip):
        states [n_sets, rows, S] set-major, tightly packed -> [n_sets, rows]."""
        import ctypes
        rows = n_agents // self.n_sets
        if out is None:
            out = torch.empty(self.n_sets, rows, dtype=torch.float32, device=self.device)
        need = ctypes.c_size_t(0)
        call("avd_actor_forward_shared_workspace", self._layp, n_agents, self.n_sets, ctypes.byref(need))
        ws = getattr(self, "_wide_act_ws", None)
        if ws is None or ws.numel() < need.value:
            ws = self._wide_act_ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        call("avd_actor_forward_shared_bf16", self._layp, n_agents, self.n_sets, ptr(self.theta), ptr(self.stats),
             ptr(states_set_major), self.high, ptr(out), ptr(ws), ws.numel(), stream_handle())
        return out

    def apply(self, grads, guarded=False):
        """critic Adam, actor Adam, then Polyak (workers/trainer.py:348-356) for every weight set. guarded: a set whose gradient
        slab is NaN (what the 16-bit set learners write for a non-finite input or an fp16 overflow) takes no step at all and is
        counted in ``self.nonfinite_skipped`` (device int32; no host synchronisation) -- avd_adam_polyak_guarded_f32."""
        c = self.config
        self.step += 1
        if guarded:
            if getattr(self, "nonfinite_skipped", None) is None:
                self.nonfinite_skipped = torch.zeros(1, dtype=torch.int32, device=self.device)
            call("avd_adam_polyak_guarded_f32", self._layp, self.n_sets, ptr(self.theta), ptr(self.stats), ptr(self.theta_t),
                 ptr(self.stats_t), ptr(self.m), ptr(self.v), ptr(grads), ptr(self.step), c.actor_lr, c.critic_lr,
                 float(c.tau), ptr(self.nonfinite_skipped), stream_handle())
            return
        call("avd_adam_polyak_f32", self._layp, self.n_sets, ptr(self.theta), ptr(self.stats), ptr(self.theta_t),
             ptr(self.stats_t), ptr(self.m), ptr(self.v), ptr(grads), ptr(self.step), c.actor_lr, c.critic_lr,
             float(c.tau), stream_handle())

    def apply_intra(self, grads, P, M, weights=None, lead_skip=False, lo=0, hi=None, stream=None, advance=True):
        """intrafrl + gradients (workers/trainer.py:417-431) for platoons [lo, hi): every agent of a platoon steps with the (weighted)
        mean of the platoon's M gradient rows, averaged where it is consumed (avd_adam_polyak_intra_f32: one pass over the slab;
        same values as fed_mean + fed_scatter + apply). lead_skip: intra_directional_averaging -- vehicle 0 of every platoon takes
        no step at all (:417-418). weights [P, M] or None."""
        c = self.config
        hi = P if hi is None else hi
        if self.n_sets != P * M:
            raise _hip.AvdError("apply_intra needs one weight set per agent (P * M sets)")
        if advance:  # the Adam iteration counts of the agents that step (advance=False: the caller has done it for all platoons)
            if stream is None:
                self.step.view(P, M)[lo:hi, (1 if lead_skip else 0):] += 1
            else:  # (on the stream the kernel runs on: it reads the counts)
                with torch.cuda.stream(stream):
                    self.step.view(P, M)[lo:hi, (1 if lead_skip else 0):] += 1
        a, b = lo * M, hi * M
        call("avd_adam_polyak_intra_f32", self._layp, hi - lo, M, 1 if lead_skip else 0, ptr(self.theta[a:b]), ptr(self.stats[a:b]),
             ptr(self.theta_t[a:b]), ptr(self.stats_t[a:b]), ptr(self.m[a:b]), ptr(self.v[a:b]), ptr(grads[a:b]), ptr(self.step[a:b]),
             ptr(None if weights is None else weights.reshape(-1)[a:b]), c.actor_lr, c.critic_lr, float(c.tau),
             stream_handle() if stream is None else _hip.C.c_void_p(stream.cuda_stream))

    def learn_apply_intra(self, s, a, r, s2, grads, P, M, losses=None, chunks=8, weights=None, lead_skip=False, timers=None):
        """intrafrl + gradients as a two-stream pipeline over PLATOON chunks (platoons are the unit of independence: the mean never
        leaves a platoon): chunk c's learn kernel (matrix-core bound, gradients to the slab) runs on the caller's stream, its
        mean + Adam + Polyak pass (an HBM stream, apply_intra) on a side stream under chunk c + 1's learn kernel. Same results as
        learn() followed by apply_intra()."""
        main = torch.cuda.current_stream()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        c = self.config
        chunks = max(1, min(chunks, P))
        bounds = [(P * i) // chunks for i in range(chunks + 1)]
        T = lambda: torch.cuda.Event(enable_timing=timers is not None)
        # the iteration counts of every stepping agent, ONCE, on the caller's stream: each chunk's pass on the side stream waits for an
        # event recorded after this (one small launch per step instead of one per chunk)
        self.step.view(P, M)[:, (1 if lead_skip else 0):] += 1
        for i in range(chunks):
            lo, hi = bounds[i] * M, bounds[i + 1] * M
            if hi == lo:
                continue
            t0, t1 = T(), T()
            if timers is not None:
                t0.record(main)
            call("avd_learn_f32", self._layp, hi - lo, 0, ptr(self.theta[lo:hi]), ptr(self.stats[lo:hi]), ptr(self.theta_t[lo:hi]),
                 ptr(self.stats_t[lo:hi]), ptr(s[lo:hi]), ptr(a[lo:hi]), ptr(r[lo:hi]), ptr(s2[lo:hi]), c.gamma, self.high,
                 ptr(grads[lo:hi]), ptr(losses[lo:hi]) if losses is not None else None, _hip.C.c_void_p(main.cuda_stream))
            t1.record(main)
            side.wait_event(t1)
            u0, u1 = T(), T()
            if timers is not None:
                u0.record(side)
            self.apply_intra(grads, P, M, weights=weights, lead_skip=lead_skip, lo=bounds[i], hi=bounds[i + 1], stream=side, advance=False)
            if timers is not None:
                u1.record(side)
                timers.setdefault("learn", []).append((t0, t1))
                timers.setdefault("update", []).append((u0, u1))
        done = torch.cuda.Event()
        done.record(side)
        main.wait_event(done)

    def learn_update(self, s, a, r, s2, grads, losses=None, next_states=None, x_stride=None, next_actions=None):
        """Fused Trainer.learn + Adam x2 + update_target for per-agent weight sets (reference nofrl,
        workers/trainer.py:325-356) in ONE kernel: every gradient is consumed where it is produced, the updated
        weights go to the alternate slab (theta ping-pong: all forward/backward passes of the step read the
        pre-update weights, trainer.py:492-506). Same result as learn() followed by apply().
        next_states [n, x_stride] + next_actions [n]: also leaves actor(next_states) of the UPDATED weights in next_actions
        (bit-identical to actor() called afterwards; saves that launch's re-read of every actor from HBM)."""
        n = self.n_sets
        if s.shape[0] != n:
            raise _hip.AvdError("learn_update needs one weight set per agent (set_mod == 0)")
        if getattr(self, "theta_alt", None) is None:
            self.theta_alt = self.theta.clone()  # alignment padding stays zero in both slabs
        c = self.config
        self.step += 1
        args = (self._layp, n, ptr(self.theta), ptr(self.stats), ptr(self.theta_alt), ptr(self.theta_t), ptr(self.stats_t),
                ptr(self.m), ptr(self.v), ptr(self.step), ptr(s), ptr(a), ptr(r), ptr(s2), c.gamma, self.high, c.actor_lr,
                c.critic_lr, float(c.tau), ptr(grads), ptr(losses))
        if next_actions is not None:
            # + the agents' next actions actor(next_states) with the updated weights, from the workgroup that wrote them
            if self.lay.A != 1:
                raise _hip.AvdError("next-action epilogue: A == 1 only")
            xs = next_states.shape[-1] if x_stride is None else x_stride
            call("avd_learn_update_act_f32", *args, ptr(next_states), xs, ptr(next_actions), stream_handle())
        else:
            call("avd_learn_update_f32", *args, stream_handle())
        self.theta, self.theta_alt = self.theta_alt, self.theta

    def learn_apply(self, s, a, r, s2, grads, losses=None, chunks=4, timers=None):
        """learn + local update for per-agent weight sets (reference nofrl, workers/trainer.py:325-356).
        Agents are independent, so they are processed in `chunks` slices: the Adam/Polyak kernel of slice c
        (an HBM stream) runs on a side HIP stream underneath the learn kernel of slice c+1 (matrix-core
        bound, ~1/5 of HBM bandwidth). Results are identical to learn() followed by apply().
        timers: optional dict of lists collecting (start, end) event pairs per kernel ("learn", "update")."""
        n = self.n_sets
        if s.shape[0] != n:
            raise _hip.AvdError("learn_apply needs one weight set per agent (set_mod == 0)")
        main = torch.cuda.current_stream()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        c = self.config
        self.step += 1
        chunks = max(1, min(chunks, n))
        bounds = [(n * i) // chunks for i in range(chunks + 1)]
        T = lambda: torch.cuda.Event(enable_timing=timers is not None)
        for i in range(chunks):
            lo, hi = bounds[i], bounds[i + 1]
            if hi == lo:
                continue
            t0, t1 = T(), T()
            if timers is not None:
                t0.record(main)
            call("avd_learn_f32", self._layp, hi - lo, 0, ptr(self.theta[lo:hi]), ptr(self.stats[lo:hi]),
                 ptr(self.theta_t[lo:hi]), ptr(self.stats_t[lo:hi]), ptr(s[lo:hi]), ptr(a[lo:hi]), ptr(r[lo:hi]),
                 ptr(s2[lo:hi]), c.gamma, self.high, ptr(grads[lo:hi]), ptr(losses[lo:hi]) if losses is not None else None,
                 _hip.C.c_void_p(main.cuda_stream))
            t1.record(main)
            side.wait_event(t1)
            u0, u1 = T(), T()
            if timers is not None:
                u0.record(side)
            call("avd_adam_polyak_f32", self._layp, hi - lo, ptr(self.theta[lo:hi]), ptr(self.stats[lo:hi]),
                 ptr(self.theta_t[lo:hi]), ptr(self.stats_t[lo:hi]), ptr(self.m[lo:hi]), ptr(self.v[lo:hi]),
                 ptr(grads[lo:hi]), ptr(self.step[lo:hi]), c.actor_lr, c.critic_lr, float(c.tau),
                 _hip.C.c_void_p(side.cuda_stream))
            if timers is not None:
                u1.record(side)
                timers.setdefault("learn", []).append((t0, t1))
                timers.setdefault("update", []).append((u0, u1))
        done = torch.cuda.Event()
        done.record(side)
        main.wait_event(done)

    # -- Keras-style weight access (host copies) ---------------------------------------------------
    def get_weights(self, set_idx, which, target=False, trainable_only=False):
        th, st = (self.theta_t, self.stats_t) if target else (self.theta, self.stats)
        return params.unpack(self.lay, th[set_idx].cpu().numpy(), st[set_idx].cpu().numpy(), which, trainable_only,
                             dims=self.dims)

    def set_weights(self, set_idx, which, weights, target=False):
        th, st = (self.theta_t, self.stats_t) if target else (self.theta, self.stats)
        h_th, h_st = th[set_idx].cpu().numpy(), st[set_idx].cpu().numpy()
        params.pack(self.lay, weights, h_th, h_st, which, dims=self.dims)
        th[set_idx].copy_(torch.from_numpy(h_th))
        st[set_idx].copy_(torch.from_numpy(h_st))

    def grads_as_lists(self, grads_row):
        """One row of a grads slab -> (critic_grad[14], actor_grad[10]) in trainable_variables order."""
        g = grads_row.cpu().numpy()
        dummy = np.zeros(self.lay.stats_size, dtype=np.float32)
        return (params.unpack(self.lay, g, dummy, "critic", trainable_only=True, dims=self.dims),
                params.unpack(self.lay, g, dummy, "actor", trainable_only=True, dims=self.dims))


def fed_mean(grads, P, M, weights=None, group=None, method="interfrl", total=None):
    """Federated average of per-agent rows grads[P*M, n] (agent id v = p*M + m)
    (reference src/server/federated.py:47-63 / :99-118).
      interfrl: mean over platoons  -> [M, n];  intrafrl: mean over a platoon's vehicles -> [P, n].
    With a torch.distributed ``group`` whose ranks each hold P platoons, interfrl's local sums are
    all-reduced (RCCL over xGMI) before the division; intrafrl never leaves the GPU."""
    n = grads.shape[-1]
    if method == "interfrl":
        n_out, n_in, so, si = M, P, 1, M
    elif method == "intrafrl":
        n_out, n_in, so, si = P, M, M, 1
    else:
        raise ValueError(method)
    out = torch.empty(n_out, n, dtype=torch.float32, device=grads.device)
    wsum = torch.empty(n_out, dtype=torch.float32, device=grads.device) if weights is not None else None
    call("avd_fed_sum_f32", n_out, n_in, so, si, n, ptr(grads), ptr(weights), ptr(out), ptr(wsum), stream_handle())
    count = float(n_in)
    if group is not None and method == "interfrl":
        from .dist import exchange_fed_sums
        count = exchange_fed_sums(out, wsum, n_in, group, total=total)  # total: cached platoon count over all ranks
    call("avd_fed_finalize_f32", n_out, n, ptr(out), count, ptr(wsum), stream_handle())
    return out


def fed_scatter(avg, dst, P, M, method="interfrl", i_begin=0):
    """Write each group's average back to its member agents' rows of dst[P*M, n]."""
    n = avg.shape[-1]
    n_out, n_in, so, si = (M, P, 1, M) if method == "interfrl" else (P, M, M, 1)
    call("avd_fed_scatter_f32", n_out, n_in, so, si, i_begin, n, ptr(avg), ptr(dst), stream_handle())
    return dst
