"""Oracle: platoon dynamics (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates reference ``src/environment.py`` (Platoon :8-301, Vehicle :304-559) and
``src/util.py:55-70`` in two shapes:

* ``RefPlatoon`` -- one Python object per platoon with a per-vehicle loop, shaped
  like the reference (used for the golden-vector pin and the CPU baseline);
* ``batched_*`` -- the same arithmetic over ``x[P, L, 4]`` arrays in a chosen
  dtype (the comparator for the HIP kernels).
"""
from dataclasses import dataclass, field

import numpy as np

MODEL_A = "ModelA"
MODEL_B = "ModelB"


@dataclass
class EnvParams:
    """Hyper-parameters consumed by the env; defaults = reference src/config.py:39-88."""

    model: str = MODEL_B  # config.py:7
    framework: str = "decentralized"  # config.py:26
    method: str = "euler"  # config.py:47
    pl_leader_reset_a: float = 0.0  # :41
    reset_max_u: float = 0.1  # :42
    pl_leader_tau: float = 0.1  # :44
    timegap: float = 1.0  # :49
    dyn_coeff: float = 0.1  # :50
    reward_ep_coeff: float = 0.4  # :52
    reward_ev_coeff: float = 0.2
    reward_u_coeff: float = 0.2
    reward_jerk_coeff: float = 0.2  # :55
    max_ep: float = 20.0  # :57
    max_ev: float = 20.0
    reset_ep_max: float = 1.5  # :60
    reset_max_ev: float = 1.5
    reset_max_a: float = 0.05
    reset_ep_eval_max: float = 1.0  # :64
    reset_ev_eval_max: float = 1.0
    reset_a_eval_max: float = 0.03
    action_high: float = 2.5  # :68
    action_low: float = -2.5
    re_scalar: float = 1.0  # :71
    terminal_reward: float = 0.5  # :72
    can_terminate: bool = True  # :75
    rand_gen: str = "normal"  # :81
    sample_rate: float = 0.1  # :86
    centrl_hidd_mult: float = 1.2  # :95
    stand_still: float = 8.0  # environment.py:343

    @property
    def num_obs(self):  # environment.py:47-52
        return 3 if self.model == MODEL_A else 4


def system_matrices(method, T, tau, tau_lead, h):
    """A(4x4), B(4), C(4) in float64 -- reference src/environment.py:390-451."""
    e = np.exp(-T / tau)
    e_lead = np.exp(-T / tau_lead)
    if method == "euler":  # :393-408
        A = np.array([[1, T, -h * T, 0], [0, 1, -T, T], [0, 0, 1 - (T / tau), 0], [0, 0, 0, 1 - (T / tau_lead)]],
                     dtype=np.float64)
        B = np.array([0, 0, T / tau, 0], dtype=np.float64)
        C = np.array([0, 0, 0, T / tau_lead], dtype=np.float64)
    elif method == "exact":  # :410-445
        A_13 = -h * tau + h * tau * e - tau * T + tau ** 2 - (tau ** 2) * e
        A_14 = tau_lead * T - tau_lead ** 2 + (tau_lead ** 2) * e_lead
        A_23 = -tau + tau * e
        A_24 = tau_lead - tau_lead * e_lead
        A = np.array([[1, T, A_13, A_14], [0, 1, A_23, A_24], [0, 0, e, 0], [0, 0, 0, e_lead]], dtype=np.float64)
        B_11 = -h * T + h * tau * e - h * tau - (T ** 2) / 2 + tau * T + (tau ** 2) * e - tau ** 2
        B_21 = -T - tau * e + tau
        B = np.array([B_11, B_21, -e + 1, 0], dtype=np.float64)
        C_11 = (T ** 2) / 2 - tau_lead * T - (tau_lead ** 2) * e_lead + tau_lead ** 2
        C_21 = T + tau_lead * e_lead - tau_lead
        C = np.array([C_11, C_21, 0, -e_lead + 1], dtype=np.float64)
    else:
        raise ValueError(method)
    return A, B, C


def platoon_matrices(ep: EnvParams, L):
    """Per-vehicle-index (A,B,C) tables [L,4,4],[L,4],[L,4]: tau_lead of vehicle 0 is
    pl_leader_tau, of i>0 the predecessor's tau (src/environment.py:55-63)."""
    As, Bs, Cs = [], [], []
    for i in range(L):
        tau_lead = ep.pl_leader_tau if i == 0 else ep.dyn_coeff
        A, B, C = system_matrices(ep.method, ep.sample_rate, ep.dyn_coeff, tau_lead, ep.timegap)
        As.append(A), Bs.append(B), Cs.append(C)
    return np.array(As), np.array(Bs), np.array(Cs)


def get_random_val(mode, val=None, std_dev=None, size=None):
    """src/util.py:55-70 -- draws from the GLOBAL legacy numpy stream."""
    if mode == "uniform":
        return np.random.uniform(-1 * val, val)
    return np.random.normal(0, std_dev, size=size)


# ----------------------------------------------------------------------------
# reference-shaped scalar objects
# ----------------------------------------------------------------------------
class RefVehicle:
    """One vehicle; src/environment.py:304-559 (rendering/printing omitted)."""

    def __init__(self, idx, ep: EnvParams, tau_lead, a_lead, num_obs, rand_states=True, evaluator_states=True):
        self.idx, self.ep, self.num_obs = idx, ep, num_obs
        self.rand_states, self.evaluator_states = rand_states, evaluator_states
        self.u = 0
        self.reset(a_lead)  # :385 (draws happen before matrices are built)
        self.A, self.B, self.C = system_matrices(ep.method, ep.sample_rate, ep.dyn_coeff, tau_lead, ep.timegap)

    def reset(self, a_lead=None):  # :520-559
        ep = self.ep
        self.u = 0
        self.cumulative_accel = 0
        self.velocity = 0
        self.desired_headway = 0
        self.headway = 0
        self.jerk = 0
        if self.evaluator_states:
            if self.rand_states:
                self.x = np.array([ep.reset_ep_eval_max, ep.reset_ev_eval_max, ep.reset_a_eval_max, a_lead])
            else:
                self.x = np.array([ep.reset_ep_max, ep.reset_max_ev, ep.reset_max_a, a_lead])
        elif self.rand_states:
            self.x = np.array([get_random_val(ep.rand_gen, ep.reset_ep_max, std_dev=ep.reset_ep_max),
                               get_random_val(ep.rand_gen, ep.reset_max_ev, std_dev=ep.reset_max_ev),
                               get_random_val(ep.rand_gen, ep.reset_max_a, std_dev=ep.reset_max_a),
                               a_lead])
        else:
            self.x = np.array([ep.reset_ep_max, ep.reset_max_ev, ep.reset_max_a, a_lead])
        self.prev_x = self.x
        return self.x[0:self.num_obs]

    def step(self, u, exog):  # :460-518
        ep = self.ep
        self.u = u
        x = self.x
        norm_ep = abs(x[0]) / ep.max_ep
        norm_ev = abs(x[1]) / ep.max_ev
        norm_u = abs(u) / abs(ep.action_high)
        n_jerk = abs(x[2] - self.prev_x[2]) / (2 * ep.action_high)
        self.jerk = (x[2] - self.prev_x[2]) / ep.sample_rate
        self.cumulative_accel += x[2]
        self.velocity = self.cumulative_accel * ep.sample_rate
        self.desired_headway = ep.stand_still + ep.timegap * self.velocity
        self.headway = x[0] + self.desired_headway
        terminal = False
        if (abs(x[0]) > ep.max_ep or abs(x[1]) > ep.max_ev) and ep.can_terminate:  # :505
            terminal = True
            reward = ep.terminal_reward * ep.re_scalar
        else:  # :510
            reward = (ep.reward_ep_coeff * norm_ep + ep.reward_ev_coeff * norm_ev + ep.reward_u_coeff * norm_u
                      + ep.reward_jerk_coeff * n_jerk) * ep.re_scalar
        self.prev_x = x
        self.x = self.A.dot(x) + self.B.dot(u) + self.C.dot(exog)  # :513 (advances even when terminal)
        return self.x[0:self.num_obs], -reward, terminal


class RefPlatoon:
    """One platoon; src/environment.py:8-301. ``length > 6`` is allowed here (the
    reference refuses it only for rendering colours, :84-85)."""

    def __init__(self, length, ep: EnvParams, rand_states=True, evaluator_states=False):
        self.ep, self.length = ep, length
        self.front_accel = get_random_val(ep.rand_gen, ep.pl_leader_reset_a, std_dev=ep.pl_leader_reset_a)  # :24
        self.front_u = get_random_val(ep.rand_gen, ep.reset_max_u, std_dev=ep.reset_max_u)  # :32
        self.centralized = ep.framework == "centralized"
        self.num_models = 1 if self.centralized else length
        # :45-52 -- the vehicles receive the PLATOON's num_states (def x multiplier), so a centralized
        # vehicle slices x[0:S*L] = all 4 entries (also for Model A: a reference quirk kept here)
        self.num_states = ep.num_obs * (length if self.centralized else 1)
        self.num_actions = length if self.centralized else 1
        self.followers = []
        for i in range(length):  # :55-63
            if i == 0:
                self.followers.append(RefVehicle(i, ep, ep.pl_leader_tau, self.front_accel, self.num_states,
                                                 rand_states, evaluator_states))
            else:
                self.followers.append(RefVehicle(i, ep, ep.dyn_coeff, self.followers[i - 1].x[2], self.num_states,
                                                 rand_states, evaluator_states))

    def reset(self):  # :284-301
        ep = self.ep
        states = []
        self.front_accel = get_random_val(ep.rand_gen, ep.pl_leader_reset_a, std_dev=ep.pl_leader_reset_a)
        for i, f in enumerate(self.followers):
            self.front_u = get_random_val(ep.rand_gen, ep.reset_max_u, std_dev=ep.reset_max_u)
            states.append(f.reset(self.front_accel if i == 0 else self.followers[i - 1].x[2]))
        if self.centralized:
            states = [list(np.concatenate(states).flat)]
        return states

    def exogenous(self, i, leader_exog):  # :253-269
        if self.ep.model == MODEL_B:
            if i == 0:
                return self.front_u if leader_exog is None else leader_exog
            return self.followers[i - 1].u
        if i == 0:
            return self.front_accel if leader_exog is None else leader_exog
        return self.followers[i - 1].x[2]

    def step(self, actions, leader_exog=None):  # :209-241
        states, rewards, terminals = [], [], []
        for i, action in enumerate(actions):
            s, r, t = self.followers[i].step(action, self.exogenous(i, leader_exog))
            states.append(s), rewards.append(r), terminals.append(t)
        if self.centralized:
            states = [list(np.concatenate(states).flat)]
            rewards = [(1 / self.length) * sum(rewards)]  # :281
        return states, rewards, (True in terminals)

    def get_jerk(self):  # :243-251
        return [[f.jerk] for f in self.followers]


# ----------------------------------------------------------------------------
# batched functional form (comparator for the HIP kernels)
# ----------------------------------------------------------------------------
def batched_step(ep: EnvParams, x, prev_a, cum_accel, u, leader_exog, dtype=np.float32):
    """One tick of P platoons. x[P,L,4], prev_a[P,L] (= prev_x[2]), cum_accel[P,L],
    u[P,L], leader_exog[P]. Returns dict with x', prev_a', cum_accel', reward[P,L]
    (already negated), term[P,L], done[P], jerk, velocity, headway, reward_mean[P].
    Same arithmetic/order as RefVehicle.step / RefPlatoon.step, in ``dtype``."""
    dt = np.dtype(dtype).type
    x = np.asarray(x, dtype=dtype)
    P, L, _ = x.shape
    prev_a = np.asarray(prev_a, dtype=dtype)
    u = np.asarray(u, dtype=dtype)
    leader_exog = np.asarray(leader_exog, dtype=dtype)
    A, B, C = (m.astype(dtype) for m in platoon_matrices(ep, L))
    x0, x1, x2 = x[..., 0], x[..., 1], x[..., 2]
    norm_ep = np.abs(x0) / dt(ep.max_ep)
    norm_ev = np.abs(x1) / dt(ep.max_ev)
    norm_u = np.abs(u) / dt(abs(ep.action_high))
    n_jerk = np.abs(x2 - prev_a) / dt(2 * ep.action_high)
    jerk = (x2 - prev_a) / dt(ep.sample_rate)
    cum = np.asarray(cum_accel, dtype=dtype) + x2
    velocity = cum * dt(ep.sample_rate)
    headway = x0 + (dt(ep.stand_still) + dt(ep.timegap) * velocity)
    term = ((np.abs(x0) > dt(ep.max_ep)) | (np.abs(x1) > dt(ep.max_ev))) & bool(ep.can_terminate)
    reward = (dt(ep.reward_ep_coeff) * norm_ep + dt(ep.reward_ev_coeff) * norm_ev + dt(ep.reward_u_coeff) * norm_u
              + dt(ep.reward_jerk_coeff) * n_jerk) * dt(ep.re_scalar)
    reward = np.where(term, dt(ep.terminal_reward) * dt(ep.re_scalar), reward).astype(dtype)
    xn = np.empty_like(x)
    for i in range(L):  # sequential like Platoon.step :224-232 (Model A reads the predecessor's POST-step accel)
        if i == 0:
            exog = leader_exog
        elif ep.model == MODEL_B:
            exog = u[:, i - 1]
        else:
            exog = xn[:, i - 1, 2]
        Ax = np.zeros((P, 4), dtype=dtype)
        for r in range(4):
            acc = A[i, r, 0] * x[:, i, 0]
            for c in range(1, 4):
                acc = acc + A[i, r, c] * x[:, i, c]
            Ax[:, r] = acc
        xn[:, i] = Ax + B[i][None, :] * u[:, i, None] + C[i][None, :] * exog[:, None]
    neg = (-reward).astype(dtype)
    rmean = (dt(1.0) / dt(L)) * neg.sum(axis=1, dtype=dtype)
    return dict(x=xn, prev_a=x2.copy(), cum_accel=cum, reward=neg, term=term, done=term.any(axis=1), jerk=jerk,
                velocity=velocity, headway=headway, reward_mean=rmean)


def batched_reset(ep: EnvParams, draws, front_accel, mode="train", dtype=np.float32):
    """Reset P platoons from pre-drawn values. draws[P,L,3] = the three per-vehicle
    draws (already scaled: N(0,1.5),N(0,1.5),N(0,0.05) or uniform), front_accel[P].
    x3 chains the predecessor's fresh x2 (src/environment.py:291-294, 547-557).
    mode: 'train' | 'evaluator' | 'fixed' (:534-555)."""
    draws = np.asarray(draws, dtype=dtype)
    P, L, _ = draws.shape
    x = np.zeros((P, L, 4), dtype=dtype)
    if mode == "evaluator":
        x[..., 0], x[..., 1], x[..., 2] = ep.reset_ep_eval_max, ep.reset_ev_eval_max, ep.reset_a_eval_max
    elif mode == "fixed":
        x[..., 0], x[..., 1], x[..., 2] = ep.reset_ep_max, ep.reset_max_ev, ep.reset_max_a
    else:
        x[..., 0:3] = draws
    x[:, 0, 3] = np.asarray(front_accel, dtype=dtype)
    x[:, 1:, 3] = x[:, :-1, 2]
    return x, x[..., 2].copy()


def host_reset_draws(ep: EnvParams, P, L, mode="train"):
    """Consume the global legacy RNG exactly like P x Platoon.reset()
    (src/environment.py:284-301): per platoon 1 front_accel draw, then per vehicle
    1 front_u draw (+3 state draws in 'train' mode). Returns (draws[P,L,3], front_accel[P])."""
    draws = np.zeros((P, L, 3))
    fa = np.zeros(P)
    for p in range(P):
        fa[p] = get_random_val(ep.rand_gen, ep.pl_leader_reset_a, std_dev=ep.pl_leader_reset_a)
        for i in range(L):
            get_random_val(ep.rand_gen, ep.reset_max_u, std_dev=ep.reset_max_u)  # front_u: drawn, unused
            if mode == "train":
                draws[p, i, 0] = get_random_val(ep.rand_gen, ep.reset_ep_max, std_dev=ep.reset_ep_max)
                draws[p, i, 1] = get_random_val(ep.rand_gen, ep.reset_max_ev, std_dev=ep.reset_max_ev)
                draws[p, i, 2] = get_random_val(ep.rand_gen, ep.reset_max_a, std_dev=ep.reset_max_a)
    return draws, fa
