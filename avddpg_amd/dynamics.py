"""Host-side constants of the platoon model: discretised system matrices and the constants
block handed to the HIP kernels.  Follows reference ``src/environment.py:390-451`` (matrices),
``:55-63`` (tau chaining) and ``src/config.py:39-75``; computed once in float64, rounded to
float32 when packed into ``avd_env_consts``."""
import math

import numpy as np

from . import _hip


def system_matrices(method, T, tau, tau_lead, h):
    """(A[4,4], B[4], C[4]) float64 for one vehicle; ``euler`` (:393-408) or ``exact`` (:410-445)."""
    if method == "euler":
        A = [[1, T, -h * T, 0], [0, 1, -T, T], [0, 0, 1 - (T / tau), 0], [0, 0, 0, 1 - (T / tau_lead)]]
        B = [0, 0, T / tau, 0]
        Cm = [0, 0, 0, T / tau_lead]
    elif method == "exact":
        e, el = math.exp(-T / tau), math.exp(-T / tau_lead)
        a13 = -h * tau + h * tau * e - tau * T + tau ** 2 - (tau ** 2) * e
        a14 = tau_lead * T - tau_lead ** 2 + (tau_lead ** 2) * el
        a23 = -tau + tau * e
        a24 = tau_lead - tau_lead * el
        A = [[1, T, a13, a14], [0, 1, a23, a24], [0, 0, e, 0], [0, 0, 0, el]]
        b11 = -h * T + h * tau * e - h * tau - (T ** 2) / 2 + tau * T + (tau ** 2) * e - tau ** 2
        b21 = -T - tau * e + tau
        B = [b11, b21, -e + 1, 0]
        c11 = (T ** 2) / 2 - tau_lead * T - (tau_lead ** 2) * el + tau_lead ** 2
        c21 = T + tau_lead * el - tau_lead
        Cm = [c11, c21, 0, -el + 1]
    else:
        raise ValueError(f"unknown discretisation method {method!r}")
    return np.array(A, dtype=np.float64), np.array(B, dtype=np.float64), np.array(Cm, dtype=np.float64)


def env_consts(conf, L):
    """Pack an ``avd_env_consts`` for a platoon of L vehicles."""
    if not 1 <= L <= _hip.AVD_MAX_L:
        raise ValueError(f"platoon length {L} outside 1..{_hip.AVD_MAX_L}")
    c = _hip.EnvConsts()
    c.L = L
    c.model_a = 1 if conf.model == conf.modelA else 0
    c.can_terminate = 1 if conf.can_terminate else 0
    c.uniform_reset = 1 if conf.rand_gen == conf.uniform else 0
    c.max_ep, c.max_ev = conf.max_ep, conf.max_ev
    c.abs_action_high = abs(conf.action_high)
    c.two_max_a = 2 * conf.action_high
    c.T = conf.sample_rate
    c.ca, c.cb, c.cc, c.cd = (conf.reward_ep_coeff, conf.reward_ev_coeff, conf.reward_u_coeff,
                              conf.reward_jerk_coeff)
    c.re_scalar, c.terminal_reward = conf.re_scalar, conf.terminal_reward
    c.stand_still, c.timegap = 8.0, conf.timegap
    c.reset_ep_max, c.reset_max_ev, c.reset_max_a = conf.reset_ep_max, conf.reset_max_ev, conf.reset_max_a
    c.reset_ep_eval, c.reset_ev_eval, c.reset_a_eval = (conf.reset_ep_eval_max, conf.reset_ev_eval_max,
                                                         conf.reset_a_eval_max)
    c.leader_reset_a = conf.pl_leader_reset_a
    for i in range(L):
        tau_lead = conf.pl_leader_tau if i == 0 else conf.dyn_coeff  # followers chain the predecessor's tau
        A, B, Cm = system_matrices(conf.method, conf.sample_rate, conf.dyn_coeff, tau_lead, conf.timegap)
        if c.model_a and Cm[2] != 0.0:
            raise ValueError("Model A chain needs C[2] == 0 (acceleration row independent of the exogenous input)")
        for k in range(16):
            c.A[i][k] = A[k // 4][k % 4]
        for k in range(4):
            c.B[i][k] = B[k]
            c.C[i][k] = Cm[k]
    return c
