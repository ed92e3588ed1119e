"""``Platoon`` with the reference's object API (src/environment.py:8-301) as a one-platoon view of
the batched HIP environment.  Draws come from the global legacy ``np.random`` stream in the
reference's order, so fixed-seed runs reproduce the reference's states."""
import numpy as np
import torch

from . import vec


class Platoon:
    def __init__(self, length, config, pl_idx, rand_states=True, evaluator_states_enabled=False):
        self.pl_idx, self.config, self.length = pl_idx, config, length
        self.rand_states, self.evaluator_states_enabled = rand_states, evaluator_states_enabled
        self._v = vec.VecPlatoon(1, length, config, rand_states=rand_states,
                                 evaluator_states_enabled=evaluator_states_enabled, rng="host", track_aux=True)
        v = self._v
        for name in ("multiplier", "hidden_multiplier", "num_models", "def_num_actions", "num_actions",
                     "def_num_states", "num_states", "number_of_reward_components", "state_lbs", "jerk_lb",
                     "exog_lbl"):
            setattr(self, name, getattr(v, name))
        self._jerk = np.zeros(length)
        self._centralized = config.framework == config.cntrl
        # (the constructor itself already reset every vehicle once, environment.py:385 -- done by VecPlatoon)
        if self.length > 6:  # environment.py:84-85: raised as the LAST constructor statement
            raise ValueError(f"Platoon of length {self.length}, but only have 6! Add more colors in environment "
                             "to work with larger platoons in rendering!")

    @property
    def front_u(self):
        return self._v.front_u[0]

    @property
    def front_accel(self):
        return self._v.front_accel[0]

    def _states(self):
        obs = self._v.observations().cpu().numpy()[0].astype(np.float64)
        states = [obs[i] for i in range(self.length)]
        if self._centralized:
            states = [list(np.concatenate(states).flat)]
        return states

    def reset(self):
        """environment.py:284-301"""
        self._v.reset()
        self._jerk[:] = 0
        return self._states()

    def step(self, actions, leader_exog=None, debug_mode=False):
        """environment.py:209-241: returns (states, rewards, platoon_done)."""
        v = self._v
        if leader_exog is None:  # :259, :265
            leader_exog = self.front_accel if self.config.model == self.config.modelA else self.front_u
        a = torch.as_tensor(np.asarray(actions, dtype=np.float32).reshape(1, self.length), device=v.device)
        e = torch.as_tensor(np.asarray([leader_exog], dtype=np.float32), device=v.device)
        pa_before = v.prev_a.clone()
        v.any_done.zero_()
        v.step(a, e)
        self._jerk = v.get_jerk_from(v.x_prev, pa_before).cpu().numpy()[0].astype(np.float64)
        rewards = [float(r) for r in v.reward.cpu().numpy()[0]]
        if self._centralized:
            rewards = [float(v.reward_mean.item())]
        return self._states(), rewards, bool(v.done.item())

    def get_jerk(self):
        """environment.py:243-251"""
        return [[j] for j in self._jerk]

    @property
    def velocity(self):
        return (self._v.cum_accel.cpu().numpy()[0] * self.config.sample_rate).astype(np.float64)

    def render(self, mode="human"):
        raise NotImplementedError("pyglet rendering is outside the hot path (SURVEY section 2, row 1)")

    def close_render(self):
        pass
