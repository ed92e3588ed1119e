"""``ReplayBuffer`` with the reference's API (src/replaybuffer.py:5-63) over the HIP ring/gather kernels."""
import numpy as np
import torch

from . import vec


class ReplayBuffer:
    def __init__(self, buffer_capacity=100000, batch_size=64, num_states=None, num_actions=None, platoon_size=None):
        self.buffer_capacity, self.batch_size = buffer_capacity, batch_size
        self.num_states, self.num_actions = num_states, num_actions
        self._v = vec.VecReplay(1, buffer_capacity, batch_size, num_states, num_actions, rng="host")

    @property
    def buffer_counter(self):
        return self._v.buffer_counter

    def add(self, obs_tuple):
        """replaybuffer.py:37-47: (s, a, r, s') written at counter % capacity."""
        dev = self._v.device
        f = lambda x, n: torch.as_tensor(np.asarray(x, dtype=np.float32).reshape(1, n), device=dev)
        self._v.add(f(obs_tuple[0], self.num_states), f(obs_tuple[1], self.num_actions),
                    f(obs_tuple[2], 1).reshape(1), f(obs_tuple[3], self.num_states), self.num_states)

    def sample(self):
        """replaybuffer.py:50-63: np.random.choice(range, B) (with replacement) then four gathers.
        Returns device tensors s[B,S], a[B,A], r[B,1], s2[B,S] (float32)."""
        s, a, r, s2 = self._v.sample()
        return s[0].clone(), a[0].clone(), r[0].clone().reshape(-1, 1), s2[0].clone()
