"""``OUActionNoise`` with the reference's API (src/noise.py:3-29) over the HIP OU kernel."""
import numpy as np

from . import vec


class OUActionNoise:
    def __init__(self, mean, x_init=None, config=None):
        self.config, self.mean, self.x_init = config, np.asarray(mean, dtype=np.float64), x_init
        if np.any(self.mean != 0):
            raise NotImplementedError("non-zero OU mean is never used by the reference trainer")
        self.theta, self.dt = config.theta, config.ou_dt
        self.std_dev = float(config.std_dev) * np.ones(1)
        self._v = vec.VecOUNoise(self.mean.size, config, rng="host")
        self.reset()

    def __call__(self):
        """noise.py:14-23 -- one N(0,1) draw per element from the global legacy RNG."""
        normals = np.random.normal(0, 1.0, size=self.mean.shape)
        x = self._v(normals.reshape(-1)).cpu().numpy().astype(np.float64).reshape(self.mean.shape)
        self.x_prev = x
        return x

    def reset(self):
        import torch
        self.x_prev = np.asarray(self.x_init, dtype=np.float64) if self.x_init is not None else np.zeros_like(self.mean)
        self._v.state.copy_(torch.as_tensor(self.x_prev.reshape(-1), dtype=torch.float32))
