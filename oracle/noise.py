"""Oracle: Ornstein-Uhlenbeck exploration noise (TEST INFRASTRUCTURE).

Restates reference ``src/noise.py:3-29``; defaults from ``src/config.py:101-103``.
"""
import numpy as np


class RefOUNoise:
    """Scalar object shaped like the reference class (draws from the global legacy RNG)."""

    def __init__(self, mean, std_dev=0.02, theta=0.15, dt=1e-2, x_init=None):
        self.theta, self.mean, self.dt, self.x_init = theta, mean, dt, x_init
        self.std_dev = float(std_dev) * np.ones(1)  # noise.py:9
        self.reset()

    def __call__(self):  # noise.py:14-23
        x = (self.x_prev + self.theta * (self.mean - self.x_prev) * self.dt
             + self.std_dev * np.sqrt(self.dt) * np.random.normal(0, 1.0, size=self.mean.shape))
        self.x_prev = x
        return x

    def reset(self):  # noise.py:25-29
        self.x_prev = self.x_init if self.x_init is not None else np.zeros_like(self.mean)


def batched_ou_step(x_prev, normals, std_dev=0.02, theta=0.15, dt=1e-2, mean=0.0, dtype=np.float32):
    """x' = x + theta*(mean-x)*dt + std*sqrt(dt)*n, elementwise in ``dtype`` (noise.py:15-19)."""
    t = np.dtype(dtype).type
    x_prev = np.asarray(x_prev, dtype=dtype)
    n = np.asarray(normals, dtype=dtype)
    return (x_prev + t(theta) * (t(mean) - x_prev) * t(dt) + (t(std_dev) * t(np.sqrt(dt))) * n).astype(dtype)
