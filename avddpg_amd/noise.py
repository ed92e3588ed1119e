"""``OUActionNoise`` with the reference's API (src/noise.py:3-29) over the HIP OU kernel."""
import numpy as np

from . import vec


class OUActionNoise:
    def __init__(self, mean, x_init=None, config=None):
        self.config, self.mean, self.x_init = config, np.asarray(mean, dtype=np.float64), x_init
        self.theta, self.dt = config.theta, config.ou_dt
        self.std_dev = float(config.std_dev) * np.ones(1)
        # (src/noise.py:15-19 reverts every element towards ITS mean: the kernel takes one mean per launch, so elements with different
        #  means -- never built by the reference trainer, which passes zeros: workers/trainer.py:138 -- run as one process each)
        flat = self.mean.reshape(-1)
        if flat.size and np.all(flat == flat[0]):
            self._vs = [vec.VecOUNoise(flat.size, config, rng="host", mean=float(flat[0]))]
        else:
            self._vs = [vec.VecOUNoise(1, config, rng="host", mean=float(mu)) for mu in flat]
        self.reset()

    def __call__(self):
        """noise.py:14-23 -- one N(0,1) draw per element from the global legacy RNG."""
        normals = np.random.normal(0, 1.0, size=self.mean.shape).reshape(-1)
        if len(self._vs) == 1:
            x = self._vs[0](normals).cpu().numpy()
        else:
            x = np.array([v(normals[k:k + 1]).cpu().numpy()[0] for k, v in enumerate(self._vs)])
        x = x.astype(np.float64).reshape(self.mean.shape)
        self.x_prev = x
        return x

    def reset(self):
        import torch
        self.x_prev = np.asarray(self.x_init, dtype=np.float64) if self.x_init is not None else np.zeros_like(self.mean)
        flat = torch.as_tensor(self.x_prev.reshape(-1), dtype=torch.float32)
        if len(self._vs) == 1:
            self._vs[0].state.copy_(flat)
        else:
            for k, v in enumerate(self._vs):
                v.state.copy_(flat[k:k + 1])
