"""Actor / critic builders with the reference's call surface (agent/model.py:4-85): the returned
objects support ``model(x)`` / ``model([s, a])``, ``get_weights()``, ``set_weights(list)``,
``.weights``, ``.trainable_variables`` and ``save(path)``; arithmetic runs in the HIP kernels."""
import numpy as np
import torch

from . import _hip, params
from ._hip import call, ptr, stream_handle


class _Net:
    which = None

    def __init__(self, lay, dims, high_bound, seed_int, nominal):
        self.lay, self.dims, self.high = lay, dims, float(high_bound) if high_bound is not None else 0.0
        rs = np.random.RandomState(seed_int)  # seed_int=None -> OS entropy, like an unseeded Keras initialiser
        th, st = params.init_weights(lay, rs, nominal=nominal, dims=dims)
        self.theta = torch.from_numpy(th).cuda().reshape(1, -1)
        self.stats = torch.from_numpy(st).cuda().reshape(1, -1)

    # Keras-style accessors (host copies, Keras ordering)
    def get_weights(self):
        return params.unpack(self.lay, self.theta[0].cpu().numpy(), self.stats[0].cpu().numpy(), self.which,
                             dims=self.dims)

    @property
    def weights(self):
        return self.get_weights()

    @property
    def trainable_variables(self):
        return params.unpack(self.lay, self.theta[0].cpu().numpy(), self.stats[0].cpu().numpy(), self.which,
                             trainable_only=True, dims=self.dims)

    def set_weights(self, weights):
        th, st = self.theta[0].cpu().numpy(), self.stats[0].cpu().numpy()
        params.pack(self.lay, [np.asarray(w) for w in weights], th, st, self.which, dims=self.dims)
        self.theta.copy_(torch.from_numpy(th).reshape(1, -1))
        self.stats.copy_(torch.from_numpy(st).reshape(1, -1))

    def save(self, path):
        np.savez(path, *self.get_weights())

    def _rows(self, x, width):
        x = torch.as_tensor(np.asarray(x, dtype=np.float32) if not torch.is_tensor(x) else x, dtype=torch.float32)
        return x.reshape(-1, width).cuda().contiguous()


class ActorModel(_Net):
    which = "actor"

    def __call__(self, inputs):
        s = self._rows(inputs, self.lay.S)
        out = torch.empty(s.shape[0], self.lay.A, dtype=torch.float32, device="cuda")
        call("avd_actor_forward_f32", _hip.C.byref(self.lay), s.shape[0], 1, ptr(self.theta), ptr(self.stats), ptr(s),
             self.lay.S, self.high, ptr(out), stream_handle())
        return out


class CriticModel(_Net):
    which = "critic"

    def __call__(self, inputs):
        s = self._rows(inputs[0], self.lay.S)
        a = self._rows(inputs[1], self.lay.A)
        q = torch.empty(s.shape[0], self.lay.A, dtype=torch.float32, device="cuda")
        call("avd_critic_forward_f32", _hip.C.byref(self.lay), s.shape[0], 1, ptr(self.theta), ptr(self.stats),
             ptr(s), self.lay.S, ptr(a), ptr(q), stream_handle())
        return q


def _layout(num_states, num_actions, hidd_mult, layer1_size, layer2_size, action_layer_size, batch=64):
    """(slab layout with zero-padded widths, logical Dims) -- agent/model.py:27, 30, 65, 70: int(size * hidd_mult)."""
    dims = params.Dims(num_states, num_actions, int(layer1_size * hidd_mult), int(layer2_size * hidd_mult),
                       int(action_layer_size * hidd_mult))
    return _hip.make_layout(num_states, num_actions, *params.padded_widths(dims.H1, dims.H2, dims.Ha), batch), dims


def get_actor(num_states, num_actions, high_bound, seed_int=None, hidd_mult=1, layer1_size=400, layer2_size=300,
              action_layer_size=48):
    """agent/model.py:4-38 (``action_layer_size`` only fixes the shared slab layout)."""
    lay, dims = _layout(num_states, num_actions, hidd_mult, layer1_size, layer2_size, action_layer_size)
    return ActorModel(lay, dims, high_bound, seed_int, (layer1_size, layer2_size))


def get_critic(num_states, num_actions, hidd_mult=1, seed_int=None, layer1_size=400, layer2_size=300,
               action_layer_size=64):
    """agent/model.py:41-85 (the kernel_regularizer='l2' terms never enter the loss: workers/trainer.py:496)."""
    lay, dims = _layout(num_states, num_actions, hidd_mult, layer1_size, layer2_size, action_layer_size)
    return CriticModel(lay, dims, None, seed_int, (layer1_size, layer2_size))
