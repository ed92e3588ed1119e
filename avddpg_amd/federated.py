"""``federated.Server``: the reference's averaging server (src/server/federated.py:13-122) over the HIP kernels.

Same constructor and call signatures as the reference class::

    server = Server(name, debug_enabled)
    avg  = server.get_avg_params(system_params)                       # :18-67
    wavg = server.get_weighted_avg_params(system_params, weight_sums) # :69-122

``system_params[g][i][layer]``: for every system ``g`` (interfrl: a vehicle index, workers/trainer.py:141-171) the
members ``i`` (interfrl: the platoons), each a list of per-layer arrays (gradients or weights). Returned is one list of
per-layer float32 arrays per system: the mean over its members (tf.reduce_mean, :62), or -- members arriving PRE-multiplied
by their weight (workers/trainer.py:372-377) -- ``(1 / weight_sums[g]) * sum`` (:109-110).

The arithmetic runs on the GPU through the C ABI (``avd_fed_sum_f32`` + ``avd_fed_finalize_f32``, include/avddpg_hip.h): the
members of all systems are flattened into one ``[systems * members, n]`` float32 slab, group-major. ``VecTrainer`` never goes
through these lists -- its gradients already live in such a slab (``vec.fed_mean``) -- so this class is the drop-in for a
trainer written against the reference's object API. There is no CPU path: without the HIP library every call raises."""
import logging

import numpy as np
import torch

from ._hip import call, ptr, stream_handle

logger = logging.getLogger(__name__)


class Server:
    def __init__(self, name, debug_enabled, device=None):
        logger.info(f"Launching FRL Server: {name}")
        self.name = name
        self.debug = debug_enabled
        self.device = torch.device(device if device is not None else "cuda")

    # -- list-of-lists <-> slab ---------------------------------------------------------------------------------
    def _flatten(self, system_params):
        n_out = len(system_params)
        if n_out == 0:
            return None, [], 0, 0
        n_in = len(system_params[0])
        shapes = [np.shape(layer) for layer in system_params[0][0]]
        sizes = [int(np.prod(s)) for s in shapes]
        n = (sum(sizes) + 3) // 4 * 4  # the kernels move 16-byte pieces: rows padded with zeros to a multiple of 4 floats
        rows = np.zeros((n_out * n_in, n), dtype=np.float32)
        for g, system in enumerate(system_params):
            if len(system) != n_in:
                raise ValueError(f"system {g} has {len(system)} members, system 0 has {n_in}")
            for i, member in enumerate(system):
                if len(member) != len(shapes):
                    raise ValueError(f"system {g} member {i} has {len(member)} layers, expected {len(shapes)}")
                o = 0
                for layer, shape, size in zip(member, shapes, sizes):
                    a = np.asarray(layer, dtype=np.float32)
                    if a.shape != tuple(shape):
                        raise ValueError(f"system {g} member {i}: layer shape {a.shape}, expected {tuple(shape)}")
                    rows[g * n_in + i, o:o + size] = a.reshape(-1)
                    o += size
        return torch.from_numpy(rows).to(self.device), shapes, n_out, n_in

    @staticmethod
    def _unflatten(out, shapes):
        res = []
        for row in out.cpu().numpy():
            layers, o = [], 0
            for shape in shapes:
                size = int(np.prod(shape))
                layers.append(row[o:o + size].reshape(shape).copy())
                o += size
            res.append(layers)
        return res

    def _reduce(self, system_params, weight_sums):
        slab, shapes, n_out, n_in = self._flatten(system_params)
        if n_out == 0:
            return []
        n = slab.shape[1]
        out = torch.empty(n_out, n, dtype=torch.float32, device=self.device)
        # rows are group-major: member i of system g is row g * n_in + i
        call("avd_fed_sum_f32", n_out, n_in, n_in, 1, n, ptr(slab), None, ptr(out), None, stream_handle())
        wsum = None
        if weight_sums is not None:
            ws = np.asarray([float(w) for w in weight_sums], dtype=np.float32)
            if ws.shape != (n_out,):
                raise ValueError(f"{len(ws)} weight sums for {n_out} systems")
            wsum = torch.from_numpy(ws).to(self.device)
        call("avd_fed_finalize_f32", n_out, n, ptr(out), float(n_in), ptr(wsum), stream_handle())
        res = self._unflatten(out, shapes)
        if self.debug:
            logger.info(f"System params after averaging: {res}")
        return res

    # -- reference API ---------------------------------------------------------------------------------------------
    def get_avg_params(self, system_params: list):
        """Mean over the members of every system, layer by layer (src/server/federated.py:47-63)."""
        return self._reduce(system_params, None)

    def get_weighted_avg_params(self, system_params: list, weight_sums):
        """``(1 / weight_sums[g]) * sum`` over the (pre-weighted) members of system g (src/server/federated.py:99-118)."""
        return self._reduce(system_params, weight_sums)
