"""Oracle: federated averaging (TEST INFRASTRUCTURE; parity unpinned for the TF reductions).

Restates reference ``src/server/federated.py:18-122`` and the weighting of
``workers/trainer.py:385-398``.  ``system_params[idx1][idx2][layer]`` -> mean over
idx2 (interfrl: idx1 = vehicle index, idx2 = platoon).
"""
import numpy as np


def get_avg_params(system_params):
    """federated.py:47-63: per layer, stack over idx2 and reduce_mean(axis=0)."""
    out = []
    for group in system_params:
        n_layers = len(group[0])
        out.append([np.mean(np.stack([member[i] for member in group], axis=0), axis=0, dtype=group[0][i].dtype)
                    for i in range(n_layers)])
    return out


def get_weighted_avg_params(system_params, weight_sums):
    """federated.py:99-118: params arrive PRE-multiplied by their weight
    (trainer.py:372-377); result = (1/sum_w) * reduce_sum(axis=0)."""
    out = []
    for group, ws in zip(system_params, weight_sums):
        n_layers = len(group[0])
        layers = []
        for i in range(n_layers):
            st = np.stack([member[i] for member in group], axis=0)
            layers.append((st.dtype.type(1 / ws) * st.sum(axis=0, dtype=st.dtype)).astype(st.dtype))
        out.append(layers)
    return out


def frl_weight(ep_rewards, window=10):
    """trainer.py:395: |1 / mean(last `window` episodic rewards)|."""
    return abs(1 / np.mean(ep_rewards[-window:]))
