"""Parameter slab <-> Keras-ordered weight lists, and the reference initialisers.

Slab layout comes from the C ABI (``avd_mlp_layout_init``) and uses PADDED widths (H1, Ha multiples of 16, H2 a
multiple of 32); the Keras-ordered lists carry the LOGICAL shapes of reference ``agent/model.py``
(``model.weights`` / ``model.trainable_variables`` of the functional models: actor 14 / 10 tensors, critic
20 / 14 tensors). Padding is exact: padded units have zero weights and bias, beta = mean = 0 (gamma = var = 1),
so they output 0, receive zero gradients and stay zero under Adam."""
from collections import namedtuple

import numpy as np

Dims = namedtuple("Dims", "S A H1 H2 Ha")  # logical widths


def round_up(x, m):
    return (x + m - 1) // m * m


def padded_widths(H1, H2, Ha):
    return round_up(H1, 16), round_up(H2, 32), round_up(Ha, 16)


def logical_dims(lay, dims=None):
    return Dims(lay.S, lay.A, lay.H1, lay.H2, lay.Ha) if dims is None else Dims(*dims)


# (name in layout, kind, shape fn(d)) in Keras ``.weights`` order; kind 't' = theta (trainable), 's' = stats
ACTOR_WEIGHTS = [("aW1", "t", lambda d: (d.S, d.H1)), ("ab1", "t", lambda d: (d.H1,)), ("ag1", "t", lambda d: (d.H1,)),
                 ("abe1", "t", lambda d: (d.H1,)), ("amm1", "s", lambda d: (d.H1,)), ("amv1", "s", lambda d: (d.H1,)),
                 ("aW2", "t", lambda d: (d.H1, d.H2)), ("ab2", "t", lambda d: (d.H2,)),
                 ("ag2", "t", lambda d: (d.H2,)), ("abe2", "t", lambda d: (d.H2,)),
                 ("amm2", "s", lambda d: (d.H2,)), ("amv2", "s", lambda d: (d.H2,)),
                 ("aW3", "t", lambda d: (d.H2, d.A)), ("ab3", "t", lambda d: (d.A,))]
CRITIC_WEIGHTS = [("cWs", "t", lambda d: (d.S, d.H1)), ("cbs", "t", lambda d: (d.H1,)),
                  ("cWa", "t", lambda d: (d.A, d.Ha)), ("cba", "t", lambda d: (d.Ha,)),
                  ("cgs", "t", lambda d: (d.H1,)), ("cbes", "t", lambda d: (d.H1,)),
                  ("cmms", "s", lambda d: (d.H1,)), ("cmvs", "s", lambda d: (d.H1,)),
                  ("cga", "t", lambda d: (d.Ha,)), ("cbea", "t", lambda d: (d.Ha,)),
                  ("cmma", "s", lambda d: (d.Ha,)), ("cmva", "s", lambda d: (d.Ha,)),
                  ("cW2", "t", lambda d: (d.H1 + d.Ha, d.H2)), ("cb2", "t", lambda d: (d.H2,)),
                  ("cg3", "t", lambda d: (d.H2,)), ("cbe3", "t", lambda d: (d.H2,)),
                  ("cmm3", "s", lambda d: (d.H2,)), ("cmv3", "s", lambda d: (d.H2,)),
                  ("cW3", "t", lambda d: (d.H2, d.A)), ("cb3", "t", lambda d: (d.A,))]
_ONES = {"ag1", "ag2", "amv1", "amv2", "cgs", "cga", "cg3", "cmvs", "cmva", "cmv3"}  # padded entries of these are 1


def _offset(lay, name, kind):
    off = getattr(lay, name)
    if kind == "t" and name.startswith("c"):
        off += lay.actor_size  # critic offsets are relative to the critic block
    return off


def _embed(name, w, d, p):
    """logical tensor -> padded tensor (d: logical Dims, p: padded Dims)."""
    fn = next(f for n, _, f in ACTOR_WEIGHTS + CRITIC_WEIGHTS if n == name)
    out = np.full(fn(p), 1.0 if name in _ONES else 0.0, dtype=np.float32)
    if name == "cW2":  # concat rows: state units then action units, each padded separately
        out[:] = 0.0
        out[:d.H1, :d.H2] = w[:d.H1]
        out[p.H1:p.H1 + d.Ha, :d.H2] = w[d.H1:]
    elif w.ndim == 2:
        out[:] = 0.0
        out[:w.shape[0], :w.shape[1]] = w
    else:
        out[:w.shape[0]] = w
    return out


def _extract(name, wp, d, p):
    """padded tensor -> logical tensor."""
    fn = next(f for n, _, f in ACTOR_WEIGHTS + CRITIC_WEIGHTS if n == name)
    shape = fn(d)
    if name == "cW2":
        return np.concatenate([wp[:d.H1, :d.H2], wp[p.H1:p.H1 + d.Ha, :d.H2]], axis=0)
    if len(shape) == 2:
        return np.array(wp[:shape[0], :shape[1]])
    return np.array(wp[:shape[0]])


def unpack(lay, theta, stats, which, trainable_only=False, dims=None):
    """theta[theta_size], stats[stats_size] (numpy) -> list of arrays in Keras order, logical shapes."""
    d, p = logical_dims(lay, dims), logical_dims(lay)
    spec = ACTOR_WEIGHTS if which == "actor" else CRITIC_WEIGHTS
    out = []
    for name, kind, shp in spec:
        if trainable_only and kind == "s":
            continue
        pshape = shp(p)
        n = int(np.prod(pshape))
        src = theta if kind == "t" else stats
        off = _offset(lay, name, kind)
        out.append(_extract(name, np.asarray(src[off:off + n]).reshape(pshape), d, p))
    return out


def pack(lay, weights, theta, stats, which, trainable_only=False, dims=None):
    """Inverse of unpack: writes the list into theta/stats in place (padding gets its neutral values)."""
    d, p = logical_dims(lay, dims), logical_dims(lay)
    spec = ACTOR_WEIGHTS if which == "actor" else CRITIC_WEIGHTS
    spec = [s for s in spec if not (trainable_only and s[1] == "s")]
    if len(weights) != len(spec):
        raise ValueError(f"{which}: expected {len(spec)} tensors, got {len(weights)}")
    for (name, kind, shp), w in zip(spec, weights):
        shape = shp(d)
        w = np.asarray(w, dtype=np.float32)
        if tuple(w.shape) != tuple(shape):
            raise ValueError(f"{which}.{name}: shape {w.shape} != {shape}")
        wp = _embed(name, w, d, p)
        dst = theta if kind == "t" else stats
        off = _offset(lay, name, kind)
        dst[off:off + wp.size] = wp.reshape(-1)


def init_weights(lay, rs, nominal=None, dims=None):
    """Fresh (theta, stats) float32 numpy slabs with the reference initialisers
    (agent/model.py:17-24, 53-60): U(+-1/sqrt(nominal layer size)) -- the layer's OWN nominal
    width, not fan-in; the critic action layer shares the layer-2 bound; last layers U(+-0.003)
    / U(+-0.0003); biases 0; BatchNormalization gamma=1, beta=0, mean=0, var=1."""
    d = logical_dims(lay, dims)
    H1n, H2n = nominal or (d.H1, d.H2)
    b1, b2 = 1 / np.sqrt(H1n), 1 / np.sqrt(H2n)
    theta = np.zeros(lay.theta_size, dtype=np.float32)
    stats = np.zeros(lay.stats_size, dtype=np.float32)
    actor = [rs.uniform(-b1, b1, (d.S, d.H1)), np.zeros(d.H1), np.ones(d.H1), np.zeros(d.H1), np.zeros(d.H1),
             np.ones(d.H1), rs.uniform(-b2, b2, (d.H1, d.H2)), np.zeros(d.H2), np.ones(d.H2), np.zeros(d.H2),
             np.zeros(d.H2), np.ones(d.H2), rs.uniform(-0.003, 0.003, (d.H2, d.A)), np.zeros(d.A)]
    critic = [rs.uniform(-b1, b1, (d.S, d.H1)), np.zeros(d.H1), rs.uniform(-b2, b2, (d.A, d.Ha)), np.zeros(d.Ha),
              np.ones(d.H1), np.zeros(d.H1), np.zeros(d.H1), np.ones(d.H1),
              np.ones(d.Ha), np.zeros(d.Ha), np.zeros(d.Ha), np.ones(d.Ha),
              rs.uniform(-b2, b2, (d.H1 + d.Ha, d.H2)), np.zeros(d.H2), np.ones(d.H2), np.zeros(d.H2),
              np.zeros(d.H2), np.ones(d.H2), rs.uniform(-0.0003, 0.0003, (d.H2, d.A)), np.zeros(d.A)]
    pack(lay, actor, theta, stats, "actor", dims=dims)
    pack(lay, critic, theta, stats, "critic", dims=dims)
    return theta, stats
