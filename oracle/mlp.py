"""Oracle: actor/critic MLP, Trainer.learn, Adam, Polyak (TEST INFRASTRUCTURE).

**Parity unpinned** (see oracle/__init__.py): the arithmetic of Dense /
BatchNormalization / tanh / GradientTape / Adam lives in third-party
``tensorflow==2.4.1`` (reference requirements.txt:2), absent here.  This file
restates the reference call sites:

* architecture + initialiser bounds  -- agent/model.py:4-38 (actor), :41-85 (critic)
* losses / gradient targets           -- workers/trainer.py:472-508
* optimiser                            -- workers/trainer.py:138-139, 348-349
  (tf.keras.optimizers.Adam defaults; TF 2.4.1 ``ApplyAdam`` functor,
  tensorflow/core/kernels/training_ops.cc:
  ``alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g^2-v)(1-b2);
  var -= m*alpha/(sqrt(v)+eps)``, eps = 1e-7)
* soft target update                   -- agent/ddpgagent.py:31-55
* policy (noise add + clip)            -- agent/ddpgagent.py:6-29

Weights are lists of arrays in Keras ``model.weights`` order.  BatchNormalization
layers are ALWAYS in inference mode (models are never called with training=True,
workers/trainer.py:289, 493-503): ``y = x*inv + (beta - mean*inv)``,
``inv = rsqrt(var + 1e-3) * gamma`` (tf.nn.batch_normalization); gamma/beta are
trainable, moving stats are not.
"""
import numpy as np

BN_EPS = 1e-3  # Keras BatchNormalization default epsilon
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7  # tf.keras.optimizers.Adam defaults

# index maps into the Keras ``.weights`` lists ---------------------------------
# actor (14): W1 b1 g1 be1 mm1 mv1 W2 b2 g2 be2 mm2 mv2 W3 b3
ACTOR_TRAINABLE = [0, 1, 2, 3, 6, 7, 8, 9, 12, 13]
# critic (20), functional-model layer order (depth, then creation order):
# Ws bs Wa ba gs bes mms mvs ga bea mma mva W2 b2 g3 be3 mm3 mv3 W3 b3
CRITIC_TRAINABLE = [0, 1, 2, 3, 4, 5, 8, 9, 12, 13, 14, 15, 18, 19]


def init_actor(rs, S, A, H1=256, H2=128, hidd_mult=1, dtype=np.float32):
    """agent/model.py:17-24: U(+-1/sqrt(layer_size)) (the layer's OWN nominal width,
    not fan-in), last layer U(+-0.003); biases 0; BN gamma=1 beta=0 mean=0 var=1."""
    h1, h2 = int(H1 * hidd_mult), int(H2 * hidd_mult)
    b1, b2 = 1 / np.sqrt(H1), 1 / np.sqrt(H2)
    w = [rs.uniform(-b1, b1, (S, h1)), np.zeros(h1), np.ones(h1), np.zeros(h1), np.zeros(h1), np.ones(h1),
         rs.uniform(-b2, b2, (h1, h2)), np.zeros(h2), np.ones(h2), np.zeros(h2), np.zeros(h2), np.ones(h2),
         rs.uniform(-0.003, 0.003, (h2, A)), np.zeros(A)]
    return [x.astype(dtype) for x in w]


def init_critic(rs, S, A, H1=256, H2=128, Ha=48, hidd_mult=1, dtype=np.float32):
    """agent/model.py:53-80: state layer U(+-1/sqrt(H1)); action layer and second
    layer both use layer2_init U(+-1/sqrt(H2)) (:70, :76); output U(+-0.0003)."""
    h1, h2, ha = int(H1 * hidd_mult), int(H2 * hidd_mult), int(Ha * hidd_mult)
    b1, b2 = 1 / np.sqrt(H1), 1 / np.sqrt(H2)
    w = [rs.uniform(-b1, b1, (S, h1)), np.zeros(h1), rs.uniform(-b2, b2, (A, ha)), np.zeros(ha),
         np.ones(h1), np.zeros(h1), np.zeros(h1), np.ones(h1),
         np.ones(ha), np.zeros(ha), np.zeros(ha), np.ones(ha),
         rs.uniform(-b2, b2, (h1 + ha, h2)), np.zeros(h2), np.ones(h2), np.zeros(h2), np.zeros(h2), np.ones(h2),
         rs.uniform(-0.0003, 0.0003, (h2, A)), np.zeros(A)]
    return [x.astype(dtype) for x in w]


def _bn_coeffs(g, be, mm, mv):
    dt = g.dtype.type
    inv = (dt(1) / np.sqrt(mv + dt(BN_EPS))) * g
    return inv, be - mm * inv


def actor_forward(w, s, high, cache=False):
    """agent/model.py:26-36."""
    W1, b1, g1, be1, mm1, mv1, W2, b2, g2, be2, mm2, mv2, W3, b3 = w
    dt = W1.dtype.type
    s = np.asarray(s, dtype=W1.dtype)
    p1 = np.maximum(s @ W1 + b1, 0)
    i1, sh1 = _bn_coeffs(g1, be1, mm1, mv1)
    y1 = p1 * i1 + sh1
    p2 = np.maximum(y1 @ W2 + b2, 0)
    i2, sh2 = _bn_coeffs(g2, be2, mm2, mv2)
    y2 = p2 * i2 + sh2
    t = np.tanh(y2 @ W3 + b3)
    out = t * dt(high)
    if cache:
        return out, (s, p1, y1, p2, y2, t)
    return out


def critic_forward(w, s, a, cache=False):
    """agent/model.py:63-83."""
    Ws, bs, Wa, ba, gs, bes, mms, mvs, ga, bea, mma, mva, W2, b2, g3, be3, mm3, mv3, W3, b3 = w
    s = np.asarray(s, dtype=Ws.dtype)
    a = np.asarray(a, dtype=Ws.dtype)
    ps = np.maximum(s @ Ws + bs, 0)
    is_, shs = _bn_coeffs(gs, bes, mms, mvs)
    ys = ps * is_ + shs
    pa = np.maximum(a @ Wa + ba, 0)
    ia, sha = _bn_coeffs(ga, bea, mma, mva)
    ya = pa * ia + sha
    c = np.concatenate([ys, ya], axis=1)
    p2 = np.maximum(c @ W2 + b2, 0)
    i3, sh3 = _bn_coeffs(g3, be3, mm3, mv3)
    y2 = p2 * i3 + sh3
    q = y2 @ W3 + b3
    if cache:
        return q, (s, a, ps, pa, c, p2, y2)
    return q


def _bn_backward(dy, p, g, mm, mv):
    """dy wrt BN output -> (dgamma, dbeta, dz) where z is the pre-relu activation."""
    dt = g.dtype.type
    rs = dt(1) / np.sqrt(mv + dt(BN_EPS))
    dg = (dy * (p - mm) * rs).sum(axis=0)
    dbe = dy.sum(axis=0)
    dz = dy * (rs * g) * (p > 0)
    return dg, dbe, dz


def critic_backward(w, cache, dq, need_params=True):
    """Returns (grads in CRITIC_TRAINABLE order or None, d/da)."""
    Ws, bs, Wa, ba, gs, bes, mms, mvs, ga, bea, mma, mva, W2, b2, g3, be3, mm3, mv3, W3, b3 = w
    s, a, ps, pa, c, p2, y2 = cache
    h1 = Ws.shape[1]
    dW3 = y2.T @ dq
    db3 = dq.sum(axis=0)
    dy2 = dq @ W3.T
    dg3, dbe3, dz2 = _bn_backward(dy2, p2, g3, mm3, mv3)
    dW2 = c.T @ dz2
    db2 = dz2.sum(axis=0)
    dc = dz2 @ W2.T
    dga, dbea, dza = _bn_backward(dc[:, h1:], pa, ga, mma, mva)
    dWa = a.T @ dza
    dba = dza.sum(axis=0)
    da = dza @ Wa.T
    if not need_params:
        return None, da
    dgs, dbes, dzs = _bn_backward(dc[:, :h1], ps, gs, mms, mvs)
    dWs = s.T @ dzs
    dbs = dzs.sum(axis=0)
    return [dWs, dbs, dWa, dba, dgs, dbes, dga, dbea, dW2, db2, dg3, dbe3, dW3, db3], da


def actor_backward(w, cache, dout, high):
    """Returns grads in ACTOR_TRAINABLE order."""
    W1, b1, g1, be1, mm1, mv1, W2, b2, g2, be2, mm2, mv2, W3, b3 = w
    dt = W1.dtype.type
    s, p1, y1, p2, y2, t = cache
    dz3 = dout * dt(high) * (dt(1) - t * t)
    dW3 = y2.T @ dz3
    db3 = dz3.sum(axis=0)
    dy2 = dz3 @ W3.T
    dg2, dbe2, dz2 = _bn_backward(dy2, p2, g2, mm2, mv2)
    dW2 = y1.T @ dz2
    db2 = dz2.sum(axis=0)
    dy1 = dz2 @ W2.T
    dg1, dbe1, dz1 = _bn_backward(dy1, p1, g1, mm1, mv1)
    dW1 = s.T @ dz1
    db1 = dz1.sum(axis=0)
    return [dW1, db1, dg1, dbe1, dW2, db2, dg2, dbe2, dW3, db3]


def learn(batch, actor, critic, t_actor, t_critic, gamma=0.99, high=2.5):
    """workers/trainer.py:472-508. batch = (s[B,S], a[B,A], r[B,1], s2[B,S]).
    Returns (critic_grad[14], actor_grad[10], aux) with both gradients taken at
    the PRE-update weights; no done mask in the TD target (:494); the critic L2
    regularisers never enter the loss (:496)."""
    s, a, r, s2 = batch
    dtype = actor[0].dtype
    dt = dtype.type
    s, a, s2 = (np.asarray(v, dtype=dtype) for v in (s, a, s2))
    r = np.asarray(r, dtype=dtype).reshape(len(s), -1)
    ta = actor_forward(t_actor, s2, high)  # :493
    y = r + dt(gamma) * critic_forward(t_critic, s2, ta)  # :494
    q, cc = critic_forward(critic, s, a, cache=True)  # :495
    n = dt(q.size)
    critic_loss = np.mean(np.square(y - q))  # :496
    dq = (dt(2) * (q - y) / n).astype(dtype)
    critic_grad, _ = critic_backward(critic, cc, dq)  # :498
    a1, ac = actor_forward(actor, s, high, cache=True)  # :502
    q1, cc1 = critic_forward(critic, s, a1, cache=True)  # :503
    actor_loss = -np.mean(q1)  # :504
    dq1 = np.full_like(q1, dt(-1) / dt(q1.size))
    _, da = critic_backward(critic, cc1, dq1, need_params=False)
    actor_grad = actor_backward(actor, ac, da, high)  # :506
    return critic_grad, actor_grad, dict(critic_loss=critic_loss, actor_loss=actor_loss, y=y, q=q, q1=q1, a1=a1)


class RefAdam:
    """tf.keras.optimizers.Adam(lr) as used at workers/trainer.py:138-139, 348-349."""

    def __init__(self, lr, n_vars=None):
        self.lr, self.t, self.m, self.v = lr, 0, None, None

    def apply_gradients(self, grads, variables):
        """Updates ``variables`` (list of arrays) IN PLACE."""
        dt = variables[0].dtype.type
        if self.m is None:
            self.m = [np.zeros_like(x) for x in variables]
            self.v = [np.zeros_like(x) for x in variables]
        self.t += 1
        alpha = adam_alpha(self.lr, self.t, variables[0].dtype)
        for g, x, m, v in zip(grads, variables, self.m, self.v):
            adam_update(x, m, v, np.asarray(g, dtype=x.dtype), alpha, dt)


def adam_alpha(lr, t, dtype=np.float32):
    dt = np.dtype(dtype).type
    b1p = dt(np.power(dt(ADAM_B1), dt(t)))
    b2p = dt(np.power(dt(ADAM_B2), dt(t)))
    return dt(dt(lr) * np.sqrt(dt(1) - b2p) / (dt(1) - b1p))


def adam_update(x, m, v, g, alpha, dt=np.float32):
    """In-place ApplyAdam (non-nesterov) in the arrays' dtype."""
    m += (g - m) * (dt(1) - dt(ADAM_B1))
    v += (g * g - v) * (dt(1) - dt(ADAM_B2))
    x -= (m * alpha) / (np.sqrt(v) + dt(ADAM_EPS))


def update_target(tau, t_critic_w, critic_w, t_actor_w, actor_w):
    """agent/ddpgagent.py:31-55 -- over ALL weights incl. BN moving stats; tau and
    (1-tau) are Python doubles rounded to the variable dtype when multiplied."""
    def mix(ws, ts):
        out = []
        for w, t in zip(ws, ts):
            d = w.dtype.type
            out.append(w * d(tau) + t * d(1 - tau))
        return out
    return mix(critic_w, t_critic_w), mix(actor_w, t_actor_w)


def policy(actor_out, noise=None, lbound=None, hbound=None):
    """agent/ddpgagent.py:6-29: squeeze, add (float64) noise, clip."""
    sampled = np.squeeze(actor_out)
    if noise is not None:
        sampled = sampled + noise
    return [np.squeeze(np.clip(sampled, lbound, hbound))]
