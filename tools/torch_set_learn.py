"""TOOL / TEST INFRASTRUCTURE: Trainer.learn + federated mean for SHARED weight sets as plain PyTorch float32 (or float64) autograd
on the GPU, reading and writing the package's slabs in place -- the "plain PyTorch fp32 reference of the same op" for the
floating-point set learners (csrc/wide.hip, fset.hip, fsplit.hip) at any width, incl. hidden 1024 where no exact-f32 HIP engine
exists. It restates workers/trainer.py:472-508 (losses, no done mask, both gradients at the pre-update weights), agent/model.py:4-85
(Dense -> ReLU -> BatchNormalization in inference form with eps 1e-3, tanh * high, the critic's concatenation) for one weight set
per vehicle index whose gradient is the mean over the set's P x B rows (src/server/federated.py:47-63). Nothing under avddpg_amd/
imports it; tools/train_curves.py swaps it in for VecTrainer's set learner to draw the f32 curve beside the bf16 one.
"""
import math

import torch

BN_EPS = 1e-3


def _views(lay, th, st, k, critic):
    """Keras-shaped views of set k's parameters inside the flat slabs (padded widths == logical widths is assumed)."""
    g = lambda off, *shape: th[k, off:off + math.prod(shape)].view(*shape)
    s = lambda off, n: st[k, off:off + n]
    S, A, H1, H2, Ha = lay.S, lay.A, lay.H1, lay.H2, lay.Ha
    if not critic:
        return dict(W1=g(lay.aW1, S, H1), b1=g(lay.ab1, H1), g1=g(lay.ag1, H1), be1=g(lay.abe1, H1), mm1=s(lay.amm1, H1), mv1=s(lay.amv1, H1),
                    W2=g(lay.aW2, H1, H2), b2=g(lay.ab2, H2), g2=g(lay.ag2, H2), be2=g(lay.abe2, H2), mm2=s(lay.amm2, H2), mv2=s(lay.amv2, H2),
                    W3=g(lay.aW3, H2, A), b3=g(lay.ab3, A))
    o = lay.actor_size
    return dict(Ws=g(o + lay.cWs, S, H1), bs=g(o + lay.cbs, H1), Wa=g(o + lay.cWa, A, Ha), ba=g(o + lay.cba, Ha),
                gs=g(o + lay.cgs, H1), bes=g(o + lay.cbes, H1), mms=s(lay.cmms, H1), mvs=s(lay.cmvs, H1),
                ga=g(o + lay.cga, Ha), bea=g(o + lay.cbea, Ha), mma=s(lay.cmma, Ha), mva=s(lay.cmva, Ha),
                W2=g(o + lay.cW2, H1 + Ha, H2), b2=g(o + lay.cb2, H2), g3=g(o + lay.cg3, H2), be3=g(o + lay.cbe3, H2),
                mm3=s(lay.cmm3, H2), mv3=s(lay.cmv3, H2), W3=g(o + lay.cW3, H2, A), b3=g(o + lay.cb3, A))


def _bn(p, g, be, mm, mv):
    inv = torch.rsqrt(mv + BN_EPS) * g
    return p * inv + (be - mm * inv)


def actor_forward(w, s, high):
    y1 = _bn(torch.relu(s @ w["W1"] + w["b1"]), w["g1"], w["be1"], w["mm1"], w["mv1"])
    y2 = _bn(torch.relu(y1 @ w["W2"] + w["b2"]), w["g2"], w["be2"], w["mm2"], w["mv2"])
    return torch.tanh(y2 @ w["W3"] + w["b3"]) * high


def critic_forward(w, s, a):
    ys = _bn(torch.relu(s @ w["Ws"] + w["bs"]), w["gs"], w["bes"], w["mms"], w["mvs"])
    ya = _bn(torch.relu(a @ w["Wa"] + w["ba"]), w["ga"], w["bea"], w["mma"], w["mva"])
    y2 = _bn(torch.relu(torch.cat([ys, ya], dim=1) @ w["W2"] + w["b2"]), w["g3"], w["be3"], w["mm3"], w["mv3"])
    return y2 @ w["W3"] + w["b3"]


def learn_sets(agents, s, a, r, s2, n_agents, grads=None, losses=None, dtype=torch.float32):
    """agents: vec.AgentGroup with n_sets shared sets; batches AGENT-major as sampled (agent v uses set v % n_sets):
    s, s2 [n_agents, B, S], a [n_agents, B(, 1)], r [n_agents, B]. Returns the mean gradient per set [n_sets, theta_size] float32
    in the slab layout (padding stays 0)."""
    lay, M = agents.lay, agents.n_sets
    if (lay.A, lay.S) not in ((1, 4), (1, 3)):
        raise ValueError("torch_set_learn: decentralized agents only (A = 1)")
    if grads is None:
        grads = torch.zeros(M, lay.theta_size, dtype=torch.float32, device=agents.theta.device)
    A = lay.actor_size
    gamma, high = float(agents.config.gamma), float(agents.high)
    th_t, st, st_t = agents.theta_t.to(dtype), agents.stats.to(dtype), agents.stats_t.to(dtype)
    for k in range(M):
        th = agents.theta.to(dtype).clone().requires_grad_(True)
        sel = slice(k, n_agents, M)
        S = lay.S
        x, x2 = s[sel].reshape(-1, s.shape[-1])[:, :S].to(dtype), s2[sel].reshape(-1, s2.shape[-1])[:, :S].to(dtype)
        act, rew = a[sel].reshape(-1, 1).to(dtype), r[sel].reshape(-1, 1).to(dtype)
        actor, critic = _views(lay, th, st, k, False), _views(lay, th, st, k, True)
        with torch.no_grad():
            y = rew + gamma * critic_forward(_views(lay, th_t, st_t, k, True), x2, actor_forward(_views(lay, th_t, st_t, k, False), x2, high))
        critic_loss = torch.mean((y - critic_forward(critic, x, act)) ** 2)       # workers/trainer.py:495-496
        actor_loss = -torch.mean(critic_forward(critic, x, actor_forward(actor, x, high)))  # :502-504
        gc = torch.autograd.grad(critic_loss, th, retain_graph=True)[0]
        ga = torch.autograd.grad(actor_loss, th)[0]
        grads[k, A:] = gc[k, A:].float()  # critic_grad: d critic_loss / d critic variables (:498)
        grads[k, :A] = ga[k, :A].float()  # actor_grad: d actor_loss / d actor variables (:506)
        if losses is not None:
            losses[k, 0], losses[k, 1] = critic_loss.detach().float(), actor_loss.detach().float()
    return grads


def act_sets(agents, x, n_agents, out, dtype=torch.float32):
    """actor(state) for n_agents agents sharing the group's sets (agent v uses set v % n_sets): x [n_agents, >= S] -> out [n_agents]
    (agent/model.py:26-36), float32 PyTorch."""
    lay, M = agents.lay, agents.n_sets
    with torch.no_grad():
        th, st = agents.theta.to(dtype), agents.stats.to(dtype)
        for k in range(M):
            out.view(-1)[k:n_agents:M] = actor_forward(_views(lay, th, st, k, False), x[k:n_agents:M, :lay.S].to(dtype), float(agents.high)).view(-1).float()
    return out
