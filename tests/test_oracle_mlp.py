"""Cross-check the (TF-unpinned) NN half of the oracle against an independent implementation:
torch-CPU float64 autograd, finite differences, and torch.optim-free hand Adam. CPU only."""
import numpy as np
import pytest
import torch

from oracle import mlp


def _torch_actor(w, s, high):
    W1, b1, g1, be1, mm1, mv1, W2, b2, g2, be2, mm2, mv2, W3, b3 = w
    h = torch.relu(s @ W1 + b1)
    h = torch.nn.functional.batch_norm(h, mm1, mv1, g1, be1, training=False, eps=1e-3)
    h = torch.relu(h @ W2 + b2)
    h = torch.nn.functional.batch_norm(h, mm2, mv2, g2, be2, training=False, eps=1e-3)
    return torch.tanh(h @ W3 + b3) * high


def _torch_critic(w, s, a):
    Ws, bs, Wa, ba, gs, bes, mms, mvs, ga, bea, mma, mva, W2, b2, g3, be3, mm3, mv3, W3, b3 = w
    hs = torch.nn.functional.batch_norm(torch.relu(s @ Ws + bs), mms, mvs, gs, bes, training=False, eps=1e-3)
    ha = torch.nn.functional.batch_norm(torch.relu(a @ Wa + ba), mma, mva, ga, bea, training=False, eps=1e-3)
    h = torch.relu(torch.cat([hs, ha], dim=1) @ W2 + b2)
    h = torch.nn.functional.batch_norm(h, mm3, mv3, g3, be3, training=False, eps=1e-3)
    return h @ W3 + b3


def _rand_nets(seed, S=4, A=1, H1=32, H2=16, Ha=16, dtype=np.float64, perturb=True):
    rs = np.random.RandomState(seed)
    nets = [mlp.init_actor(rs, S, A, H1, H2, dtype=dtype), mlp.init_critic(rs, S, A, H1, H2, Ha, dtype=dtype),
            mlp.init_actor(rs, S, A, H1, H2, dtype=dtype), mlp.init_critic(rs, S, A, H1, H2, Ha, dtype=dtype)]
    if perturb:  # move gamma/beta/bias/moving stats off their defaults so every term is exercised
        for net in nets:
            for i, x in enumerate(net):
                if x.ndim == 1:
                    x += rs.uniform(-0.3, 0.3, x.shape).astype(dtype)
            for i in ([5, 11] if len(net) == 14 else [7, 11, 17]):
                net[i][:] = np.abs(net[i]) + 0.5  # variances positive
        nets[0][12] *= 50  # make tanh non-trivial
        nets[2][12] *= 50
    return nets


def _batch(seed, B=64, S=4, A=1):
    rs = np.random.RandomState(seed)
    return rs.normal(size=(B, S)), rs.uniform(-2.5, 2.5, (B, A)), rs.normal(size=(B, 1)), rs.normal(size=(B, S))


@pytest.mark.parametrize("seed", [0, 1])
def test_learn_matches_torch_autograd_f64(seed):
    actor, critic, t_actor, t_critic = _rand_nets(seed)
    s, a, r, s2 = _batch(seed)
    cg, ag, aux = mlp.learn((s, a, r, s2), actor, critic, t_actor, t_critic, 0.99, 2.5)
    def T(ws, trainable):
        return [torch.tensor(x, dtype=torch.float64, requires_grad=(i in trainable)) for i, x in enumerate(ws)]
    ta, tc = T(actor, mlp.ACTOR_TRAINABLE), T(critic, mlp.CRITIC_TRAINABLE)
    tta, ttc = T(t_actor, ()), T(t_critic, ())
    ts, ta_, tr, ts2 = (torch.tensor(v) for v in (s, a, r, s2))
    y = tr + 0.99 * _torch_critic(ttc, ts2, _torch_actor(tta, ts2, 2.5))
    lc = torch.mean((y - _torch_critic(tc, ts, ta_)) ** 2)
    gc = torch.autograd.grad(lc, [tc[i] for i in mlp.CRITIC_TRAINABLE])
    la = -torch.mean(_torch_critic(tc, ts, _torch_actor(ta, ts, 2.5)))
    ga = torch.autograd.grad(la, [ta[i] for i in mlp.ACTOR_TRAINABLE])
    assert abs(lc.item() - aux["critic_loss"]) < 1e-12 and abs(la.item() - aux["actor_loss"]) < 1e-12
    for mine, ref in zip(cg, gc):
        assert np.allclose(mine, ref.numpy(), rtol=1e-9, atol=1e-13)
    for mine, ref in zip(ag, ga):
        assert np.allclose(mine, ref.numpy(), rtol=1e-9, atol=1e-13)


def test_learn_finite_differences():
    actor, critic, t_actor, t_critic = _rand_nets(3, H1=8, H2=8, Ha=8)
    batch = _batch(3, B=16)
    cg, ag, aux = mlp.learn(batch, actor, critic, t_actor, t_critic)
    rs = np.random.RandomState(0)
    for which, net, grads, tr_idx, key in (("c", critic, cg, mlp.CRITIC_TRAINABLE, "critic_loss"),
                                           ("a", actor, ag, mlp.ACTOR_TRAINABLE, "actor_loss")):
        for gi, wi in enumerate(tr_idx):
            flat = net[wi].reshape(-1)
            j = rs.randint(flat.size)
            old = flat[j]
            h = 1e-6
            flat[j] = old + h
            lp = mlp.learn(batch, actor, critic, t_actor, t_critic)[2][key]
            flat[j] = old - h
            lm = mlp.learn(batch, actor, critic, t_actor, t_critic)[2][key]
            flat[j] = old
            fd = (lp - lm) / (2 * h)
            assert abs(fd - grads[gi].reshape(-1)[j]) < 1e-6 * max(1.0, abs(fd)), (which, gi)


def test_f32_learn_close_to_f64():
    n64 = _rand_nets(5, H1=256, H2=128, Ha=48)
    n32 = [[x.astype(np.float32) for x in net] for net in n64]
    b = _batch(5)
    c64, a64, _ = mlp.learn(b, *n64)
    c32, a32, _ = mlp.learn(b, *n32)
    for x, y in zip(c64 + a64, c32 + a32):
        assert y.dtype == np.float32
        assert np.max(np.abs(x - y)) <= 2e-5 * max(1e-3, np.max(np.abs(x)))


def test_adam_matches_closed_form_and_torch():
    """TF ApplyAdam == textbook Adam up to eps placement: compare against torch.optim.Adam in the
    regime where eps is negligible, and against the closed form for the first step."""
    rs = np.random.RandomState(0)
    x0 = rs.normal(size=1000)
    grads = [rs.normal(size=1000) for _ in range(5)]
    x = x0.copy()
    opt = mlp.RefAdam(5e-4)
    for g in grads:
        opt.apply_gradients([g], [x])
    tx = torch.tensor(x0.copy(), requires_grad=True)
    topt = torch.optim.Adam([tx], lr=5e-4, betas=(0.9, 0.999), eps=1e-7)
    for g in grads:
        tx.grad = torch.tensor(g)
        topt.step()
    # torch puts eps outside the bias-corrected sqrt(v_hat); TF inside alpha -- differs at O(eps/|g|) relative
    # (eps_eff = eps/sqrt(1-b2^t) <= 3.2e-6), so compare where every |g| is well above that
    ok = np.min(np.abs(np.array(grads)), axis=0) > 0.01
    assert ok.sum() > 900
    assert np.allclose(x[ok], tx.detach().numpy()[ok], rtol=0, atol=5 * 5e-4 * 3.2e-6 / 0.01)
    x1 = x0.copy()
    o = mlp.RefAdam(0.01)
    o.apply_gradients([grads[0]], [x1])
    g = grads[0]
    alpha = 0.01 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.allclose(x1, x0 - alpha * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-7), rtol=1e-12)


def test_update_target_and_policy():
    rs = np.random.RandomState(0)
    w = [rs.normal(size=(4, 3)).astype(np.float32), rs.normal(size=3).astype(np.float32)]
    t = [rs.normal(size=(4, 3)).astype(np.float32), rs.normal(size=3).astype(np.float32)]
    tc, ta = mlp.update_target(0.001, t, w, t, w)
    for n, a, b in zip(tc, w, t):
        assert n.dtype == np.float32
        assert np.array_equal(n, a * np.float32(0.001) + b * np.float32(0.999))
    # frozen BN stats stay exactly (0, 1) under the f32 Polyak update
    one, zero = [np.ones(4, np.float32)], [np.zeros(4, np.float32)]
    assert np.array_equal(mlp.update_target(0.001, one, one, zero, zero)[0][0], one[0])
    out = mlp.policy(np.array([[2.4]], dtype=np.float32), np.array([0.3]), -2.5, 2.5)
    assert isinstance(out, list) and out[0] == 2.5 and out[0].dtype == np.float64
    out = mlp.policy(np.array([[1.0]], dtype=np.float32), None, -2.5, 2.5)
    assert out[0] == 1.0
