"""GPU parity: actor/critic forward, Trainer.learn gradients, Adam + Polyak, federated mean vs the
oracle (oracle/mlp.py, oracle/federated.py). The NN arithmetic is third-party TensorFlow in the
reference ("parity unpinned"); the oracle restates it and is itself cross-checked against torch
float64 autograd in tests/test_oracle_mlp.py."""
import numpy as np
import pytest
import torch

from avddpg_amd import config, params, vec
from oracle import federated as ofed
from oracle import mlp as omlp
from tests.gpu_util import need_gpu, t

pytestmark = pytest.mark.gpu
# float32 accumulation-order tolerance for network outputs / gradients, relative to the tensor's max
FWD_TOL = 2e-5
GRAD_TOL = 1e-4


def _perturbed_group(n_sets, S=4, seed=0, A=1, hidd_mult=1.0, **confkw):
    """AgentGroup whose every set has different, non-default weights (gamma/beta/bias/moving stats
    off their initial values, tanh unsaturated but non-trivial) so every term is exercised."""
    conf = config.Config(**confkw)
    grp = vec.AgentGroup(n_sets, S, A, conf, seed=seed, hidd_mult=hidd_mult)
    lay, dims = grp.lay, grp.dims
    rs = np.random.RandomState(seed + 100)
    th = np.zeros((n_sets, lay.theta_size), np.float32)
    st = np.zeros((n_sets, lay.stats_size), np.float32)
    tht, stt = th.copy(), st.copy()
    for dst_th, dst_st in ((th, st), (tht, stt)):
        for k in range(n_sets):
            a, s_ = params.init_weights(lay, rs, dims=dims)
            aw = params.unpack(lay, a, s_, "actor", dims=dims)
            cw = params.unpack(lay, a, s_, "critic", dims=dims)
            for net, var_idx in ((aw, (5, 11)), (cw, (7, 11, 17))):
                for i, w in enumerate(net):
                    if w.ndim == 1:
                        w += rs.uniform(-0.3, 0.3, w.shape).astype(np.float32)
                for i in var_idx:
                    net[i][:] = np.abs(net[i]) + 0.5
            aw[12] *= 30
            cw[18] *= 300
            params.pack(lay, aw, dst_th[k], dst_st[k], "actor", dims=dims)
            params.pack(lay, cw, dst_th[k], dst_st[k], "critic", dims=dims)
    grp.theta.copy_(t(th)), grp.stats.copy_(t(st)), grp.theta_t.copy_(t(tht)), grp.stats_t.copy_(t(stt))
    return conf, grp


def _nets(grp, k, dtype=np.float32):
    c = lambda ws: [w.astype(dtype) for w in ws]
    return (c(grp.get_weights(k, "actor")), c(grp.get_weights(k, "critic")),
            c(grp.get_weights(k, "actor", target=True)), c(grp.get_weights(k, "critic", target=True)))


def _relerr(got, ref):
    return np.max(np.abs(got - ref)) / max(1e-12, np.max(np.abs(ref)))


@pytest.mark.parametrize("S", [4, 3])
def test_actor_critic_forward_per_agent_and_shared_sets(S):
    need_gpu()
    n_sets, n_agents = 5, 40
    conf, grp = _perturbed_group(n_sets, S=S, seed=1)
    rs = np.random.RandomState(2)
    x = rs.normal(0, 1.5, size=(n_agents, 4)).astype(np.float32)  # 4-wide rows; Model A reads 3
    act = rs.uniform(-2.5, 2.5, size=n_agents).astype(np.float32)
    out = grp.actor(t(x), set_mod=n_sets).cpu().numpy()
    q = grp.critic(t(x), t(act), set_mod=n_sets).cpu().numpy()
    out_t = grp.actor(t(x), set_mod=n_sets, target=True).cpu().numpy()
    for v in range(n_agents):
        a, c, ta, tc = _nets(grp, v % n_sets, np.float64)
        ref = omlp.actor_forward(a, x[v:v + 1, :S], 2.5)[0, 0]
        assert abs(out[v] - ref) <= FWD_TOL * 2.5, (v, out[v], ref)
        assert abs(out_t[v] - omlp.actor_forward(ta, x[v:v + 1, :S], 2.5)[0, 0]) <= FWD_TOL * 2.5
        refq = omlp.critic_forward(c, x[v:v + 1, :S], act[v:v + 1, None])[0, 0]
        assert abs(q[v] - refq) <= FWD_TOL * max(1.0, abs(refq)), (v, q[v], refq)
    assert np.abs(out).max() <= 2.5 and np.abs(out).max() > 0.05  # tanh*high, non-trivial
    # one set per agent (set_mod = 0): agent v uses set v
    out0 = grp.actor(t(x[:n_sets]), set_mod=0).cpu().numpy()
    assert np.array_equal(out0, out[:n_sets])


@pytest.mark.parametrize("S,set_mod", [(4, 0), (4, 3), (3, 0)])
def test_learn_gradients_match_oracle(S, set_mod):
    """critic_grad and actor_grad of Trainer.learn for several agents, vs the float64 oracle."""
    need_gpu()
    n_agents = 6
    n_sets = set_mod if set_mod else n_agents
    conf, grp = _perturbed_group(n_sets, S=S, seed=3)
    rs = np.random.RandomState(4)
    B = 64
    s = rs.normal(0, 1.5, size=(n_agents, B, S)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(n_agents, B, 1)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n_agents, B))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(n_agents, B, S)).astype(np.float32)
    losses = torch.zeros(n_agents, 2, device="cuda")
    grads = grp.learn(t(s), t(a), t(r), t(s2), set_mod, losses=losses)
    torch.cuda.synchronize()
    assert torch.isfinite(grads).all()
    for v in range(n_agents):
        k = v % set_mod if set_mod else v
        nets64 = _nets(grp, k, np.float64)
        cg, ag, aux = omlp.learn((s[v], a[v], r[v][:, None], s2[v]), *nets64, gamma=conf.gamma, high=2.5)
        cg32, ag32, _ = omlp.learn((s[v], a[v], r[v][:, None], s2[v]), *_nets(grp, k, np.float32), gamma=conf.gamma, high=2.5)
        gcg, gag = grp.grads_as_lists(grads[v])
        assert abs(losses[v, 0].item() - aux["critic_loss"]) <= 1e-4 * max(1.0, abs(aux["critic_loss"]))
        assert abs(losses[v, 1].item() - aux["actor_loss"]) <= 1e-4 * max(1.0, abs(aux["actor_loss"]))
        # tolerance: GRAD_TOL of the tensor's max, or -- for cancellation-prone sums such as
        # dbeta3 = W3 * sum_r dq[r] -- 4x the error a float32 CPU implementation (the f32 oracle) makes
        for i, (got, ref, r32) in enumerate(zip(gcg, cg, cg32)):
            assert got.shape == ref.shape
            assert _relerr(got, ref) <= max(GRAD_TOL, 4 * _relerr(r32, ref)), ("critic", v, i, _relerr(got, ref))
        for i, (got, ref, r32) in enumerate(zip(gag, ag, ag32)):
            assert _relerr(got, ref) <= max(GRAD_TOL, 4 * _relerr(r32, ref)), ("actor", v, i, _relerr(got, ref))
        assert np.max(np.abs(ag[0])) > 1e-8 and np.max(np.abs(cg[0])) > 1e-8  # non-degenerate case
    # alignment padding of the slab stays zero
    lay = grp.lay
    g = grads.cpu().numpy()
    assert np.all(g[:, lay.ab3 + 1:lay.actor_size] == 0) and np.all(g[:, lay.actor_size + lay.cb3 + 1:] == 0)


def test_learn_at_reference_init_matches_f32_oracle():
    """Fresh reference-initialised networks (gamma=1, beta=0, tiny last layers) at config-#2 scale of
    agents per launch (one weight set per agent)."""
    need_gpu()
    n_agents = 512
    conf = config.Config()
    grp = vec.AgentGroup(n_agents, 4, 1, conf, seed=5)
    rs = np.random.RandomState(6)
    s = rs.normal(0, 1.5, size=(n_agents, 64, 4)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(n_agents, 64, 1)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n_agents, 64))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(n_agents, 64, 4)).astype(np.float32)
    grads = grp.learn(t(s), t(a), t(r), t(s2), 0)
    nets = _nets(grp, 0, np.float64)
    for v in (0, 255, 511):
        cg, ag, _ = omlp.learn((s[v], a[v], r[v][:, None], s2[v]), *nets)
        gcg, gag = grp.grads_as_lists(grads[v])
        for got, ref in zip(gcg + gag, cg + ag):
            assert _relerr(got, ref) <= GRAD_TOL


def test_adam_polyak_bit_exact_vs_f32_oracle():
    """K11+K12: three successive updates of 7 weight sets; theta, targets, m, v equal the float32
    oracle (TF ApplyAdam formulation, eps=1e-7; Polyak over all weights incl. BN stats) bit for bit."""
    need_gpu()
    n_sets = 7
    conf, grp = _perturbed_group(n_sets, seed=8)
    lay = grp.lay
    rs = np.random.RandomState(9)
    th = grp.theta.cpu().numpy().copy()
    tht = grp.theta_t.cpu().numpy().copy()
    st, stt = grp.stats.cpu().numpy().copy(), grp.stats_t.cpu().numpy().copy()
    m, v = np.zeros_like(th), np.zeros_like(th)
    f = np.float32
    for step in range(1, 4):
        g = (rs.normal(size=th.shape) * rs.choice([1e-6, 1e-3, 1.0], size=th.shape)).astype(np.float32)
        grp.apply(t(g))
        for k in range(n_sets):
            for lo, hi, lr in ((0, lay.actor_size, conf.actor_lr), (lay.actor_size, lay.theta_size, conf.critic_lr)):
                alpha = omlp.adam_alpha(lr, step)
                omlp.adam_update(th[k, lo:hi], m[k, lo:hi], v[k, lo:hi], g[k, lo:hi], alpha)
        tht = th * f(conf.tau) + tht * f(1 - conf.tau)
        stt = st * f(conf.tau) + stt * f(1 - conf.tau)
        assert np.array_equal(grp.theta.cpu().numpy(), th), step
        assert np.array_equal(grp.m.cpu().numpy(), m) and np.array_equal(grp.v.cpu().numpy(), v)
        assert np.array_equal(grp.theta_t.cpu().numpy(), tht)
        assert np.array_equal(grp.stats_t.cpu().numpy(), stt)
        assert np.array_equal(grp.stats.cpu().numpy(), st)  # online BN stats never move
    assert np.array_equal(grp.step.cpu().numpy(), np.full(n_sets, 3, np.int32))
    # the list-based oracle update_target gives the same target weights
    tc, ta = omlp.update_target(conf.tau, grp.get_weights(0, "critic", target=True), grp.get_weights(0, "critic"),
                                grp.get_weights(0, "actor", target=True), grp.get_weights(0, "actor"))
    before = grp.get_weights(0, "critic", target=True)
    grp.apply(t(np.zeros_like(th)))  # zero grads still decay m and move weights slightly; check Polyak only via API
    from avddpg_amd._hip import call, ptr, stream_handle
    w = t(rs.normal(size=1000)); tt = t(rs.normal(size=1000)); tt0 = tt.cpu().numpy().copy()
    call("avd_polyak_f32", 1000, ptr(w), ptr(tt), 0.001, stream_handle())
    assert np.array_equal(tt.cpu().numpy(), w.cpu().numpy() * f(0.001) + tt0 * f(0.999))
    assert len(tc) == len(before) == 20 and len(ta) == 14


def test_federated_mean_unweighted_and_weighted():
    need_gpu()
    P, M, n = 37, 5, 76488
    rs = np.random.RandomState(10)
    g = rs.normal(size=(P, M, n)).astype(np.float32)
    out = vec.fed_mean(t(g).reshape(P * M, n), P, M).cpu().numpy()
    ref = ofed.get_avg_params([[[g[p, m]] for p in range(P)] for m in range(M)])
    for m in range(M):
        assert np.allclose(out[m], ref[m][0], rtol=1e-5, atol=1e-6)
    w = rs.uniform(0.5, 6.0, size=(P, M)).astype(np.float32)
    outw = vec.fed_mean(t(g).reshape(P * M, n), P, M, weights=t(w)).cpu().numpy()
    refw = ofed.get_weighted_avg_params([[[w[p, m] * g[p, m]] for p in range(P)] for m in range(M)],
                                        [float(w[:, m].sum()) for m in range(M)])
    for m in range(M):
        assert np.allclose(outw[m], refw[m][0], rtol=1e-5, atol=1e-6)
    # the reference's own table (src/server/test_federated.py:26-42): model 1, layer 1
    tab = np.zeros((2, 1, 4), np.float32)
    tab[0, 0, :3], tab[1, 0, :3] = [1, 2, 3], [10, 11, 12]
    ww = np.array([[2.0], [1.0]], np.float32)
    o = vec.fed_mean(t(tab).reshape(2, 4), 2, 1, weights=t(ww)).cpu().numpy()
    assert np.allclose(o[0, :3], (2 * np.array([1, 2, 3.]) + np.array([10, 11, 12.])) / 3)
    # intrafrl: mean over the vehicles of each platoon
    outi = vec.fed_mean(t(g).reshape(P * M, n), P, M, method="intrafrl").cpu().numpy()
    assert np.allclose(outi, g.mean(axis=1), rtol=1e-5, atol=1e-6)
    # scatter each group's average back to its members (directional: skip the lead vehicle)
    dst = torch.zeros(P * M, n, device="cuda")
    vec.fed_scatter(t(out), dst, P, M, "interfrl")
    assert torch.equal(dst.reshape(P, M, n)[11], t(out))
    dst.zero_()
    vec.fed_scatter(t(outi), dst, P, M, "intrafrl", i_begin=1)
    d = dst.reshape(P, M, n).cpu().numpy()
    assert np.all(d[:, 0] == 0) and np.array_equal(d[:, 3], outi)


def test_unsupported_shapes_fail_loudly():
    need_gpu()
    from avddpg_amd._hip import AvdError
    conf = config.Config(actor_layer1_size=1024, actor_layer2_size=1024, critic_layer1_size=1024, critic_layer2_size=1024)
    grp = vec.AgentGroup(1, 4, 1, conf)
    z = torch.zeros(1, 64, 4, device="cuda")
    with pytest.raises(AvdError, match="LDS|H2 a multiple of 32 and <="):
        grp.learn(z, torch.zeros(1, 64, 1, device="cuda"), torch.zeros(1, 64, device="cuda"), z, 0)


def test_generic_and_specialised_learn_kernels_agree(monkeypatch, diag_lib):
    """Reference widths run the dimension-specialised kernel, other widths the generic one: both against the
    oracle, and against each other on the same inputs."""
    need_gpu()
    n_agents = 4
    conf, grp = _perturbed_group(n_agents, S=4, seed=21)
    rs = np.random.RandomState(22)
    s = rs.normal(0, 1.5, size=(n_agents, 64, 4)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(n_agents, 64, 1)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n_agents, 64))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(n_agents, 64, 4)).astype(np.float32)
    # three kernels serve the reference widths: learn_kernel_l (lean.hip, default: two workgroups per CU, first-layer
    # activations recomputed), learn_kernel_t (AVD_LEARN_KERNEL=fast) and the general one (AVD_LEARN_GENERAL=1) -- switches of
    # the diagnostic build only (diag_lib fixture); the shipped library always runs learn_kernel_l here
    lean = grp.learn(t(s), t(a), t(r), t(s2), 0).cpu().numpy()
    monkeypatch.setenv("AVD_LEARN_KERNEL", "fast")
    fast = grp.learn(t(s), t(a), t(r), t(s2), 0).cpu().numpy()
    monkeypatch.delenv("AVD_LEARN_KERNEL")
    monkeypatch.setenv("AVD_LEARN_GENERAL", "1")
    gen = grp.learn(t(s), t(a), t(r), t(s2), 0).cpu().numpy()
    monkeypatch.delenv("AVD_LEARN_GENERAL")
    # summation orders differ; values must agree closely
    assert np.max(np.abs(fast - gen)) <= 2e-5 * np.max(np.abs(gen))
    assert np.max(np.abs(lean - gen)) <= 2e-5 * np.max(np.abs(gen))
    assert np.max(np.abs(lean - fast)) <= 2e-5 * np.max(np.abs(fast))
    # non-reference widths -> generic kernel, checked against the oracle
    conf2, grp2 = _perturbed_group(3, S=4, seed=23, actor_layer1_size=128, actor_layer2_size=64, critic_layer1_size=128,
                                   critic_layer2_size=64, critic_act_layer_size=32)
    g2 = grp2.learn(t(s[:3]), t(a[:3]), t(r[:3]), t(s2[:3]), 0)
    for v in range(3):
        cg, ag, _ = omlp.learn((s[v], a[v], r[v][:, None], s2[v]), *_nets(grp2, v, np.float64))
        cg32, ag32, _ = omlp.learn((s[v], a[v], r[v][:, None], s2[v]), *_nets(grp2, v, np.float32))
        gcg, gag = grp2.grads_as_lists(g2[v])
        for got, ref, r32 in zip(gcg + gag, cg + ag, cg32 + ag32):
            assert _relerr(got, ref) <= max(GRAD_TOL, 4 * _relerr(r32, ref))


def test_full_size_learn_properties_20480_agents():
    """BASELINE configs[1] size (4096 x 5 = 20480 agents, one weight set each): size-independent properties of
    Trainer.learn / the federated sum that need no oracle run at this size --
    lane independence (an agent's gradient depends only on its own batch and weights: duplicated agents give
    bit-identical rows wherever they sit in the grid), determinism (two launches agree bit for bit), spot checks of
    three agents against the float64 oracle, and linearity of the federated sum."""
    need_gpu()
    n = 4096 * 5
    conf = config.Config()
    grp = vec.AgentGroup(n, 4, 1, conf, seed=11)
    g = torch.Generator(device="cuda").manual_seed(5)
    grp.theta.add_(torch.randn(grp.theta.shape, device="cuda", generator=g) * 0.01 * (grp.theta != 0))  # agents differ; padding stays 0
    grp.theta_t.copy_(grp.theta)
    s = torch.randn(n, 64, 4, device="cuda", generator=g) * 1.5
    a = torch.rand(n, 64, 1, device="cuda", generator=g) * 5 - 2.5
    r = -torch.rand(n, 64, device="cuda", generator=g)
    s2 = torch.randn(n, 64, 4, device="cuda", generator=g) * 1.5
    # duplicate agent 7 into far-away lanes (different workgroups / XCDs)
    for dst in (8, 4097, n - 1):
        for x in (grp.theta, grp.theta_t, grp.stats, grp.stats_t, s, a, r, s2):
            x[dst].copy_(x[7])
    g1 = grp.learn(s, a, r, s2, 0)
    g2 = grp.learn(s, a, r, s2, 0)
    assert torch.equal(g1, g2) and torch.isfinite(g1).all()
    for dst in (8, 4097, n - 1):
        assert torch.equal(g1[dst], g1[7])
    assert not torch.equal(g1[7], g1[9])
    for v in (7, 12345, n - 2):
        nets = _nets(grp, v, np.float64)
        cg, ag, _ = omlp.learn((s[v].cpu().numpy(), a[v].cpu().numpy(), r[v].cpu().numpy()[:, None], s2[v].cpu().numpy()), *nets)
        nets32 = _nets(grp, v, np.float32)
        cg32, ag32, _ = omlp.learn((s[v].cpu().numpy(), a[v].cpu().numpy(), r[v].cpu().numpy()[:, None], s2[v].cpu().numpy()), *nets32)
        gcg, gag = grp.grads_as_lists(g1[v])
        for got, ref, r32 in zip(gcg + gag, cg + ag, cg32 + ag32):
            assert _relerr(got, ref) <= max(GRAD_TOL, 4 * _relerr(r32, ref))
    # federated sum is linear: mean(2*g + h) == 2*mean(g) + mean(h)
    h = torch.randn(n, grp.lay.theta_size, device="cuda", generator=g)
    lhs = vec.fed_mean(2 * g1 + h, 4096, 5)
    rhs = 2 * vec.fed_mean(g1, 4096, 5) + vec.fed_mean(h, 4096, 5)
    assert torch.allclose(lhs, rhs, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("L", [3, 5])
def test_centralized_shapes_forward_and_learn(L):
    """Centralized framework (SURVEY f-3; workers/trainer.py:80-87, 108-113): one agent per platoon with
    S = 4L states, A = L actions and widths int(256*1.2)/int(128*1.2)/int(48*1.2) = 307/153/57, held in slabs
    padded to 320/160/64. The critic has A outputs, the TD target broadcasts r over them (trainer.py:494) and
    both losses average over B*A."""
    need_gpu()
    n_agents = 3
    S, A = 4 * L, L  # L = 5: two chunks of output-layer columns (4 + 1) and five chunks of inputs in the general kernel
    conf, grp = _perturbed_group(n_agents, S=S, A=A, hidd_mult=1.2, seed=41)
    assert tuple(grp.dims) == (S, A, 307, 153, 57) and (grp.lay.H1, grp.lay.H2, grp.lay.Ha) == (320, 160, 64)
    rs = np.random.RandomState(42)
    x = rs.normal(0, 1.5, size=(n_agents, S)).astype(np.float32)
    act = rs.uniform(-2.5, 2.5, size=(n_agents, A)).astype(np.float32)
    out = grp.actor(t(x), set_mod=0).cpu().numpy()
    q = grp.critic(t(x), t(act), set_mod=0).cpu().numpy()
    assert out.shape == (n_agents, A) and q.shape == (n_agents, A)
    for v in range(n_agents):
        aw, cw, _, _ = _nets(grp, v, np.float64)
        ref = omlp.actor_forward(aw, x[v:v + 1], 2.5)[0]
        assert np.max(np.abs(out[v] - ref)) <= FWD_TOL * 2.5
        refq = omlp.critic_forward(cw, x[v:v + 1], act[v:v + 1])[0]
        assert np.max(np.abs(q[v] - refq)) <= FWD_TOL * max(1.0, np.max(np.abs(refq)))
    s = rs.normal(0, 1.5, size=(n_agents, 64, S)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(n_agents, 64, A)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n_agents, 64))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(n_agents, 64, S)).astype(np.float32)
    losses = torch.zeros(n_agents, 2, device="cuda")
    g = grp.learn(t(s), t(a), t(r), t(s2), 0, losses=losses)
    gh = g.cpu().numpy()
    for v in range(n_agents):
        cg, ag, aux = omlp.learn((s[v], a[v], r[v][:, None], s2[v]), *_nets(grp, v, np.float64))
        cg32, ag32, _ = omlp.learn((s[v], a[v], r[v][:, None], s2[v]), *_nets(grp, v, np.float32))
        gcg, gag = grp.grads_as_lists(g[v])
        assert [w.shape for w in gcg] == [w.shape for w in cg] and [w.shape for w in gag] == [w.shape for w in ag]
        for got, ref, r32 in zip(gcg + gag, cg + ag, cg32 + ag32):
            assert _relerr(got, ref) <= max(GRAD_TOL, 4 * _relerr(r32, ref))
        lo = losses[v].cpu().numpy()
        assert abs(lo[0] - aux["critic_loss"]) <= 1e-4 * abs(aux["critic_loss"])
        assert abs(lo[1] - aux["actor_loss"]) <= 1e-4 * max(1e-3, abs(aux["actor_loss"]))
        # padded units receive exactly zero gradient, so Adam keeps them at zero forever
        lay = grp.lay
        gW1 = gh[v, lay.aW1:lay.aW1 + S * 320].reshape(S, 320)
        assert np.all(gW1[:, 307:] == 0) and np.any(gW1[:, :307] != 0)
        gcW2 = gh[v, lay.actor_size + lay.cW2:lay.actor_size + lay.cW2 + 384 * 160].reshape(384, 160)
        assert np.all(gcW2[307:320] == 0) and np.all(gcW2[377:] == 0) and np.all(gcW2[:, 153:] == 0)
        assert np.all(gh[v, lay.ag1 + 307:lay.ag1 + 320] == 0) and np.all(gh[v, lay.abe2 + 153:lay.abe2 + 160] == 0)


def _central_batches(n_agents, S, A, seed):
    rs = np.random.RandomState(seed)
    s = rs.normal(0, 1.5, size=(n_agents, 64, S)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(n_agents, 64, A)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n_agents, 64))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(n_agents, 64, S)).astype(np.float32)
    return t(s), t(a), t(r), t(s2)


@pytest.mark.parametrize("L", [3, 5])
def test_centralized_kernel_agrees_with_the_general_kernel(L, monkeypatch):
    """cen::learn_kernel_c (csrc/cen.hip: the centralized shapes' own eight-wave kernel, the shipped path of avd_learn_f32 at
    S = 4 L, A = L, widths x 1.2) against gen::learn_kernel_g (mlp.hip) on the same inputs -- the diagnostic build's
    AVD_LEARN_GENERAL=1 routes the same call to the general kernel. Same algorithm, different summation orders: every gradient tensor
    within 3e-6 of its max (both sit ~1e-6 from the float64 oracle, test_centralized_shapes_forward_and_learn), losses within 1e-6."""
    need_gpu()
    from avddpg_amd import _hip

    n_agents, S, A = 7, 4 * L, L
    batches = _central_batches(n_agents, S, A, seed=50 + L)
    got = {}
    for which in ("cen", "gen"):
        conf, grp = _perturbed_group(n_agents, S=S, A=A, hidd_mult=1.2, seed=43)
        losses = torch.zeros(n_agents, 2, device="cuda")
        if which == "gen":
            monkeypatch.setenv("AVD_LEARN_GENERAL", "1")
            with _hip.diag_library():
                g = grp.learn(*batches, 0, losses=losses)
                torch.cuda.synchronize()
            monkeypatch.delenv("AVD_LEARN_GENERAL")
        else:
            g = grp.learn(*batches, 0, losses=losses)
        got[which] = (g.cpu().numpy(), losses.cpu().numpy(), grp)
    (gc, lc, grp), (gg, lg, _) = got["cen"], got["gen"]
    assert not np.array_equal(gc, gg)  # (two kernels did run)
    assert np.allclose(lc, lg, rtol=1e-6, atol=1e-7)
    for v in range(n_agents):
        for x, y in zip(sum(grp.grads_as_lists(torch.from_numpy(gc[v])), []), sum(grp.grads_as_lists(torch.from_numpy(gg[v])), [])):
            assert _relerr(np.asarray(x), np.asarray(y)) <= 3e-6
    # shared weight sets (set_mod > 0: several agents read one set) take the same kernel: agent v reads set v % 2
    conf, grp = _perturbed_group(2, S=S, A=A, hidd_mult=1.2, seed=43)
    b6 = _central_batches(6, S, A, seed=60)
    g6 = grp.learn(*b6, 2)
    for v in (0, 2, 4):  # alone in a launch of one agent, set_mod = 0, the same batch reads set 0 as well
        g1 = grp.learn(*[x[v:v + 1].contiguous() for x in b6], 0)
        assert torch.equal(g1[0], g6[v])
    assert not torch.equal(g6[0], g6[1])


def test_centralized_update_in_chunks_on_two_streams_is_learn_then_adam_bit_for_bit():
    """avd_learn_update_f32 at the centralized shapes cuts the agents into chunks of 256: chunk c's learn kernel in the caller's stream,
    its whole-row Adam + Polyak pass on a side stream under chunk c + 1's learn kernel (csrc/cen.hip cen_launch_update). 700 models =
    three chunks (the last one ragged): weights, targets, BN statistics' soft update and moments are exactly those of avd_learn_f32 +
    avd_adam_polyak_f32, three updates in a row (theta ping-pong), and a repeat of the whole sequence gives the same bits (the
    two-stream schedule does not leak into the result)."""
    need_gpu()
    n_agents, S, A = 700, 20, 5
    runs = []
    for mode in ("update", "update", "separate"):
        conf, grp = _perturbed_group(n_agents, S=S, A=A, hidd_mult=1.2, seed=47)
        losses = torch.zeros(n_agents, 2, device="cuda")
        scratch = torch.empty(n_agents, grp.lay.theta_size, device="cuda")
        for k in range(3):
            b = _central_batches(n_agents, S, A, seed=70 + k)
            if mode == "update":
                grp.learn_update(*b, scratch, losses=losses)
            else:
                grp.apply(grp.learn(*b, 0, losses=losses))
        torch.cuda.synchronize()
        runs.append([x.clone() for x in (grp.theta, grp.theta_t, grp.stats_t, grp.m, grp.v, losses)])
    for x, y, z in zip(*runs):
        assert torch.equal(x, y) and torch.equal(x, z)
    assert not torch.equal(runs[0][0], _perturbed_group(n_agents, S=S, A=A, hidd_mult=1.2, seed=47)[1].theta)


def test_learn_update_next_action_epilogue_and_conditional_actor():
    """avd_learn_update_act_f32: next_action = actor(next_state) with the updated weights, bit-identical to the actor
    launch on the updated slab; avd_actor_forward_cond_f32 runs only when its device flag is non-zero."""
    need_gpu()
    n = 37
    conf, grp = _perturbed_group(n, S=4, seed=31)
    rs = np.random.RandomState(32)
    s = t(rs.normal(0, 1.5, size=(n, 64, 4)).astype(np.float32))
    a = t(rs.uniform(-2.5, 2.5, size=(n, 64, 1)).astype(np.float32))
    r = t(-np.abs(rs.normal(0, 0.3, size=(n, 64))).astype(np.float32))
    s2 = t(rs.normal(0, 1.5, size=(n, 64, 4)).astype(np.float32))
    nxt = t(rs.normal(0, 1.5, size=(n, 4)).astype(np.float32))
    scratch = torch.zeros(n, grp.lay.theta_size, device="cuda")
    out = torch.full((n,), 7.0, device="cuda")
    grp.learn_update(s, a, r, s2, scratch, next_states=nxt, next_actions=out)
    want = grp.actor(nxt, 0)  # grp.theta is the updated slab now
    assert torch.equal(out, want) and float(out.abs().max()) <= conf.action_high
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    keep = torch.full((n,), 3.0, device="cuda")
    grp.actor(nxt, 0, out=keep, run_if_nonzero=flag)
    assert torch.all(keep == 3.0)                      # flag 0: nothing written
    flag.fill_(1)
    grp.actor(nxt, 0, out=keep, run_if_nonzero=flag)
    assert torch.equal(keep, want)


def test_full_size_fused_update_is_bitwise_the_two_kernel_update_20480_agents():
    """BASELINE configs[1] size: avd_learn_update_act_f32 (learn_kernel_l<fused>: learn + Adam x2 + Polyak + next action, one
    launch) leaves exactly the weights, targets, moments and next actions of avd_learn_f32 + avd_adam_polyak_f32 +
    avd_actor_forward_f32 for all 20480 agents, two steps in a row (the second reads the ping-ponged slab)."""
    need_gpu()
    n = 4096 * 5
    conf = config.Config()
    g = torch.Generator(device="cuda").manual_seed(17)
    grp_a, grp_b = (vec.AgentGroup(n, 4, 1, conf, seed=11) for _ in range(2))
    grp_a.theta.add_(torch.randn(grp_a.theta.shape, device="cuda", generator=g) * 0.01 * (grp_a.theta != 0))
    grp_a.theta_t.copy_(grp_a.theta)
    grp_b.theta.copy_(grp_a.theta), grp_b.theta_t.copy_(grp_a.theta_t)
    scratch = torch.zeros(n, grp_a.lay.theta_size, device="cuda")
    grads = torch.zeros(n, grp_a.lay.theta_size, device="cuda")
    nxt_act = torch.zeros(n, device="cuda")
    for step in range(2):
        s = torch.randn(n, 64, 4, device="cuda", generator=g) * 1.5
        a = torch.rand(n, 64, 1, device="cuda", generator=g) * 5 - 2.5
        r = -torch.rand(n, 64, device="cuda", generator=g)
        s2 = torch.randn(n, 64, 4, device="cuda", generator=g) * 1.5
        nxt = torch.randn(n, 4, device="cuda", generator=g) * 1.5
        grp_a.learn_update(s, a, r, s2, scratch, next_states=nxt, next_actions=nxt_act)
        grp_b.learn(s, a, r, s2, 0, grads=grads)
        grp_b.apply(grads)
        for x, y in ((grp_a.theta, grp_b.theta), (grp_a.theta_t, grp_b.theta_t), (grp_a.m, grp_b.m), (grp_a.v, grp_b.v),
                     (grp_a.stats_t, grp_b.stats_t)):
            assert torch.equal(x, y), step
        assert torch.equal(nxt_act, grp_b.actor(nxt, 0))
    assert int(grp_a.step[0]) == int(grp_b.step[-1]) == 2 and torch.isfinite(grp_a.theta).all()


@pytest.mark.parametrize("S,P,M", [(4, 8, 1), (4, 13, 3), (3, 37, 5), (4, 4096, 5)])
def test_shared_set_actor_rows_are_bitwise_the_batch1_kernel(S, P, M):
    """avd_actor_forward_f32 with shared weight sets (set_mod = M) evaluates 8 agents of a set per workgroup; every row goes
    through the batch-1 kernel's exact operation sequence, so the actions are the same BITS as one workgroup per agent on
    replicated weights (the reference's P copies of vehicle m's actor, workers/trainer.py:121-128, 287-289). P = 13 / 37
    leave a ragged last group; rows of 4 floats with S = 3 exercise the stride."""
    need_gpu()
    conf, shared = _perturbed_group(M, S=S, seed=131)
    rs = np.random.RandomState(132)
    n = P * M
    x = t(rs.normal(0, 1.5, size=(n, 4)).astype(np.float32))
    got = shared.actor(x, set_mod=M, x_stride=4)
    # the same weights, one copy per agent (agent v = p*M + m holds set m), through the one-agent-per-workgroup kernel
    per = vec.AgentGroup(n, S, 1, conf)
    per.theta.copy_(shared.theta.repeat(P, 1))
    per.stats.copy_(shared.stats.repeat(P, 1))
    ref = per.actor(x, set_mod=0, x_stride=4)
    assert torch.equal(got, ref) and got.abs().max() > 0.05
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    keep = torch.full_like(got, 7.0)
    shared.actor(x, set_mod=M, x_stride=4, out=keep, run_if_nonzero=flag)
    assert (keep == 7.0).all()  # flag 0: the conditional launch leaves `out` alone
    flag.fill_(1)
    shared.actor(x, set_mod=M, x_stride=4, out=keep, run_if_nonzero=flag)
    assert torch.equal(keep, ref)


def test_federated_server_facade_matches_reference_table_and_oracle(golden_dir):
    """avddpg_amd.federated.Server keeps src/server/federated.py's class API (Server(name, debug).get_avg_params /
    .get_weighted_avg_params on lists of systems of members of per-layer arrays) over avd_fed_sum / avd_fed_finalize:
    the reference's own table (src/server/test_federated.py:26-42, golden G7) and the oracle's restatement on
    network-shaped gradient lists (24 layers of different shapes)."""
    import json
    import os

    from avddpg_amd import federated

    need_gpu()
    server = federated.Server("fed", False)
    tab = json.load(open(os.path.join(golden_dir, "g7_federated.json")))
    pl, w = tab["grads_list"], tab["weights"]
    P, M = 2, 2
    sysu = [[[np.array(pl[p][m][i], dtype=np.float32) for i in range(3)] for p in range(P)] for m in range(M)]
    got = server.get_avg_params(sysu)
    assert len(got) == M and all(len(g) == 3 for g in got)
    for m in range(M):
        for i in range(3):
            assert got[m][i].dtype == np.float32 and got[m][i].shape == np.shape(pl[0][m][i])
            assert np.allclose(got[m][i], tab["interfrl_unweighted"][m][i], rtol=1e-6)
    sysw = [[[np.float32(w[p][m]) * sysu[m][p][i] for i in range(3)] for p in range(P)] for m in range(M)]
    ws = [sum(w[p][m] for p in range(P)) for m in range(M)]
    gotw = server.get_weighted_avg_params(sysw, ws)
    for m in range(M):
        for i in range(3):
            assert np.allclose(gotw[m][i], tab["interfrl_weighted"][m][i], rtol=1e-6)
    # network-shaped lists: 5 platoons x 3 vehicles of critic gradients (14 tensors) from the learn kernel
    conf, grp = _perturbed_group(3, S=4, seed=121)
    rs = np.random.RandomState(122)
    Pn, Mn = 5, 3
    n = Pn * Mn
    s = rs.normal(0, 1.5, size=(n, 64, 4)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(n, 64, 1)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n, 64))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(n, 64, 4)).astype(np.float32)
    g = grp.learn(t(s), t(a), t(r), t(s2), Mn)
    lists = [[grp.grads_as_lists(g[p * Mn + m])[0] for p in range(Pn)] for m in range(Mn)]  # [vehicle][platoon][layer]
    got = server.get_avg_params(lists)
    ref = ofed.get_avg_params(lists)
    slab = vec.fed_mean(g, Pn, Mn)  # the slab path VecTrainer uses
    for m in range(Mn):
        for x, y, z in zip(got[m], ref[m], grp.grads_as_lists(slab[m])[0]):
            assert x.shape == y.shape and np.allclose(x, y, rtol=1e-5, atol=1e-9) and np.array_equal(x, z)
    wts = rs.uniform(0.5, 6.0, size=(Mn, Pn)).astype(np.float32)
    listw = [[[wts[m, p] * layer for layer in lists[m][p]] for p in range(Pn)] for m in range(Mn)]
    sums = [float(wts[m].sum()) for m in range(Mn)]
    gotw, refw = server.get_weighted_avg_params(listw, sums), ofed.get_weighted_avg_params(listw, sums)
    for m in range(Mn):
        for x, y in zip(gotw[m], refw[m]):
            assert np.allclose(x, y, rtol=1e-5, atol=1e-9)
    with pytest.raises(ValueError, match="members"):
        server.get_avg_params([lists[0], lists[1][:3]])


@pytest.mark.parametrize("S,P,M", [(4, 100, 5), (3, 33, 2), (4, 4096, 5)])
def test_actor_on_the_f32_matrix_cores_for_shared_sets_matches_rows_kernel_and_oracle(S, P, M):
    """avd_actor_forward_set_f32 (csrc/act.hip: v_mfma_f32_32x32x2_f32, exact f32 products) vs the batch-1 rows kernel with
    shared sets (same f32 values up to the summation order: 2e-6 of the action range) and vs the float64 oracle (FWD_TOL),
    P not a multiple of the 32-row tile, and the conditional-launch flag."""
    need_gpu()
    conf, grp = _perturbed_group(M, S=S, seed=131)
    rs = np.random.RandomState(132)
    x = rs.normal(0, 1.5, size=(P * M, 4)).astype(np.float32)  # env layout: 4 floats per agent, first S are the observation
    ref = grp.actor(t(x), set_mod=M, x_stride=4).cpu().numpy()
    got = grp.actor_set(t(x), P * M, x_stride=4)
    assert np.abs(got.cpu().numpy() - ref).max() <= 2e-6 * 2.5 and np.abs(ref).max() > 0.05
    for v in (0, P * M // 2 + 1, P * M - 1):
        aw = [w.astype(np.float64) for w in grp.get_weights(v % M, "actor")]
        want = omlp.actor_forward(aw, x[v:v + 1, :S].astype(np.float64), 2.5)[0]
        assert abs(float(got[v]) - float(np.ravel(want)[0])) <= FWD_TOL * 2.5
    keep = torch.full((P * M,), 7.0, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    grp.actor_set(t(x), P * M, x_stride=4, out=keep, run_if_nonzero=flag)
    assert torch.all(keep == 7.0)
    flag.fill_(1)
    grp.actor_set(t(x), P * M, x_stride=4, out=keep, run_if_nonzero=flag)
    assert torch.equal(keep, got)
