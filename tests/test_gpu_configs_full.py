"""BASELINE.json configs[2] (4096 platoons x 10 vehicles) and configs[4] (hidden = 1024, bf16, 4096 platoons) at FULL
size through replay / learn / update / trainer, not only the env kernel: oracle spot checks on individual agents (the
oracle finishes those in seconds) plus size-independent properties -- determinism, lane independence, fused ==
two-kernel, mean-of-halves -- for the whole batch."""
import numpy as np
import pytest
import torch

from avddpg_amd import config, trainer, vec
from oracle import mlp as omlp
from tests.gpu_util import need_gpu
from tests.test_gpu_mlp import GRAD_TOL, _nets, _perturbed_group, _relerr

pytestmark = pytest.mark.gpu


def _free(*objs):
    del objs
    torch.cuda.empty_cache()


def test_config3_trainer_4096x10_learns_like_the_oracle_and_fused_equals_two_kernel_path():
    """configs[2]: 40960 agents (pl_size = 10 is beyond the reference's `Platoon` length check, src/environment.py:84-85,
    which fires after full construction; the dynamics are defined for any L). Two VecTrainers on the same Philox
    streams, one with the fused learn+Adam+Polyak kernel, one with learn -> gradient slab -> Adam/Polyak: 70 steps =
    5 updates of every agent. Checked: the replay gate opens at the 65th add for all 40960 agents; the two paths end in
    bit-identical weights / targets / moments (determinism across kernels at this grid size); three agents' gradients of
    the 70th step against the float64 oracle on the batch they actually sampled; every platoon's chain of 10 vehicles
    fed the replay rows the env produced."""
    need_gpu()
    P, L, steps = 4096, 10, 70
    mk = lambda fused: trainer.VecTrainer(config.Config(num_platoons=P, pl_size=L, buffer_size=128), rng="device", auto_reset=True,
                                          seed=3, fused_update=fused)
    a, b = mk(True), mk(False)
    assert a.n_agents == b.n_agents == P * L and a.agents.n_sets == P * L and not a.shared
    for vt in (a, b):
        vt.reset_episode()
    for i in range(steps - 1):
        a.step()
        b.step()
    assert a.updates == b.updates == (steps - 1 - 64) * P * L
    spots = (0, 17 * L + 9, P * L - 1)  # first / a chain tail / last agent
    pre = {v: _nets(b.agents, v, np.float64) for v in spots}
    pre32 = {v: _nets(b.agents, v, np.float32) for v in spots}
    a.step()
    b.step()
    torch.cuda.synchronize()
    # replay rows of the last step: [s a r s'] with s' = the env's current state, s = its previous one
    row = b.replay.ring[:, (b.replay.buffer_counter - 1) % b.replay.cap]
    assert torch.equal(row[:, :4], b.env.x_prev.view(P * L, 4)) and torch.equal(row[:, 6:10], b.env.x.view(P * L, 4))
    assert torch.equal(row[:, 4], b.actions.view(-1)) and torch.equal(row[:, 5], b.env.reward.view(-1))
    for v in spots:
        batch = (b.replay.s[v].cpu().numpy(), b.replay.a[v].cpu().numpy(), b.replay.r[v].cpu().numpy()[:, None],
                 b.replay.s2[v].cpu().numpy())
        cg, ag, _ = omlp.learn(batch, *pre[v])
        cg32, ag32, _ = omlp.learn(tuple(x.astype(np.float32) for x in batch), *pre32[v])
        gcg, gag = b.agents.grads_as_lists(b.grads[v])
        for got, ref, r32 in zip(gcg + gag, cg + ag, cg32 + ag32):
            assert _relerr(got, ref) <= max(GRAD_TOL, 4 * _relerr(r32, ref)), v
    for name in ("theta", "theta_t", "stats_t", "m", "v"):
        assert torch.equal(getattr(a.agents, name), getattr(b.agents, name)), name
    assert torch.equal(a.env.x, b.env.x) and torch.equal(a.replay.ring, b.replay.ring)
    assert torch.isfinite(a.agents.theta).all() and int(a.agents.step.min()) == int(a.agents.step.max()) == steps - 64
    # the agents did diverge from each other (independent replay streams), i.e. 40960 different updates were made
    assert not torch.equal(a.agents.theta[0], a.agents.theta[1])
    _free(a, b)


def test_centralized_trainer_4096x5_pipeline_of_16_chunks_equals_the_two_kernel_path_and_tracks_the_oracle():
    """The centralized framework (SURVEY f-3) at BASELINE configs[1]'s size: 4096 models of S = 20 / A = 5 (widths x 1.2). The fused
    update runs as 16 chunks of 256 models -- learn kernels on the caller's stream, their Adam + Polyak passes on a side stream
    (csrc/cen.hip) -- the other trainer does learn -> gradient slab -> Adam / Polyak in two launches: 69 steps = 5 updates of every
    model on the same Philox streams end in bit-identical weights / targets / BN statistics / moments, three models' gradients of the
    last step sit at the float64 oracle on the batch they sampled, and the models did take 4096 different updates."""
    need_gpu()
    P, L, steps = 4096, 5, 69
    mk = lambda fused: trainer.VecTrainer(config.Config(num_platoons=P, pl_size=L, buffer_size=128, framework="centralized"), rng="device",
                                          auto_reset=True, seed=3, fused_update=fused)
    a, b = mk(True), mk(False)
    assert a.n_agents == b.n_agents == P and a.agents.n_sets == P and not a.shared
    assert (a.agents.lay.S, a.agents.lay.A, a.agents.lay.H1, a.agents.lay.H2, a.agents.lay.Ha) == (20, 5, 320, 160, 64)
    for vt in (a, b):
        vt.reset_episode()
    for i in range(steps - 1):
        a.step()
        b.step()
    spots = (0, 255 + 256 * 7, P - 1)  # first model / the last of a middle chunk / the last model
    pre = {v: _nets(b.agents, v, np.float64) for v in spots}
    pre32 = {v: _nets(b.agents, v, np.float32) for v in spots}
    a.step()
    b.step()
    torch.cuda.synchronize()
    for v in spots:
        batch = (b.replay.s[v].cpu().numpy(), b.replay.a[v].cpu().numpy(), b.replay.r[v].cpu().numpy()[:, None],
                 b.replay.s2[v].cpu().numpy())
        cg, ag, _ = omlp.learn(batch, *pre[v])
        cg32, ag32, _ = omlp.learn(tuple(x.astype(np.float32) for x in batch), *pre32[v])
        gcg, gag = b.agents.grads_as_lists(b.grads[v])
        for got, ref, r32 in zip(gcg + gag, cg + ag, cg32 + ag32):
            assert _relerr(got, ref) <= max(GRAD_TOL, 4 * _relerr(r32, ref)), v
    for name in ("theta", "theta_t", "stats_t", "m", "v"):
        assert torch.equal(getattr(a.agents, name), getattr(b.agents, name)), name
    assert torch.equal(a.env.x, b.env.x) and torch.equal(a.replay.ring, b.replay.ring) and torch.equal(a.losses, b.losses)
    assert torch.isfinite(a.agents.theta).all() and int(a.agents.step.min()) == int(a.agents.step.max()) == steps - 64
    assert not torch.equal(a.agents.theta[0], a.agents.theta[1])
    _free(a, b)


def test_config3_learn_kernels_lane_independent_at_40960_agents():
    """Lane independence + determinism of Trainer.learn at configs[2]'s grid (40960 workgroups): a duplicated agent gives
    the same bits wherever it sits; two launches agree bit for bit."""
    need_gpu()
    n = 4096 * 10
    grp = vec.AgentGroup(n, 4, 1, config.Config(), seed=12)
    g = torch.Generator(device="cuda").manual_seed(6)
    grp.theta.add_(torch.randn(grp.theta.shape, device="cuda", generator=g) * 0.01 * (grp.theta != 0))
    grp.theta_t.copy_(grp.theta)
    s = torch.randn(n, 64, 4, device="cuda", generator=g) * 1.5
    a = torch.rand(n, 64, 1, device="cuda", generator=g) * 5 - 2.5
    r = -torch.rand(n, 64, device="cuda", generator=g)
    s2 = torch.randn(n, 64, 4, device="cuda", generator=g) * 1.5
    for dst in (6, 20481, n - 1):
        for x in (grp.theta, grp.theta_t, grp.stats, grp.stats_t, s, a, r, s2):
            x[dst].copy_(x[5])
    g1 = grp.learn(s, a, r, s2, 0)
    g2 = grp.learn(s, a, r, s2, 0)
    assert torch.equal(g1, g2) and torch.isfinite(g1).all()
    for dst in (6, 20481, n - 1):
        assert torch.equal(g1[dst], g1[5])
    assert not torch.equal(g1[5], g1[7])
    _free(grp, g1, g2)


def _wide_group(M, seed):
    return _perturbed_group(M, S=4, seed=seed, actor_layer1_size=1024, actor_layer2_size=1024, critic_layer1_size=1024,
                            critic_layer2_size=1024)


def test_config5_hidden1024_against_the_oracle_on_4096_rows_per_set():
    """configs[4]'s widths with 64 platoons per set = 4096 rows per weight set (20480 rows in all): every gradient
    tensor against the float64 oracle on the set's concatenated batch. bf16 GEMM operands behind 1024-long reductions:
    6 % of each tensor's max (8 % allowed at 512 rows in tests/test_gpu_wide.py; more rows average the rounding)."""
    need_gpu()
    P, M, B = 64, 5, 64
    conf, grp = _wide_group(M, 111)
    rs = np.random.RandomState(112)
    rows = P * B
    s = rs.normal(0, 1.5, size=(M, rows, 4)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(M, rows, 1)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(M, rows))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(M, rows, 4)).astype(np.float32)
    tt = lambda x: torch.from_numpy(x).cuda()
    losses = torch.zeros(M, 2, device="cuda")
    g = grp.learn_shared(tt(s), tt(a), tt(r), tt(s2), P * M, losses=losses)
    torch.cuda.synchronize()
    worst = 0.0
    for k in (0, M - 1):
        cg, ag, aux = omlp.learn((s[k], a[k], r[k][:, None], s2[k]), *_nets(grp, k, np.float64))
        gcg, gag = grp.grads_as_lists(g[k])
        for got, ref in zip(gcg + gag, cg + ag):
            worst = max(worst, _relerr(got, ref))
        assert abs(float(losses[k, 0]) - aux["critic_loss"]) <= 1e-2 * abs(aux["critic_loss"])
    assert worst <= 6e-2, worst
    _free(grp, g)


def test_config5_hidden1024_full_size_mean_of_halves_and_determinism_of_the_step():
    """configs[4] at 4096 platoons x 5 sets x 64 rows, hidden 1024 (25.6 GB of workspace): the mean gradient over all
    platoons equals the average of the mean gradients over the two halves of the platoons (rows are independent:
    inference-mode BN), 1e-3 of each slab's max (same operand rounding, split-K summation order differs); and a
    VecTrainer of that shape takes federated steps with finite results."""
    need_gpu()
    P, M, B, S = 4096, 5, 64, 4
    conf, grp = _wide_group(M, 121)
    g = torch.Generator(device="cuda").manual_seed(122)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    s, a = 1.5 * rn(M, P * B, S), 2.5 * (2 * torch.rand(M, P * B, 1, device="cuda", generator=g) - 1)
    r, s2 = -rn(M, P * B).abs() * 0.3, 1.5 * rn(M, P * B, S)
    full = grp.learn_shared(s, a, r, s2, P * M).clone()
    h = P * B // 2
    halves = []
    for lo in (0, h):
        sl = lambda x: x[:, lo:lo + h].contiguous()
        halves.append(grp.learn_shared(sl(s), sl(a), sl(r), sl(s2), P * M // 2).clone())
    avg = 0.5 * (halves[0] + halves[1])
    assert torch.isfinite(full).all() and full.abs().max() > 0
    lay = grp.lay
    for lo, hi in ((0, lay.actor_size), (lay.actor_size, lay.theta_size)):
        d = (full[:, lo:hi] - avg[:, lo:hi]).abs().max().item()
        assert d <= 1e-3 * full[:, lo:hi].abs().max().item(), (lo, d)
    assert not torch.allclose(halves[0], halves[1])
    _free(grp, full, halves, s, a, r, s2)
    c = config.Config(num_platoons=P, pl_size=M, buffer_size=128, fed_method="interfrl", weighted_average_enabled=False,
                      actor_layer1_size=1024, actor_layer2_size=1024, critic_layer1_size=1024, critic_layer2_size=1024)
    vt = trainer.VecTrainer(c, rng="device", auto_reset=True)
    assert vt.shared and vt.shared_engine == "batched"
    th0 = vt.agents.theta.clone()
    vt.reset_episode()
    for _ in range(67):
        vt.step()
    torch.cuda.synchronize()
    assert vt.updates == 3 * P * M and torch.isfinite(vt.agents.theta).all() and not torch.equal(vt.agents.theta, th0)
    assert (vt.set_losses[:, 0] >= 0).all() and int(vt.agents.step[0]) == 3
    _free(vt)


def test_config5_hidden1024_repeats_of_a_learn_agree():
    """A race detector for the streamed kernels of the wide learner (csrc/wide.hip, fw::*): the same learn_shared call 120 times
    on the same inputs at configs[4]'s full size. Beyond the summation order of f32 atomics (~1e-6 of a slab's max) every
    repeat must agree -- a fragment read that beats its LDS-DMA request shows up as garbage in one wave's features once in a
    few dozen launches (r03: fw::dx_gen_kernel read chunk 2 of a tile unwaited; tools/determinism_c5.py runs longer)."""
    need_gpu()
    P, M, B, S = 4096, 5, 64, 4
    conf, grp = _wide_group(M, 131)
    g = torch.Generator(device="cuda").manual_seed(132)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    s, a = 1.5 * rn(M, P * B, S), 2.5 * (2 * torch.rand(M, P * B, 1, device="cuda", generator=g) - 1)
    r, s2 = -rn(M, P * B).abs() * 0.3, 1.5 * rn(M, P * B, S)
    lay = grp.lay
    ref = grp.learn_shared(s, a, r, s2, P * M).clone()
    scale = [ref[:, :lay.actor_size].abs().max(), ref[:, lay.actor_size:].abs().max()]
    worst = torch.zeros(2, device="cuda")
    for _ in range(120):
        out = grp.learn_shared(s, a, r, s2, P * M)
        worst[0] = torch.maximum(worst[0], (out[:, :lay.actor_size] - ref[:, :lay.actor_size]).abs().max() / scale[0])
        worst[1] = torch.maximum(worst[1], (out[:, lay.actor_size:] - ref[:, lay.actor_size:]).abs().max() / scale[1])
    w = worst.tolist()
    assert torch.isfinite(ref).all() and max(w) <= 1e-4, w
    _free(grp, ref, s, a, r, s2)


# ---- the bench's primary engine (interfrl, shared weight sets, csrc/fsplit.hip) at full size, at TRAINER level -------------------
SPLIT_TOL = 2e-5  # tests/test_gpu_fsplit.py
# (Nearly untrained networks -- 70 steps from the initialiser: critic output weights U(+-3e-4), actor gradients of 1e-8 -- were
# the worst case of r03's bf16 operand pairs: 2.1e-5 on actor tensors, 2.07e-5 on cWa at 4096 x 10. With every operand an fp16
# pair (r04) the same states measure <= 9e-6, one tensor 1.6e-5 where the exact-f32 engine shows the same 1.6e-5 -- a relu tie;
# tools/r04_trainer_state_errors.py @ tag r06-pre-prune, profiles/r04_trainer_state_errors.txt. No looser constant is needed here.)


@pytest.mark.parametrize("L", [5, 10])
def test_split_engine_trainer_at_full_size_tracks_the_exact_engine_and_the_oracle(L):
    """BASELINE configs[1] (4096 x 5) and configs[2] (4096 x 10) through VecTrainer with the engine the bench line runs
    (interfrl, one weight set per vehicle index, shared_engine='fused3' = avd_learn_set_split_f16x3) -- VERDICT r03 #6. Two
    trainers on the same device Philox streams, 'fused3' and the exact-f32 'per_agent' engine (learn_kernel_l per agent +
    fed_sum), 70 steps = 6 federated updates of every set:
      * before the first update (65th add) the two differ only by the acting kernel's f32 summation order: 2e-6; afterwards
        actions within 2e-4 of the action range and states within 2e-4, step by step (the tolerance of
        tests/test_gpu_fsplit.py::test_trainer_split_engine_tracks_per_agent_engine_under_interfrl at 6 x 3);
      * the 70th step's gradient of the first and last weight set against the FLOAT64 ORACLE on the 262 144 rows the trainer
        actually sampled for it, from the weights it actually held: every tensor within SPLIT_TOL = 2e-5 of its max -- except
        that ONE (set, net) group may sit at up to 4e-5: among 262 144 rows x ~1000 differentiated units a handful of
        pre-activations lie within float32 resolution of 0, and one such relu-tie row decided the other way moves every tensor of
        the net behind it by the same ~2e-5 (all of set 4's actor tensors at L = 5 in one build, none in the next; the exact-f32
        engine shows the same on other sets: profiles/r04_trainer_state_errors.txt; tie accounting at the parity sizes:
        tests/test_gpu_fsplit.py). The exact-f32 engine's error on the very same batch and weights is printed beside it."""
    need_gpu()
    P, steps = 4096, 70
    mk = lambda engine: trainer.VecTrainer(config.Config(num_platoons=P, pl_size=L, buffer_size=128, fed_method="interfrl",
                                                         weighted_average_enabled=False),
                                           rng="device", auto_reset=True, seed=7, shared_engine=engine)
    a, b = mk("per_agent"), mk("fused3")
    assert a.shared and b.shared and b.shared_engine == "fused3" and b.agents.n_sets == L and b.grads is None
    for vt in (a, b):
        vt.reset_episode()
    high = a.conf.action_high
    for i in range(steps - 1):
        a.step()
        b.step()
        tol = 2e-6 if i < 65 else 2e-4
        assert (a.actions - b.actions).abs().max().item() <= tol * high, i
        assert (a.env.x - b.env.x).abs().max().item() <= tol * max(1.0, a.env.x.abs().max().item()), i
    assert a.updates == b.updates == (steps - 1 - 64) * P * L
    spots = (0, L - 1)
    pre = {k: _nets(b.agents, k, np.float64) for k in spots}
    held = [x.clone() for x in (b.agents.theta, b.agents.stats, b.agents.theta_t, b.agents.stats_t)]
    a.step()
    b.step()
    torch.cuda.synchronize()
    assert (a.actions - b.actions).abs().max().item() <= 2e-4 * high
    B = 64
    rp = b.replay
    sn, an, rn, s2n = (x.cpu().numpy() for x in (rp.s, rp.a, rp.r, rp.s2))
    from tests.test_gpu_fset import NAMES
    # the exact-f32 engine on the same batch from the same (pre-update) weights
    ex = vec.AgentGroup(L, b.S, b.A, b.conf, seed=1)
    for dst, src in zip((ex.theta, ex.stats, ex.theta_t, ex.stats_t), held):
        dst.copy_(src)
    exact = vec.fed_mean(ex.learn(rp.s, rp.a, rp.r, rp.s2, L), P, L, method=b.conf.interfrl)
    worst, worst_exact, over = 0.0, 0.0, set()
    for k in spots:
        sel = np.arange(P) * L + k
        cat = lambda x: x[sel].reshape(P * B, *x.shape[2:])
        cg, ag, _ = omlp.learn((cat(sn), cat(an), cat(rn)[:, None], cat(s2n)), *pre[k])
        gcg, gag = b.agents.grads_as_lists(b.set_grads[k])
        ecg, eag = ex.grads_as_lists(exact[k])
        for name, got, eng, ref in zip(NAMES, gcg + gag, ecg + eag, cg + ag):
            e, ee = _relerr(got, ref), _relerr(eng, ref)
            worst, worst_exact = max(worst, e), max(worst_exact, ee)
            assert e <= 2 * SPLIT_TOL, (k, name, e, ee)
            if e > SPLIT_TOL:
                over.add((k, name[0]))  # (set, net: 'a' / 'c')
    assert len(over) <= 1, over
    # the allowance must be EXPLAINED, not granted. For the one (set, net) group over SPLIT_TOL: find the relu-tie
    # rows among the set's 262 144 (float64, from the weights and the batch actually used: tests/test_gpu_fsplit.py::_tie_mask),
    # leave the 64-row tiles that hold one out of BOTH means -- the split engine on the remaining agents of that set, the float64
    # oracle on the same rows -- and every tensor of the group must be back inside SPLIT_TOL.
    # (nothing over the tolerance in this build: the accounting still runs once, on the first set's critic, so that the path is
    # exercised by every run -- without the tie tiles the set must of course be inside the tolerance as well)
    for k, net in (sorted(over) or [(spots[0], "c")]):
        from tests.test_gpu_fsplit import _tie_mask
        sel = np.arange(P) * L + k
        # (tie = 2e-7 of the layer's largest pre-activation ~ two float32 ulps of it: at 262 144 rows x ~1000 units the 1e-6 of the
        # small parity cases would call half the tiles tied)
        ties = _tie_mask(ex, L, b.S, sn, an, tie=2e-7, sets=[k])[sel]  # [P, 64]
        tie_tiles = ties.any(axis=1)
        assert tie_tiles.any() or (k, net) not in over, (k, net, "over the tolerance without a relu-tie row in the set")
        keep = sel[~tie_tiles]
        assert len(keep) >= int(0.8 * P), (k, net, int(tie_tiles.sum()))
        one = vec.AgentGroup(1, b.S, b.A, b.conf, seed=1)  # the set on its own: agent v -> set v % 1
        for dst, src in zip((one.theta, one.stats, one.theta_t, one.stats_t), held):
            dst.copy_(src[k:k + 1])
        kt = torch.from_numpy(keep).cuda()
        g_keep = one.learn_set_split(rp.s[kt].contiguous(), rp.a[kt].contiguous(), rp.r[kt].contiguous(), rp.s2[kt].contiguous(), len(keep))
        cat = lambda x: x[keep].reshape(len(keep) * B, *x.shape[2:])
        cg, ag, _ = omlp.learn((cat(sn), cat(an), cat(rn)[:, None], cat(s2n)), *pre[k])
        gcg, gag = one.grads_as_lists(g_keep[0])
        for name, got, ref in zip(NAMES, gcg + gag, cg + ag):
            if name[0] == net:
                assert _relerr(got, ref) <= SPLIT_TOL, (k, name, _relerr(got, ref), "still over the tolerance without the tie tiles")
        print(f"4096 x {L}: set {k} net '{net}' {'over' if (k, net) in over else 'inside'} SPLIT_TOL on the whole set; {int(ties.sum())} relu-tie rows in "
              f"{int(tie_tiles.sum())} tiles; inside SPLIT_TOL with those tiles left out of both means")
        del one
    # weights after 6 updates: Adam normalises every step to |dw| <= lr, so the engines may differ by a fraction of lr * updates
    n_upd = steps - 64
    lay = a.agents.lay
    for lr, lo, hi in ((a.conf.actor_lr, 0, lay.actor_size), (a.conf.critic_lr, lay.actor_size, lay.theta_size)):
        d = (a.agents.theta[:, lo:hi] - b.agents.theta[:, lo:hi]).abs()
        assert d.max().item() <= 2 * lr * n_upd and d.mean().item() <= 0.02 * lr * n_upd
    assert torch.isfinite(b.agents.theta).all() and int(b.agents.step[0]) == n_upd
    print(f"4096 x {L} fused3 trainer, 70th step, sets {spots}: worst tensor error vs float64 oracle {worst:.1e} (exact-f32 engine on the same batch: {worst_exact:.1e})")
    _free(a, b)


def _run_episodes(engine, steps, diag=False):
    """VecTrainer 4096 x 5 interfrl on device Philox streams, host episode loop (any platoon terminal ends the episode for all
    platoons, workers/trainer.py:268-269): returns (episodic rewards [episodes, P, M], the step each episode ended at, theta).
    diag: run the per_agent engine with its OTHER exact-f32 learn kernel (learn_kernel_t: a switch of the diagnostic build)."""
    import contextlib
    import os

    from avddpg_amd import _hip

    ctx = _hip.diag_library() if diag else contextlib.nullcontext()
    if diag:
        os.environ["AVD_LEARN_KERNEL"] = "fast"
    try:
        with ctx:
            conf = config.Config(num_platoons=4096, pl_size=5, buffer_size=2048, fed_method="interfrl", weighted_average_enabled=False,
                                 episode_sim_time=20.0)
            vt = trainer.VecTrainer(conf, rng="device", shared_sets=True, shared_engine=engine, seed=5)
            eps, ends, i = [], [], 0
            vt.reset_episode()
            for n in range(steps):
                done = vt.step(vt.episode, i)
                i += 1
                if done or i >= conf.steps_per_episode:
                    eps.append(vt.ep_reward.cpu().numpy().copy())  # float32 counters (workers/trainer.py:249, 321)
                    ends.append(n)
                    vt.episode += 1
                    vt.reset_episode()
                    i = 0
            torch.cuda.synchronize()
            out = (np.array(eps), ends, vt.agents.theta.cpu().numpy().copy(), int(vt.agents.step[0]))
            del vt
    finally:
        if diag:
            os.environ.pop("AVD_LEARN_KERNEL", None)
    torch.cuda.empty_cache()
    return out


def test_split_engine_reward_curves_at_scale_stay_inside_the_float32_noise_floor():
    """north_star: "matching reference episode-reward curves within tolerance" (workers/trainer.py:510-517) -- VERDICT r03 #1(d),
    for the engine the headline runs, at the headline's size: VecTrainer 4096 x 5, interfrl, device Philox streams, 2100 steps
    = 2036 federated updates of every weight set = 65 episodes (an episode ends for all platoons when any platoon is terminal),
    three times on identical streams:
        E  the exact-f32 engine           (per_agent: learn_kernel_l per agent + fed_sum)
        F  the exact-f32 engine, with its other learn kernel (learn_kernel_t: the same exact f32 products, another summation order)
        S  the split-operand engine       (fused3: avd_learn_set_split_f16x3)
    |E - F| is what float32 arithmetic itself leaves undetermined after 2036 Adam updates (Adam turns a gradient component's
    sign into a step of lr: differences of 1e-7 in a gradient near zero become 1e-5 in a weight) -- the floor of the metric.
    Asserted: S ends every episode at the same step as E; every agent's episodic reward within max(1e-3, 1.5 x floor) of E's,
    relative to the episode's mean |reward|; the platoon-mean curve per vehicle index within 1e-4; mean |theta_S - theta_E| within
    max(1e-5, 2 x floor). Measured (profiles/r04_precision_probe.txt): rewards 1.3e-3 (floor 9.6e-4), mean curve 2.3e-5 (3.9e-5),
    mean |dtheta| 2.8e-5 (2.6e-5): the split engine is inside the float32 run-to-run class on the quantity the reference plots."""
    need_gpu()
    steps = 2100
    rE, eE, thE, uE = _run_episodes("per_agent", steps)
    rS, eS, thS, uS = _run_episodes("fused3", steps)
    rF, eF, thF, uF = _run_episodes("per_agent", steps, diag=True)
    assert uE == uS == uF == steps - 64 >= 2000 and len(eE) >= 40
    assert eS == eE, "the split engine ended an episode at another step than the exact engine"

    def curves(r, ends):
        n = next((k for k in range(min(len(ends), len(eE))) if ends[k] != eE[k]), min(len(ends), len(eE)))  # episodes on E's schedule
        d = np.abs(r[:n] - rE[:n])
        agent = (d.max(axis=(1, 2)) / np.abs(rE[:n]).mean(axis=(1, 2))).max()
        mean = (np.abs(r[:n].mean(axis=1) - rE[:n].mean(axis=1)).max(axis=1) / np.abs(rE[:n].mean(axis=1)).max(axis=1)).max()
        return n, agent, mean

    nS, agentS, meanS = curves(rS, eS)
    nF, agentF, meanF = curves(rF, eF)
    dS, dF = np.abs(thS - thE).mean(), np.abs(thF - thE).mean()
    print(f"2036 updates, 65 episodes at 4096 x 5: split vs exact: agent rewards {agentS:.2e}, platoon-mean curve {meanS:.2e}, mean |dtheta| {dS:.2e}; "
          f"float32 floor (learn_kernel_t vs learn_kernel_l, {nF} episodes on one schedule): {agentF:.2e}, {meanF:.2e}, {dF:.2e}")
    assert nS == len(eE)
    assert agentS <= max(1e-3, 1.5 * agentF), (agentS, agentF)
    assert meanS <= 1e-4, meanS
    assert dS <= max(1e-5, 2.0 * dF), (dS, dF)
