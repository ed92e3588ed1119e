"""GPU parity of the fused shared-weight-set learner (csrc/fset.hip, avd_learn_set_fused_bf16): Trainer.learn
(workers/trainer.py:472-508) + federated mean (src/server/federated.py:47-63, 99-118) for agents that share their
networks, against (a) the float64 oracle on the concatenated batch, (b) the exact-f32 per-agent kernel + fed_mean kernels,
(c) the layer-wise bf16 learner (csrc/wide.hip), and at full size through size-independent properties. Tolerances are bf16
ones (8 significant bits on the second-layer GEMM operands, f32 accumulation) and are written next to each assertion."""
import numpy as np
import pytest
import torch

from avddpg_amd import _hip, config, vec
from oracle import mlp as omlp
from tests.gpu_util import need_gpu, t
from tests.test_gpu_mlp import _nets, _perturbed_group, _relerr

pytestmark = pytest.mark.gpu

NAMES = ["cWs", "cbs", "cWa", "cba", "cgs", "cbes", "cga", "cbea", "cW2", "cb2", "cg3", "cbe3", "cW3", "cb3",
         "aW1", "ab1", "ag1", "abe1", "aW2", "ab2", "ag2", "abe2", "aW3", "ab3"]


def _batch(rs, n, S, B=64):
    s = rs.normal(0, 1.5, size=(n, B, S)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(n, B, 1)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n, B))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(n, B, S)).astype(np.float32)
    return s, a, r, s2


@pytest.mark.parametrize("S,P,M", [(4, 6, 2), (3, 5, 3), (4, 1, 1), (4, 70, 5)])
def test_fused_set_learner_matches_oracle_on_concatenated_batch(S, P, M):
    """The mean over a set's P agents of their 64-row batch gradients == the gradient of the P*64-row batch (inference-mode
    BN: rows are independent). P = 70 with M = 5 gives workgroups more than one tile each and ragged tile counts."""
    need_gpu()
    B = 64
    conf, grp = _perturbed_group(M, S=S, seed=61)
    rs = np.random.RandomState(62)
    n = P * M  # agent v = p*M + m uses set m
    s, a, r, s2 = _batch(rs, n, S)
    losses = torch.zeros(M, 2, device="cuda")
    g = grp.learn_set_fused(t(s), t(a), t(r), t(s2), n, losses=losses)
    torch.cuda.synchronize()
    assert torch.isfinite(g).all()
    for k in range(M):
        sel = np.arange(P) * M + k
        cat = lambda x: x[sel].reshape(P * B, *x.shape[2:])
        cg, ag, aux = omlp.learn((cat(s), cat(a), cat(r)[:, None], cat(s2)), *_nets(grp, k, np.float64))
        gcg, gag = grp.grads_as_lists(g[k])
        for name, got, ref in zip(NAMES, gcg + gag, cg + ag):
            # bf16 second-layer operands (2^-9 relative rounding per element) against the float64 oracle: 2 % of the
            # tensor's max (the layer-wise bf16 learner's bound at these widths, tests/test_gpu_wide.py)
            assert _relerr(got, ref) <= 2e-2, (k, name, _relerr(got, ref))
        lo = losses[k].cpu().numpy()
        assert abs(lo[0] - aux["critic_loss"]) <= 1e-2 * abs(aux["critic_loss"])
        assert abs(lo[1] - aux["actor_loss"]) <= 1e-2 * max(1e-2, abs(aux["actor_loss"]))


@pytest.mark.parametrize("P,M", [(8, 3), (1, 1), (70, 5)])
def test_fused_set_learner_equals_per_agent_kernel_plus_federated_mean_and_layerwise_learner(P, M):
    """Same quantity three ways on the GPU: avd_learn_f32 per agent (exact f32 MFMA) + fed_mean over the platoons; the
    layer-wise bf16 GEMM chain on the set-major batch (wide.hip); the fused set learner on the agent-major batch. On weights
    with non-trivial BatchNorm / bias terms the fused learner must be as close to the exact result as the layer-wise one
    (a folded bias carried as ONE bf16 feature was not: 2.5x ... 19x the error -- it rides as a bf16 pair now)."""
    need_gpu()
    B, S = 64, 4
    conf, grp = _perturbed_group(M, S=S, seed=71)
    rs = np.random.RandomState(72)
    n = P * M
    s, a, r, s2 = _batch(rs, n, S)
    per_agent = grp.learn(t(s), t(a), t(r), t(s2), M)
    avg = vec.fed_mean(per_agent, P, M, method=conf.interfrl).cpu().numpy()  # [M, theta]
    sm = lambda x: t(np.ascontiguousarray(x.reshape(P, M, *x.shape[1:]).swapaxes(0, 1)).reshape(M, P * B, *x.shape[2:]))
    wide = grp.learn_shared(sm(s), sm(a), sm(r), sm(s2), n).cpu().numpy()
    fused = grp.learn_set_fused(t(s), t(a), t(r), t(s2), n).cpu().numpy()
    lay = grp.lay
    for name, lo, hi in (("actor", 0, lay.actor_size), ("critic", lay.actor_size, lay.theta_size)):
        scale = np.abs(avg[:, lo:hi]).max()
        ef, ew = np.abs(avg[:, lo:hi] - fused[:, lo:hi]).max() / scale, np.abs(avg[:, lo:hi] - wide[:, lo:hi]).max() / scale
        assert ef <= 2e-2 and np.abs(wide[:, lo:hi] - fused[:, lo:hi]).max() <= 2e-2 * scale, name
        assert ef <= 2.0 * ew + 5e-4, (name, ef, ew)  # measured: 0.9 ... 1.7 x the layer-wise learner's error
    # padding floats of the slab stay zero (what Adam relies on)
    assert fused[:, lay.actor_size - 3:lay.actor_size].max() == 0.0 or lay.actor_size % 4 == 0


def test_fused_set_learner_is_deterministic():
    """Per-workgroup partial sums combined in a fixed order: two runs are bit-identical (the layer-wise learner's split-K
    float atomics are not)."""
    need_gpu()
    P, M, S = 40, 5, 4
    conf, grp = _perturbed_group(M, S=S, seed=75)
    s, a, r, s2 = (t(x) for x in _batch(np.random.RandomState(76), P * M, S))
    g1 = grp.learn_set_fused(s, a, r, s2, P * M).clone()
    g2 = grp.learn_set_fused(s, a, r, s2, P * M).clone()
    assert torch.equal(g1, g2)


def test_fused_set_learner_weighted_mean_matches_weighted_fed_mean():
    """Server.get_weighted_avg_params (src/server/federated.py:99-118) through per-agent factors w_p * P / sum(w)."""
    need_gpu()
    P, M, B, S = 6, 2, 64, 4
    conf, grp = _perturbed_group(M, S=S, seed=101)
    rs = np.random.RandomState(102)
    n = P * M
    s, a, r, s2 = _batch(rs, n, S)
    for p in range(P):  # make the platoons' gradients differ, else any weighting gives the same mean
        r[p * M:(p + 1) * M] *= 1.0 + 4.0 * p
        s[p * M:(p + 1) * M] += 0.5 * p
    w = np.linspace(0.2, 3.0, P)[:, None].repeat(M, axis=1).astype(np.float32) * rs.uniform(0.8, 1.2, size=(P, M)).astype(np.float32)
    per_agent = grp.learn(t(s), t(a), t(r), t(s2), M)
    avg = vec.fed_mean(per_agent, P, M, weights=t(w), method=conf.interfrl).cpu().numpy()
    aw = (w * (P / w.sum(axis=0))).reshape(-1).astype(np.float32)  # agent-major [P*M]
    g = grp.learn_set_fused(t(s), t(a), t(r), t(s2), n, agent_weight=t(aw)).cpu().numpy()
    unweighted = grp.learn_set_fused(t(s), t(a), t(r), t(s2), n).cpu().numpy()
    lay = grp.lay
    for lo, hi in ((0, lay.actor_size), (lay.actor_size, lay.theta_size)):
        scale = np.abs(avg[:, lo:hi]).max()
        assert np.abs(avg[:, lo:hi] - g[:, lo:hi]).max() <= 2e-2 * scale
        assert np.abs(avg[:, lo:hi] - unweighted[:, lo:hi]).max() > 5e-2 * scale  # the weights matter in this case


def test_fused_set_learner_rejects_other_widths():
    from avddpg_amd import trainer

    need_gpu()
    with pytest.raises(ValueError, match="reference widths"):
        trainer.VecTrainer(config.Config(num_platoons=2, pl_size=2, fed_method="interfrl", weighted_average_enabled=False,
                                         actor_layer1_size=512, actor_layer2_size=256, critic_layer1_size=512,
                                         critic_layer2_size=256), rng="device", shared_engine="fused")
    conf = config.Config(actor_layer1_size=512, actor_layer2_size=256, critic_layer1_size=512, critic_layer2_size=256)
    grp = vec.AgentGroup(1, 4, 1, conf)
    z = torch.zeros(1, 64, 4, device="cuda")
    with pytest.raises(_hip.AvdError, match="reference widths"):
        grp.learn_set_fused(z, torch.zeros(1, 64, 1, device="cuda"), torch.zeros(1, 64, device="cuda"), z, 1)


def test_full_size_fused_set_learner_is_the_mean_of_its_halves_and_tracks_the_layerwise_learner():
    """BASELINE config 4 / configs[1] size (4096 platoons x 5 vehicle indices, 64-row batches): size-independent properties
    instead of the oracle -- the mean gradient over all 4096 platoons equals the average of the means over platoons
    [0, 2048) and [2048, 4096) (rows are independent; identical bf16 operand rounding on both sides, only the f32 summation
    grouping differs: 1e-3 of each block's max), and it agrees with the layer-wise bf16 learner to bf16 accuracy."""
    need_gpu()
    P, M, B, S = 4096, 5, 64, 4
    conf, grp = _perturbed_group(M, S=S, seed=91)
    gen = torch.Generator(device="cuda").manual_seed(92)
    rn = lambda *sh: torch.randn(*sh, device="cuda", generator=gen)
    n = P * M
    s, a, r, s2 = 1.5 * rn(n, B, S), 2.5 * (2 * torch.rand(n, B, 1, device="cuda", generator=gen) - 1), -rn(n, B).abs() * 0.3, 1.5 * rn(n, B, S)
    full = grp.learn_set_fused(s, a, r, s2, n).clone()
    h = n // 2
    halves = []
    for lo in (0, h):
        sl = lambda x: x[lo:lo + h].contiguous()
        halves.append(grp.learn_set_fused(sl(s), sl(a), sl(r), sl(s2), h).clone())
    avg = 0.5 * (halves[0] + halves[1])
    assert torch.isfinite(full).all() and full.abs().max() > 0
    sm = lambda x: x.view(P, M, *x.shape[1:]).transpose(0, 1).reshape(M, P * B, *x.shape[2:]).contiguous()
    wide = grp.learn_shared(sm(s), sm(a), sm(r), sm(s2), n)
    lay = grp.lay
    for lo, hi in ((0, lay.actor_size), (lay.actor_size, lay.theta_size)):
        scale = full[:, lo:hi].abs().max().item()
        assert (full[:, lo:hi] - avg[:, lo:hi]).abs().max().item() <= 1e-3 * scale, lo
        assert (full[:, lo:hi] - wide[:, lo:hi]).abs().max().item() <= 2e-2 * scale, lo
    assert not torch.allclose(halves[0], halves[1])  # the halves are different batches


def test_trainer_fused_engine_tracks_per_agent_engine_under_interfrl():
    """VecTrainer with shared weight sets: the fused set learner against the exact f32 per-agent kernel + fed_sum, same host
    RNG stream (parity mode). Before the first update the trajectories are identical; afterwards they differ by bf16 rounding
    of the gradients, amplified by Adam's normalisation (|dw| <= lr per step either way)."""
    from avddpg_amd import trainer

    need_gpu()
    P, L, steps = 6, 3, 72
    conf = config.Config(num_platoons=P, pl_size=L, buffer_size=128, fed_method="interfrl", weighted_average_enabled=False)
    runs = []
    for engine in ("per_agent", "fused"):
        np.random.seed(11)
        vt = trainer.VecTrainer(conf, rng="host", shared_sets=True, shared_engine=engine)
        vt.reset_episode()
        traj = []
        for i in range(steps):
            vt.step(0, i)
            traj.append((vt.actions.cpu().numpy().copy(), vt.env.x.cpu().numpy().copy()))
        runs.append((vt, traj))
    (a, ta), (b, tb) = runs
    assert b.grads is None and b.shared_engine == "fused" and a.updates == b.updates == (steps - 64) * P * L
    for i in range(steps):
        tol = 2e-6 if i < 65 else 5e-3  # (before the first update only the acting kernels differ: csrc/act.hip vs the rows kernel)
        assert np.abs(ta[i][0] - tb[i][0]).max() <= tol * 2.5, i
        assert np.abs(ta[i][1] - tb[i][1]).max() <= tol * max(1.0, np.abs(ta[i][1]).max()), i
    n_upd = steps - 64
    for lr, lo, hi in ((conf.actor_lr, 0, a.agents.lay.actor_size), (conf.critic_lr, a.agents.lay.actor_size, a.agents.lay.theta_size)):
        d = (a.agents.theta[:, lo:hi] - b.agents.theta[:, lo:hi]).abs()
        assert d.max().item() <= 2 * lr * n_upd and d.mean().item() <= 0.1 * lr * n_upd
    assert torch.isfinite(b.agents.theta).all() and int(b.agents.step[0]) == n_upd


def test_trainer_fused_engine_weighted_federation_runs_model_a():
    """Weighted interfrl (workers/trainer.py:385-398) + Model A (S = 3) through the fused engine: the weights of episode 3
    come from episodes 1-2 and reach the kernel as per-agent factors; finite, learning."""
    from avddpg_amd import trainer

    need_gpu()
    conf = config.Config(num_platoons=4, pl_size=2, buffer_size=128, fed_method="interfrl", weighted_average_enabled=True,
                         weighted_window=2, episode_sim_time=3.0, model="ModelA")  # 30-step episodes
    np.random.seed(2)
    vt = trainer.VecTrainer(conf, rng="host", shared_engine="fused")
    assert vt.shared and vt.shared_engine == "fused" and vt.agents.lay.S == 3
    th0 = vt.agents.theta.clone()
    vt.run(number_of_episodes=4)
    assert vt.fed_weights is not None and vt.fed_weights[0] == 3
    w = vt.fed_weights[1].cpu().numpy()
    ref = np.array([[abs(1 / np.mean(vt.all_ep_reward_lists[p][m][-2 - 1:-1])) for m in range(2)] for p in range(4)])
    assert np.allclose(w, ref, rtol=1e-6)
    assert torch.isfinite(vt.agents.theta).all() and int(vt.agents.step[0]) == 4 * 30 - 64
    assert not torch.equal(vt.agents.theta, th0)


def test_mfma_layout_probe_passes_on_this_gpu():
    """The three hardware layouts csrc/fset.hip is built on -- v_mfma_f32_32x32x16_bf16 lane maps, an accumulator tile as the
    next MFMA's A operand (permuted k order), ds_read_b64_tr_b16 as the B operand of that product -- checked with exact
    small-integer data by tools/probes/mfma_layout.hip on the GPU the tests run on."""
    import os
    import subprocess

    need_gpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src, exe = os.path.join(root, "tools", "probes", "mfma_layout.hip"), os.path.join(root, "tools", "probes", "mfma_layout")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-Wno-unused-value", src, "-o", exe], check=True,
                       capture_output=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.count("PASS") == 3 and "FAIL" not in out.stdout, out.stdout + out.stderr
