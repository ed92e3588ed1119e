"""GPU parity of the shared-weight-set learner (csrc/wide.hip): the bf16 MFMA GEMM on its own, and Trainer.learn +
federated mean over P agents that share one weight set, against (a) the oracle on the concatenated batch and (b) the
per-agent f32 kernel followed by the federated mean kernels. Tolerances are bf16 ones (8 significant bits on the GEMM
operands, f32 accumulation) and are written next to each assertion."""
import numpy as np
import pytest
import torch

from avddpg_amd import _hip, config, vec
from avddpg_amd._hip import call, ptr, stream_handle
from oracle import mlp as omlp
from tests.gpu_util import need_gpu, t
from tests.test_gpu_mlp import _nets, _perturbed_group, _relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,Nc,K", [(128, 128, 64), (200, 72, 192), (1024, 304, 1088), (64, 1024, 4096), (1000, 1072, 1024),
                                    (2048, 512, 64)])
def test_bf16_gemm_matches_float64_on_the_same_bf16_operands(M, Nc, K):
    need_gpu()
    rs = np.random.RandomState(M + Nc + K)
    Mp, Np_ = (M + 255) // 256 * 256, (Nc + 255) // 256 * 256  # operands readable to the next tile multiple
    A = torch.zeros(Mp, K, dtype=torch.bfloat16, device="cuda")
    B = torch.zeros(Np_, K, dtype=torch.bfloat16, device="cuda")
    A[:M] = t(rs.normal(0, 1, (M, K)).astype(np.float32)).to(torch.bfloat16)
    B[:Nc] = t(rs.normal(0, 1, (Nc, K)).astype(np.float32)).to(torch.bfloat16)
    ldd = (Nc + 3) // 4 * 4
    D = torch.full((M, ldd), float("nan"), device="cuda")
    call("avd_gemm_bt_bf16", M, Nc, K, ptr(A), K, ptr(B), K, ptr(D), ldd, stream_handle())
    ref = A[:M].double().cpu().numpy() @ B[:Nc].double().cpu().numpy().T
    got = D[:, :Nc].cpu().numpy()
    # identical bf16 operands, f32 accumulation in a different order: 1e-5 of the result scale sqrt(K)
    assert np.max(np.abs(got - ref)) <= 1e-5 * np.sqrt(K) * 4, np.max(np.abs(got - ref))
    if ldd > Nc:
        assert torch.isnan(D[:, Nc:]).all()  # nothing stored outside the matrix


@pytest.mark.parametrize("S,P,widths,tol", [(4, 6, (256, 128, 48), 2e-2), (3, 4, (128, 64, 32), 2e-2), (3, 5, (64, 64, 16), 2e-2),
                                            (4, 8, (1024, 1024, 48), 8e-2),
                                            # the rank-one backward (r06) with the critic's action features through the 64-column GEMM (its row
                                            # factor in the epilogue; dx_gen_kernel<true> serves H1 = 1024 only), one and two 512-column blocks
                                            # (H1 = 512 / 256: four / two state feature blocks share a chunk's rows for the S2 sum)
                                            (4, 8, (512, 512, 48), 8e-2), (4, 8, (512, 1024, 32), 8e-2), (4, 8, (256, 512, 48), 8e-2),
                                            # fused forward WITHOUT the rank-one backward (H1 = 128 is no multiple of 256): fwd_gen_kernel's run-time
                                            # epilogue (relu'd / signed activations out), fwd_delta_kernel on stored activations, layer-wise backward
                                            (4, 8, (128, 512, 32), 8e-2)])
def test_shared_learner_matches_oracle_on_concatenated_batch(S, P, widths, tol):
    """n_sets = 2 weight sets, P agents each: the mean of the agents' gradients == the gradient of the P*64-row batch."""
    need_gpu()
    n_sets, B = 2, 64
    H1, H2, Ha = widths
    conf, grp = _perturbed_group(n_sets, S=S, seed=61, actor_layer1_size=H1, actor_layer2_size=H2, critic_layer1_size=H1,
                                 critic_layer2_size=H2, critic_act_layer_size=Ha)
    rs = np.random.RandomState(62)
    rows = P * B
    s = rs.normal(0, 1.5, size=(n_sets, rows, S)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(n_sets, rows, 1)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n_sets, rows))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(n_sets, rows, S)).astype(np.float32)
    losses = torch.zeros(n_sets, 2, device="cuda")
    g = grp.learn_shared(t(s), t(a), t(r), t(s2), n_sets * P, losses=losses)
    torch.cuda.synchronize()
    for k in range(n_sets):
        cg, ag, aux = omlp.learn((s[k], a[k], r[k][:, None], s2[k]), *_nets(grp, k, np.float64))
        gcg, gag = grp.grads_as_lists(g[k])
        names = ["cWs", "cbs", "cWa", "cba", "cgs", "cbes", "cga", "cbea", "cW2", "cb2", "cg3", "cbe3", "cW3", "cb3",
                 "aW1", "ab1", "ag1", "abe1", "aW2", "ab2", "ag2", "abe2", "aW3", "ab3"]
        for name, got, ref in zip(names, gcg + gag, cg + ag):
            # bf16 GEMM operands (2^-9 relative rounding per element) against the float64 oracle: 2 % of the tensor's max
            # at the reference widths; at BASELINE config 5's hidden = 1024 the first-layer actor gradients sit behind
            # five bf16 GEMMs with 1024-long reductions and only 512 rows to average over here: 8 %
            assert _relerr(got, ref) <= tol, (k, name, _relerr(got, ref))
        lo = losses[k].cpu().numpy()
        assert abs(lo[0] - aux["critic_loss"]) <= 1e-2 * abs(aux["critic_loss"])
        assert abs(lo[1] - aux["actor_loss"]) <= 1e-2 * max(1e-2, abs(aux["actor_loss"]))


def test_shared_learner_equals_per_agent_kernel_plus_federated_mean():
    """Same quantity two ways on the GPU: avd_learn_f32 per agent (f32 MFMA) + fed_mean over the platoons, and the
    shared-set learner on the set-major batch."""
    need_gpu()
    P, M, B, S = 8, 3, 64, 4
    conf, grp = _perturbed_group(M, S=S, seed=71)
    rs = np.random.RandomState(72)
    n = P * M  # agent v = p*M + m uses set m
    s = rs.normal(0, 1.5, size=(n, B, S)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(n, B, 1)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n, B))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(n, B, S)).astype(np.float32)
    per_agent = grp.learn(t(s), t(a), t(r), t(s2), M)
    avg = vec.fed_mean(per_agent, P, M, method=conf.interfrl)  # [M, theta]
    sm = lambda x: t(np.ascontiguousarray(x.reshape(P, M, *x.shape[1:]).swapaxes(0, 1)).reshape(M, P * B, *x.shape[2:]))
    g = grp.learn_shared(sm(s), sm(a), sm(r), sm(s2), n)
    a_, b_ = avg.cpu().numpy(), g.cpu().numpy()
    lay = grp.lay
    for name, lo, hi in (("actor", 0, lay.actor_size), ("critic", lay.actor_size, lay.theta_size)):
        d = np.abs(a_[:, lo:hi] - b_[:, lo:hi]).max()
        assert d <= 2e-2 * np.abs(a_[:, lo:hi]).max(), (name, d, np.abs(a_[:, lo:hi]).max())


def test_shared_learner_rejects_unsupported_shapes():
    need_gpu()
    conf = config.Config(actor_layer1_size=320, actor_layer2_size=160, critic_layer1_size=320, critic_layer2_size=160)
    grp = vec.AgentGroup(1, 4, 1, conf)
    z = torch.zeros(1, 64, 4, device="cuda")
    with pytest.raises(_hip.AvdError, match="multiples of 64"):
        grp.learn_shared(z, torch.zeros(1, 64, 1, device="cuda"), torch.zeros(1, 64, device="cuda"), z, 1)


def test_trainer_batched_engine_tracks_per_agent_engine_under_interfrl():
    """VecTrainer with shared weight sets: the batched bf16 learner against the exact f32 per-agent kernel + fed_sum,
    same host RNG stream. Before the first update the trajectories are identical; afterwards they differ by bf16
    rounding of the gradients, amplified by Adam's normalisation (|dw| <= lr per step either way)."""
    from avddpg_amd import trainer

    need_gpu()
    P, L, steps = 6, 3, 72
    conf = config.Config(num_platoons=P, pl_size=L, buffer_size=128, fed_method="interfrl", weighted_average_enabled=False)
    runs = []
    for engine in ("per_agent", "batched"):
        np.random.seed(11)
        vt = trainer.VecTrainer(conf, rng="host", shared_sets=True, shared_engine=engine)
        vt.reset_episode()
        traj = []
        for i in range(steps):
            vt.step(0, i)
            traj.append((vt.actions.cpu().numpy().copy(), vt.env.x.cpu().numpy().copy()))
        runs.append((vt, traj))
    (a, ta), (b, tb) = runs
    assert b.grads is None and b.shared_engine == "batched" and a.updates == b.updates == (steps - 64) * P * L
    for i in range(steps):
        tol = 0.0 if i < 65 else 5e-3
        assert np.abs(ta[i][0] - tb[i][0]).max() <= tol * 2.5, i
        assert np.abs(ta[i][1] - tb[i][1]).max() <= tol * max(1.0, np.abs(ta[i][1]).max()), i
    n_upd = steps - 64
    for lr, lo, hi in ((conf.actor_lr, 0, a.agents.lay.actor_size), (conf.critic_lr, a.agents.lay.actor_size, a.agents.lay.theta_size)):
        d = (a.agents.theta[:, lo:hi] - b.agents.theta[:, lo:hi]).abs()
        assert d.max().item() <= 2 * lr * n_upd and d.mean().item() <= 0.1 * lr * n_upd
    assert torch.isfinite(b.agents.theta).all() and int(b.agents.step[0]) == n_upd


def test_trainer_runs_hidden_1024_with_shared_sets():
    """BASELINE config 5's shape (actor/critic hidden = 1024) at a small platoon count: only the batched engine exists
    for it (the LDS-resident kernel refuses H2 > 256 loudly)."""
    from avddpg_amd import trainer

    need_gpu()
    conf = config.Config(num_platoons=8, pl_size=5, buffer_size=128, fed_method="interfrl", weighted_average_enabled=False,
                         actor_layer1_size=1024, actor_layer2_size=1024, critic_layer1_size=1024, critic_layer2_size=1024)
    vt = trainer.VecTrainer(conf, rng="device", auto_reset=True)
    assert vt.shared and vt.shared_engine == "batched" and vt.agents.n_sets == 5
    th0 = vt.agents.theta.clone()
    vt.reset_episode()
    for _ in range(68):
        vt.step()
    torch.cuda.synchronize()
    assert vt.updates == 4 * 40 and torch.isfinite(vt.agents.theta).all() and torch.isfinite(vt.env.x).all()
    assert not torch.equal(vt.agents.theta, th0) and (vt.set_losses[:, 0] >= 0).all()
    with pytest.raises(ValueError, match="shared weight sets"):
        trainer.VecTrainer(config.Config(num_platoons=2, pl_size=2), rng="device", shared_engine="batched")


def test_shared_actor_forward_matches_per_agent_rows_kernel():
    """Acting with shared sets as a GEMM chain (bf16 operands) vs the f32 batch-1 kernel: tanh(.)*high within 2e-2 * high."""
    need_gpu()
    P, M, S = 37, 3, 4
    conf, grp = _perturbed_group(M, S=S, seed=81, actor_layer1_size=512, actor_layer2_size=320, critic_layer1_size=512,
                                 critic_layer2_size=320)
    rs = np.random.RandomState(82)
    x = rs.normal(0, 1.5, size=(P * M, S)).astype(np.float32)  # agent v = p*M + m
    ref = grp.actor(t(x), set_mod=M).cpu().numpy()
    sm = t(np.ascontiguousarray(x.reshape(P, M, S).swapaxes(0, 1)))
    got = grp.actor_shared(sm, P * M).cpu().numpy().T.reshape(-1)
    assert np.abs(ref).max() > 0.05 and np.abs(got - ref).max() <= 2e-2 * 2.5


def test_full_size_shared_learner_is_the_mean_of_its_halves():
    """BASELINE config 4 / configs[1] size (4096 platoons x 5 vehicle indices, 64-row batches): a size-independent
    property instead of the oracle -- the mean gradient over all 4096 platoons' rows equals the average of the mean
    gradients over platoons [0, 2048) and [2048, 4096) (every row's contribution is independent of the other rows:
    inference-mode BN, no batch statistics). Identical bf16 operand rounding on both sides; only the f32 summation
    order (split-K atomics) differs: 1e-3 of each slab's max."""
    need_gpu()
    P, M, B, S = 4096, 5, 64, 4
    conf, grp = _perturbed_group(M, S=S, seed=91)
    g = torch.Generator(device="cuda").manual_seed(92)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    s, a, r, s2 = 1.5 * rn(M, P * B, S), 2.5 * (2 * torch.rand(M, P * B, 1, device="cuda", generator=g) - 1), -rn(M, P * B).abs() * 0.3, 1.5 * rn(M, P * B, S)
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
    assert not torch.allclose(halves[0], halves[1])  # the halves are different batches


def test_shared_learner_weighted_mean_matches_weighted_fed_mean():
    """Server.get_weighted_avg_params (src/server/federated.py:99-118) through per-row weights w_p * P / sum(w)."""
    need_gpu()
    P, M, B, S = 6, 2, 64, 4
    conf, grp = _perturbed_group(M, S=S, seed=101)
    rs = np.random.RandomState(102)
    n = P * M
    s = rs.normal(0, 1.5, size=(n, B, S)).astype(np.float32)
    a = rs.uniform(-2.5, 2.5, size=(n, B, 1)).astype(np.float32)
    r = -np.abs(rs.normal(0, 0.3, size=(n, B))).astype(np.float32)
    s2 = rs.normal(0, 1.5, size=(n, B, S)).astype(np.float32)
    # make the platoons' gradients differ (else any weighting gives the same mean): platoon-dependent reward scale / state offset
    for p in range(P):
        r[p * M:(p + 1) * M] *= 1.0 + 4.0 * p
        s[p * M:(p + 1) * M] += 0.5 * p
    w = np.linspace(0.2, 3.0, P)[:, None].repeat(M, axis=1).astype(np.float32) * rs.uniform(0.8, 1.2, size=(P, M)).astype(np.float32)
    per_agent = grp.learn(t(s), t(a), t(r), t(s2), M)
    avg = vec.fed_mean(per_agent, P, M, weights=t(w), method=conf.interfrl).cpu().numpy()
    sm = lambda x: t(np.ascontiguousarray(x.reshape(P, M, *x.shape[1:]).swapaxes(0, 1)).reshape(M, P * B, *x.shape[2:]))
    rw = np.repeat((w * (P / w.sum(axis=0))).T[:, :, None], B, axis=2).reshape(M, P * B).astype(np.float32)
    g = grp.learn_shared(sm(s), sm(a), sm(r), sm(s2), n, row_weight=t(rw)).cpu().numpy()
    unweighted = grp.learn_shared(sm(s), sm(a), sm(r), sm(s2), n).cpu().numpy()
    lay = grp.lay
    for lo, hi in ((0, lay.actor_size), (lay.actor_size, lay.theta_size)):
        scale = np.abs(avg[:, lo:hi]).max()
        assert np.abs(avg[:, lo:hi] - g[:, lo:hi]).max() <= 2e-2 * scale
        assert np.abs(avg[:, lo:hi] - unweighted[:, lo:hi]).max() > 5e-2 * scale  # the weights matter in this case


_DIAG_LIB = _hip.DIAG_LIB_PATH
_FWD_AB_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from tests.gpu_util import t
from tests.test_gpu_mlp import _perturbed_group
n_sets, P, S = 2, {"small": 3, "padded": 15}.get(sys.argv[3] if len(sys.argv) > 3 else "", 12), 4
conf, grp = _perturbed_group(n_sets, S=S, seed=91, actor_layer1_size=1024, actor_layer2_size=1024, critic_layer1_size=1024,
                             critic_layer2_size=1024)
rs = np.random.RandomState(92)
rows = P * 64
s = rs.normal(0, 1.5, size=(n_sets, rows, S)).astype(np.float32)
a = rs.uniform(-2.5, 2.5, size=(n_sets, rows, 1)).astype(np.float32)
r = -np.abs(rs.normal(0, 0.3, size=(n_sets, rows))).astype(np.float32)
s2 = rs.normal(0, 1.5, size=(n_sets, rows, S)).astype(np.float32)
rw = None
if len(sys.argv) > 3 and sys.argv[3] == "weighted":  # per-platoon weights w_p * P / sum(w) on the platoon's 64 rows
    w = rs.uniform(0.3, 2.5, size=(n_sets, P)).astype(np.float32)
    rw = t(np.repeat(w * (P / w.sum(axis=1, keepdims=True)), 64, axis=1).astype(np.float32))
losses = torch.zeros(n_sets, 2, device="cuda")
g = grp.learn_shared(t(s), t(a), t(r), t(s2), n_sets * P, losses=losses, row_weight=rw)
torch.cuda.synchronize()
np.save(sys.argv[2], np.concatenate([g.cpu().numpy().ravel(), losses.cpu().numpy().ravel()]))
"""


# small: 192 rows per set (256 padded): forward fused, gradients layer-wise; padded: 960 rows per set (1024 padded), every fused kernel
@pytest.mark.parametrize("mode", ["plain", "weighted", "small", "padded"])
def test_fused_forward_passes_match_the_layerwise_forward_at_hidden_1024(tmp_path, mode):
    """csrc/wide.hip, fw::fwd_gen_kernel (first layer generated on the matrix cores as the GEMM operand, W2 streamed through
    LDS, output layer in the kernel, the mu pass's dZ2 written by the forward kernel) and fw::dw_gen_kernel (the weight gradient
    in the same style) against the layer-wise l1_fwd + GEMM + bias + out_bwd kernels they replace (AVD_WIDE_FUSED_FWD=0),
    the whole learn chain on the same inputs in two processes (the switch is read once per process). Since r06 the fused backward
    is the rank-one form (dZ2 = d (x) cf (.) mask never materialised: the operands are the exact mask and bf16(d y1), bf16(cf W2),
    where the layer-wise path rounds bf16(d cf) and bf16(y1) separately): independent 2^-9 roundings per term of a 768-row sum,
    measured 4.3e-3 of a block's max in the weighted case; since r06b critic(s, a) and critic(s, mu) are ONE forward pass whose action
    gradient comes from the f32 accumulators (cf as a bf16 pair on the mask operand, inv (.) W2 on the other) where the layer-wise path
    goes through a stored bf16 gradient matrix: 8.3e-3 measured (padded case), 1.2e-2 allowed -- where the bf16 operands themselves
    cost 8e-2 against the float64 oracle (tests above and tests/test_gpu_configs_full.py hold BOTH paths to the oracle)."""
    import os
    import subprocess
    import sys

    need_gpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ab.py"
    script.write_text(_FWD_AB_SCRIPT)
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, AVD_WIDE_FUSED_FWD=flag, AVDDPG_HIP_LIB=_DIAG_LIB)  # (the switch exists in the diagnostic build only)
        out = tmp_path / f"g{flag}.npy"
        p = subprocess.run([sys.executable, str(script), root, str(out), mode], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(np.load(out))
    fused, layerwise = outs
    assert np.isfinite(fused).all() and np.abs(layerwise).max() > 0
    n = fused.size - 4
    lay = vec.AgentGroup(2, 4, 1, config.Config(actor_layer1_size=1024, actor_layer2_size=1024, critic_layer1_size=1024,
                                                  critic_layer2_size=1024)).lay
    gf, gl = fused[:n].reshape(2, -1), layerwise[:n].reshape(2, -1)
    for lo, hi in ((0, lay.actor_size), (lay.actor_size, lay.theta_size)):
        scale = np.abs(gl[:, lo:hi]).max()
        assert np.abs(gf[:, lo:hi] - gl[:, lo:hi]).max() <= 1.2e-2 * scale, (lo, np.abs(gf[:, lo:hi] - gl[:, lo:hi]).max() / scale)
    assert np.allclose(fused[n:], layerwise[n:], rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("switch", ["AVD_WIDE_FUSED_DELTA", "AVD_WIDE_DUAL"])
def test_critic_of_mu_as_a_delta_matches_the_full_forward_pass(tmp_path, switch):
    """AVD_WIDE_DUAL (r06b): the product path -- critic(s, a) and critic(s, mu) in ONE forward pass, fw::fwd_gen_kernel<true, 4>: the delta on
    the f32 accumulators, the action gradient from transposed reads of the streamed action chunks -- against the two-kernel form it
    replaced (=0: the forward pass that stores signed activations + fw::fwd_delta_kernel, still the path of shapes the one-pass form
    does not take); the critic's gradients come from the same rank-one backward in both: only the mask's producer differs.
    AVD_WIDE_FUSED_DELTA: csrc/wide.hip, fw::fwd_delta_kernel: critic(s, mu) = critic(s, a) + W2[action rows] (f(mu) - f(a)) on the activations pass 1
    stored, the action gradient from the same kernel (the dZ2 tile as the next product's operand) -- against the full fused forward
    pass + the input-gradient GEMM of the action columns + the row dot it replaces (AVD_WIDE_FUSED_DELTA=0), same inputs, two
    processes. Only the actor gradient and the two losses depend on the pass (`workers/trainer.py:502-506`); the stored
    activations cost one more bf16 rounding of z2(a) before the delta is added. Since r06 switching the delta pass off also switches
    the rank-one backward off (the delta pass leaves critic(s, a)'s relu mask for it), so the pair differs like the forward pair
    above: 6e-3 of a block's max (measured 4.3e-3)."""
    import os
    import subprocess
    import sys

    need_gpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ab.py"
    script.write_text(_FWD_AB_SCRIPT)
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, AVDDPG_HIP_LIB=_DIAG_LIB, **{switch: flag})
        out = tmp_path / f"d{flag}.npy"
        p = subprocess.run([sys.executable, str(script), root, str(out), "weighted"], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(np.load(out))
    delta, full = outs
    assert np.isfinite(delta).all() and np.abs(full).max() > 0
    n = delta.size - 4
    lay = vec.AgentGroup(2, 4, 1, config.Config(actor_layer1_size=1024, actor_layer2_size=1024, critic_layer1_size=1024,
                                                  critic_layer2_size=1024)).lay
    gd, gf = delta[:n].reshape(2, -1), full[:n].reshape(2, -1)
    assert np.abs(gd[:, :lay.actor_size] - gf[:, :lay.actor_size]).max() > 0  # (the switch did switch)
    for lo, hi in ((0, lay.actor_size), (lay.actor_size, lay.theta_size)):
        scale = np.abs(gf[:, lo:hi]).max()
        assert np.abs(gd[:, lo:hi] - gf[:, lo:hi]).max() <= 1.2e-2 * scale, (lo, np.abs(gd[:, lo:hi] - gf[:, lo:hi]).max() / scale)
    assert np.allclose(delta[n:], full[n:], rtol=2e-3, atol=1e-6)


def test_one_pass_critic_action_gradient_agrees_with_the_two_kernel_form_row_by_row(tmp_path):
    """csrc/wide.hip, fw::fwd_gen_kernel<true, 4> (critic(s, a) and critic(s, mu) in one forward pass): q(s, mu) and the action gradient
    of EVERY row against the two-kernel form (AVD_WIDE_DUAL=0: stored signed activations + fw::fwd_delta_kernel), 4096 rows per set --
    the check that caught, while the kernel was written, a missing barrier at the end of its tail, an inline-asm relu on an MFMA
    result and first-layer masks the scheduler had deferred (whole row tiles wrong, tensor-level tolerances still met). Diagnostic
    library: AVD_WIDE_DUMP writes q(s, mu), the action gradient and mu of the call. Both forms round differently (the one-pass form
    continues from f32 accumulators, cf as a bf16 pair on the mask operand), so rows whose first-layer pre-activation sits on the relu
    kink may flip either way: measured 11 / 12 rows of 4096 beyond 2 % of the largest gradient, median ratio 0.985 ... 0.997 -- allowed:
    1 % of the rows, median within 3 %, q(s, mu) within 1.5 % of max |q|."""
    import os
    import subprocess
    import sys

    need_gpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rows.py"
    script.write_text(_FWD_AB_SCRIPT.replace('{"small": 3, "padded": 15}.get(sys.argv[3] if len(sys.argv) > 3 else "", 12)', "64"))
    dumps = []
    for flag in ("0", "1"):
        out, dump = tmp_path / f"g{flag}.npy", tmp_path / f"da{flag}.bin"
        env = dict(os.environ, AVDDPG_HIP_LIB=_DIAG_LIB, AVD_WIDE_DUAL=flag, AVD_WIDE_DUMP=str(dump))
        p = subprocess.run([sys.executable, str(script), root, str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        dumps.append(np.fromfile(dump, np.float32).reshape(3, 2, -1))  # [q(s, mu), da, mu][set][Np]
    two, one = dumps
    assert np.array_equal(two[2], one[2])  # the same mu went into both (the actor's pass is the same kernel)
    for k in range(2):
        q_max = np.abs(two[0, k]).max()  # (the two-kernel form continues from bf16(z2(a)): 2^-9 per element; measured 8e-3 of max |q|)
        assert np.abs(one[0, k] - two[0, k]).max() <= 1.5e-2 * q_max, (k, np.abs(one[0, k] - two[0, k]).max() / q_max)
        sc = np.abs(two[1, k]).max()
        off = np.abs(one[1, k] - two[1, k]) > 2e-2 * sc
        assert off.mean() <= 1e-2, (k, int(off.sum()), np.nonzero(off)[0][:32])
        big = np.abs(two[1, k]) > 0.1 * sc
        assert abs(np.median(one[1, k][big] / two[1, k][big]) - 1) <= 3e-2, (k, np.median(one[1, k][big] / two[1, k][big]))

