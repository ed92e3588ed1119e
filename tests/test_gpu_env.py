"""GPU parity: platoon step / reset / OU noise / policy kernels vs the oracle and the golden
vectors captured from the reference. Everything goes through the C ABI (avddpg_amd.vec)."""
import os

import numpy as np
import pytest
import torch

from avddpg_amd import vec
from oracle import noise as onoise
from oracle import platoon
from tests.gpu_util import conf_and_ep, golden_case_kwargs, need_gpu, t

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
REL = 1e-5  # north_star: float32 states/rewards within 1e-5 relative of the reference CPU path


def _set_state(env, x, prev_a):
    env.x.copy_(t(x))
    env.prev_a.copy_(t(prev_a))
    if env.cum_accel is not None:
        env.cum_accel.zero_()


def test_step_matches_reference_golden_traces():
    """Teacher-forced per step from the reference's float64 trajectory (G3): every golden case,
    incl. Model A/B, euler/exact, L=3/5/10, forced terminal, can_terminate off, non-degenerate taus,
    centralized mean reward."""
    need_gpu()
    g = np.load(os.path.join(G, "g3_step.npz"))
    for key in g["keys"]:
        conf, ep = conf_and_ep(**golden_case_kwargs(key))
        xs = np.concatenate([g[key + "__x0"][None], g[key + "__x"]], axis=0)
        K, L = g[key + "__actions"].shape
        np.random.seed(0)
        env = vec.VecPlatoon(1, L, conf, track_aux=True)
        pa = g[key + "__prev_a0"].copy()
        cum = np.zeros(L)
        for k in range(K):
            _set_state(env, xs[k][None], pa[None])
            env.cum_accel.copy_(t(cum[None]))
            obs, rew, done = env.step(t(g[key + "__actions"][k][None]), t(g[key + "__exog"][k][None]))
            x = env.x.cpu().numpy()[0].astype(np.float64)
            scale = np.maximum(1.0, np.abs(g[key + "__x"][k]))
            assert np.all(np.abs(x - g[key + "__x"][k]) <= REL * scale), (key, k)
            if conf.framework == conf.cntrl:
                assert abs(env.reward_mean.item() - g[key + "__rewards"][k][0]) <= REL * max(1, abs(g[key + "__rewards"][k][0]))
            else:
                assert np.allclose(rew.cpu().numpy()[0], g[key + "__rewards"][k], rtol=REL, atol=1e-7), (key, k)
                assert np.array_equal(obs.cpu().numpy()[0], env.x.cpu().numpy()[0][:, :env.obs_width])
            assert bool(done.item()) == bool(g[key + "__done"][k]) == bool(env.any_done.item())
            env.any_done.zero_()
            cum = cum + xs[k][:, 2]
            vel = env.cum_accel.cpu().numpy()[0] * conf.sample_rate
            assert np.allclose(vel, g[key + "__velocity"][k], rtol=REL, atol=1e-6)
            jerk = env.get_jerk_from(t(xs[k][None]), t(pa[None])).cpu().numpy()[0]
            assert np.allclose(jerk, g[key + "__jerk"][k], rtol=1e-4, atol=1e-5)
            pa = xs[k][:, 2]


def test_free_running_trajectory_stays_within_tolerance():
    """64 un-forced steps on the GPU vs the reference float64 trajectory."""
    need_gpu()
    g = np.load(os.path.join(G, "g3_step.npz"))
    for key in ("L5_ModelB_euler", "L5_ModelA_exact", "L10_ModelB_exact", "nondegenerate_L5_A"):
        conf, _ = conf_and_ep(**golden_case_kwargs(key))
        K, L = g[key + "__actions"].shape
        np.random.seed(0)
        env = vec.VecPlatoon(1, L, conf)
        _set_state(env, g[key + "__x0"][None], g[key + "__prev_a0"][None])
        for k in range(K):
            env.step(t(g[key + "__actions"][k][None]), t(g[key + "__exog"][k][None]))
        x = env.x.cpu().numpy()[0]
        scale = np.maximum(1.0, np.abs(g[key + "__x"][-1]))
        assert np.all(np.abs(x - g[key + "__x"][-1]) <= 64 * REL * scale), key


@pytest.mark.parametrize("model,method,L", [("ModelB", "euler", 5), ("ModelA", "exact", 10), ("ModelB", "exact", 3),
                                            ("ModelA", "euler", 16), ("ModelB", "euler", 1)])
def test_step_bit_exact_vs_f32_oracle_and_lane_independence(model, method, L):
    """At full batch (P=4096): the kernel equals the float32 oracle BIT FOR BIT (same operation
    order, no FMA contraction), and lane p of the batch equals a P=1 launch on the same platoon."""
    need_gpu()
    P = 4096
    conf, ep = conf_and_ep(model=model, method=method)
    rs = np.random.RandomState(7)
    x = rs.normal(0, 8.0, size=(P, L, 4)).astype(np.float32)  # wide enough that some vehicles terminate
    pa = rs.normal(0, 0.1, size=(P, L)).astype(np.float32)
    u = rs.uniform(-2.5, 2.5, size=(P, L)).astype(np.float32)
    ex = rs.normal(0, 0.1, size=P).astype(np.float32)
    np.random.seed(0)
    env = vec.VecPlatoon(P, L, conf, rng="device", track_aux=True)
    _set_state(env, x, pa)
    env.step(t(u), t(ex))
    o = platoon.batched_step(ep, x, pa, np.zeros((P, L), np.float32), u, ex, dtype=np.float32)
    assert np.array_equal(env.x.cpu().numpy(), o["x"])
    assert np.array_equal(env.reward.cpu().numpy(), o["reward"])
    assert np.array_equal(env.prev_a.cpu().numpy(), o["prev_a"])
    assert np.array_equal(env.cum_accel.cpu().numpy(), o["cum_accel"])
    assert np.array_equal(env.term.cpu().numpy().astype(bool), o["term"])
    assert np.array_equal(env.done.cpu().numpy().astype(bool), o["done"])
    assert 0 < o["done"].sum() < P and env.any_done.item() == 1
    assert np.array_equal(env.x_prev.cpu().numpy(), x)  # pre-step state kept for the replay add
    for p in (0, 51, 52, 4095):  # block boundaries at 256//L platoons
        one = vec.VecPlatoon(1, L, conf, rng="device")
        _set_state(one, x[p:p + 1], pa[p:p + 1])
        one.step(t(u[p:p + 1]), t(ex[p:p + 1]))
        assert np.array_equal(one.x.cpu().numpy()[0], env.x.cpu().numpy()[p])
        assert np.array_equal(one.reward.cpu().numpy()[0], env.reward.cpu().numpy()[p])


def test_centralized_mean_reward_bit_exact():
    need_gpu()
    conf, ep = conf_and_ep(framework="centralized")
    rs = np.random.RandomState(1)
    P, L = 300, 5
    x = rs.normal(0, 2.0, size=(P, L, 4)).astype(np.float32)
    pa = x[..., 2].copy()
    u = rs.uniform(-2.5, 2.5, size=(P, L)).astype(np.float32)
    ex = rs.normal(0, 0.1, size=P).astype(np.float32)
    env = vec.VecPlatoon(P, L, conf, rng="device")
    assert (env.num_states, env.num_actions, env.num_models, env.hidden_multiplier) == (20, 5, 1, 1.2)
    _set_state(env, x, pa)
    env.step(t(u), t(ex))
    o = platoon.batched_step(ep, x, pa, np.zeros((P, L), np.float32), u, ex, dtype=np.float32)
    assert np.array_equal(env.reward_mean.cpu().numpy(), o["reward_mean"])


def test_reset_host_rng_reproduces_reference_states_and_stream():
    """Host-RNG mode: fixed-seed reset states equal the reference's (G2), and the global RNG ends
    at the same position (ctor 2+3L draws, reset 1+4L draws per platoon)."""
    need_gpu()
    g = np.load(os.path.join(G, "g2_reset.npz"))
    for key in g["keys"]:
        seed, L, rand_gen, model, mode = key.split("_")
        L = int(L[1:])
        conf, _ = conf_and_ep(model=model, rand_gen=rand_gen)
        np.random.seed(int(seed[1:]))
        env = vec.VecPlatoon(1, L, conf, rand_states=(mode != "fixed"), evaluator_states_enabled=(mode == "evaluator"))
        obs = env.reset()
        x = env.x.cpu().numpy()[0]
        assert np.array_equal(x, g[key + "__reset_x"].astype(np.float32)), key
        assert np.array_equal(env.prev_a.cpu().numpy()[0], x[:, 2])
        assert obs.shape == (1, L, 3 if model == "ModelA" else 4)
        assert np.random.normal(0, 1) == float(g[key + "__next_normal"]), key


def test_reset_device_rng_distribution_and_chain():
    need_gpu()
    conf, _ = conf_and_ep()
    env = vec.VecPlatoon(4096, 5, conf, rng="device", seed=3)
    env.reset()
    x = env.x.cpu().numpy()
    for col, std in ((0, 1.5), (1, 1.5), (2, 0.05)):
        v = x[..., col].ravel()
        assert abs(v.mean()) < 4 * std / np.sqrt(v.size) and abs(v.std() / std - 1) < 0.03
    assert np.array_equal(x[:, 1:, 3], x[:, :-1, 2]) and np.all(x[:, 0, 3] == 0)  # chain + leader N(0, 0)
    first = x.copy()
    env.reset()
    assert not np.array_equal(first, env.x.cpu().numpy())  # counter advances
    env2 = vec.VecPlatoon(4096, 5, conf, rng="device", seed=3)
    env2.reset()
    assert np.array_equal(first, env2.x.cpu().numpy())  # same (seed, counter) -> same draw
    # conditional reset: flag 0 leaves the state untouched, flag 1 resets
    before = env.x.clone()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    env.reset(cond=flag)
    assert torch.equal(before, env.x)
    flag.fill_(1)
    env.reset(cond=flag)
    assert not torch.equal(before, env.x)
    confu, _ = conf_and_ep(rand_gen="uniform")
    envu = vec.VecPlatoon(4096, 5, confu, rng="device")
    envu.reset()
    xu = envu.x.cpu().numpy()
    assert np.abs(xu[..., 0]).max() <= 1.5 and np.abs(xu[..., 2]).max() <= 0.05 and abs(xu[..., 0].std() - 1.5 / np.sqrt(3)) < 0.03


@pytest.mark.parametrize("L,M", [(5, 5), (3, 3), (5, 1), (10, 10)])
def test_per_platoon_episode_end_closes_only_the_platoons_that_ended(L, M):
    """avd_episode_end_f32 (the vectorised-environment form of workers/trainer.py:232-273): a platoon closes its episode when
    its step was terminal or its episode reached the limit -- float32 counters into the per-platoon statistics in vehicle order
    (trainer.py:510-517), counters and length restart, fresh states identical to what a full reset at the same (seed, counter)
    gives that platoon (bit-exact); every other platoon is left alone with its length advanced by one."""
    need_gpu()
    conf, _ = conf_and_ep()
    P, limit = 301, 12
    rs = np.random.RandomState(11)
    env = vec.VecPlatoon(P, L, conf, rng="device", seed=5)
    env.reset()
    ref = vec.VecPlatoon(P, L, conf, rng="device", seed=5)
    ref.reset()  # reset_count 1 on both from here on
    x0, pa0 = env.x.clone(), env.prev_a.clone()
    done = rs.uniform(size=P) < 0.2
    ep_len = rs.randint(0, limit, size=P).astype(np.int32)  # len + 1 >= limit closes too
    env.done.copy_(t(done.astype(np.uint8), torch.uint8))
    env.episode_end(torch.zeros(P, M, device="cuda"), M, limit)  # first call creates the statistics; redo with chosen contents
    env.x.copy_(x0), env.prev_a.copy_(pa0)
    env.ep_len.copy_(t(ep_len, torch.int32))
    for v in env.ep_stats.values():
        v.zero_()
    ret0 = rs.normal(size=P).astype(np.float32)
    env.ep_stats["ret_sum"].copy_(t(ret0))
    er = rs.normal(-20, 5, size=(P, M)).astype(np.float32)
    ep_reward = t(er)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert env.reset_count == 2
    env.episode_end(ep_reward, M, limit, any_reset=flag)
    ref.reset_count = 2
    ref.reset()  # the same (seed, counter): what a closed platoon must now hold
    end = done | (ep_len + 1 >= limit)
    assert end.any() and (~end).any() and int(flag.item()) == 1
    x, pa = env.x.cpu().numpy(), env.prev_a.cpu().numpy()
    assert np.array_equal(x[end], ref.x.cpu().numpy()[end]) and np.array_equal(pa[end], ref.prev_a.cpu().numpy()[end])
    assert np.array_equal(x[~end], x0.cpu().numpy()[~end]) and np.array_equal(pa[~end], pa0.cpu().numpy()[~end])
    got_len = env.ep_len.cpu().numpy()
    assert np.array_equal(got_len, np.where(end, 0, ep_len + 1))
    want_ret = ret0.copy()
    for p in np.nonzero(end)[0]:
        s_ = np.float32(0)
        for m in range(M):
            s_ = np.float32(s_ + er[p, m])
        want_ret[p] = np.float32(want_ret[p] + np.float32(s_ / np.float32(M)))
    assert np.array_equal(env.ep_stats["ret_sum"].cpu().numpy(), want_ret)
    assert np.array_equal(env.ep_stats["count"].cpu().numpy(), end.astype(np.int32))
    assert np.array_equal(env.ep_stats["len_sum"].cpu().numpy(), np.where(end, ep_len + 1, 0).astype(np.float32))
    got_er = ep_reward.cpu().numpy()
    assert np.all(got_er[end] == 0) and np.array_equal(got_er[~end], er[~end])
    mean_ret, mean_len, n = env.pop_episode_stats()
    assert n == int(end.sum()) and abs(mean_len - (ep_len + 1)[end].mean()) < 1e-4 and int(env.ep_stats["count"].sum()) == 0
    # nothing ends: no flag, no state change
    env.done.zero_()
    env.ep_len.zero_()
    flag.zero_()
    before = env.x.clone()
    env.episode_end(ep_reward, M, limit, any_reset=flag)
    assert int(flag.item()) == 0 and torch.equal(before, env.x) and int(env.ep_len.min()) == 1


def test_ou_noise_matches_reference_sequence():
    need_gpu()
    g = np.load(os.path.join(G, "g4_ou.npz"))
    conf, _ = conf_and_ep()
    for seed in (1, 2):
        ou = vec.VecOUNoise(1, conf)
        x32 = np.zeros(1, np.float32)
        for k in range(256):
            out = ou(g[f"seed{seed}_normals"][k:k + 1])
            x32 = onoise.batched_ou_step(x32, g[f"seed{seed}_normals"][k:k + 1], dtype=np.float32)
            assert out.item() == x32[0]  # bit-exact vs the f32 oracle
        assert abs(out.item() - g[f"seed{seed}"][-1]) <= REL * max(1.0, abs(g[f"seed{seed}"][-1]))
    # device RNG: stationary innovation std = std_dev*sqrt(dt) = 0.002 per step
    ou = vec.VecOUNoise(200000, conf, rng="device", seed=9)
    s1 = ou().clone()
    assert abs(s1.std().item() / 0.002 - 1) < 0.02 and abs(s1.mean().item()) < 5e-5
    s2 = ou()
    assert not torch.equal(s1, s2)


def test_ou_noise_with_a_non_zero_mean_follows_the_reference_recurrence():
    """src/noise.py:15-19 with mean != 0 (the class refused it although avd_ou_step_f32 takes `mean`): the
    reference-shaped OUActionNoise against the oracle's float64 restatement of the reference class on the same global RNG stream --
    a scalar mean, a vector of equal means and a vector of DIFFERENT means (one process per element) with an x_init; bit-exact
    against the float32 recurrence for the batched class."""
    from avddpg_amd import noise

    need_gpu()
    conf, _ = conf_and_ep()
    for mean, x_init in ((np.array([0.3]), None), (np.array([-0.7, -0.7, -0.7]), None), (np.array([0.5, -0.2]), np.array([1.0, 2.0]))):
        a = noise.OUActionNoise(mean=mean, x_init=x_init, config=conf)
        b = onoise.RefOUNoise(mean.astype(np.float64), std_dev=conf.std_dev, theta=conf.theta, dt=conf.ou_dt, x_init=x_init)
        np.random.seed(31)
        seq_a = [a() for _ in range(300)]
        np.random.seed(31)  # (both draw from the global legacy stream: one after the other, from the same seed)
        seq_b = [b() for _ in range(300)]
        for k, (xa, xb) in enumerate(zip(seq_a, seq_b)):
            assert xa.shape == mean.shape and np.allclose(xa, xb, rtol=1e-5, atol=1e-6), (k, xa, xb)
        assert np.all(np.abs(xa - mean) < np.abs((x_init if x_init is not None else 0.0) - mean))  # it does revert towards the mean
        a.reset()
        assert np.array_equal(a.x_prev, x_init if x_init is not None else np.zeros_like(mean))
    ou = vec.VecOUNoise(4, conf, mean=0.25)
    x32 = np.zeros(4, np.float32)
    rs = np.random.RandomState(3)
    for k in range(64):
        n = rs.normal(size=4).astype(np.float32)
        out = ou(n)
        x32 = onoise.batched_ou_step(x32, n, mean=0.25, dtype=np.float32)
        assert np.array_equal(out.cpu().numpy(), x32)  # bit-exact vs the f32 oracle


def test_policy_clip_and_noise():
    need_gpu()
    from avddpg_amd._hip import call, ptr, stream_handle
    a = t(np.array([2.4, -2.4, 0.5, 1.0]))
    n = t(np.array([0.3, -0.3, 0.1, 0.0]))
    out = torch.empty(4, device="cuda")
    call("avd_policy_f32", 4, ptr(a), ptr(n), -2.5, 2.5, ptr(out), stream_handle())
    assert np.array_equal(out.cpu().numpy(), np.clip(np.float32([2.4, -2.4, 0.5, 1.0]) + np.float32([0.3, -0.3, 0.1, 0.0]), -2.5, 2.5))
    call("avd_policy_f32", 4, ptr(a), None, -2.5, 2.5, ptr(out), stream_handle())
    assert torch.equal(out, a)
