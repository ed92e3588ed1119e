"""Pin the oracle against golden vectors captured from the reference's own NumPy code
(tests/golden/make_golden.py). CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import noise, platoon, replay

G = os.path.join(os.path.dirname(__file__), "golden")


def _ep_for(key):
    ep = platoon.EnvParams()
    if "ModelA" in key or key.endswith("_A"):
        ep.model = platoon.MODEL_A
    if "exact" in key:
        ep.method = "exact"
    if key.startswith("terminal_off"):
        ep.can_terminate = False
    if key.startswith("nondegenerate"):
        ep.dyn_coeff, ep.pl_leader_tau, ep.timegap, ep.sample_rate = 0.25, 0.15, 0.8, 0.05
        if key.endswith("_B"):
            ep.method, ep.re_scalar = "exact", 2.0
    if key.startswith("centralized"):
        ep.framework = "centralized"
    return ep


def test_g1_system_matrices():
    for c in json.load(open(os.path.join(G, "g1_matrices.json"))):
        A, B, C = platoon.system_matrices(c["method"], c["T"], c["tau"], c["tau_lead"], c["h"])
        assert np.array_equal(A, np.array([[float(v) for v in r] for r in c["A"]]))
        assert np.array_equal(B, np.array([float(v) for v in c["B"]]))
        assert np.array_equal(C, np.array([float(v) for v in c["C"]]))


def test_g2_reset_object_form_bit_exact():
    g = np.load(os.path.join(G, "g2_reset.npz"))
    for key in g["keys"]:
        seed, L, rand_gen, model, mode = key.split("_")
        ep = platoon.EnvParams(model=model, rand_gen=rand_gen)
        kw = dict(evaluator_states=(mode == "evaluator"), rand_states=(mode != "fixed"))
        np.random.seed(int(seed[1:]))
        p = platoon.RefPlatoon(int(L[1:]), ep, **kw)
        assert np.array_equal(np.array([f.x for f in p.followers]), g[key + "__ctor_x"]), key
        obs = p.reset()
        assert np.array_equal(np.array([f.x for f in p.followers]), g[key + "__reset_x"]), key
        assert np.array_equal(np.array([np.asarray(o) for o in obs]), g[key + "__reset_obs"]), key
        # same number of RNG draws consumed: the next legacy draw coincides
        assert np.random.normal(0, 1) == float(g[key + "__next_normal"]), key


def test_g2_reset_batched_form():
    g = np.load(os.path.join(G, "g2_reset.npz"))
    for key in g["keys"]:
        seed, L, rand_gen, model, mode = key.split("_")
        L = int(L[1:])
        ep = platoon.EnvParams(model=model, rand_gen=rand_gen)
        np.random.seed(int(seed[1:]))
        platoon.RefPlatoon(L, ep, evaluator_states=(mode == "evaluator"), rand_states=(mode != "fixed"))  # ctor draws
        draws, fa = platoon.host_reset_draws(ep, 1, L, mode)
        x, prev_a = platoon.batched_reset(ep, draws, fa, mode, dtype=np.float64)
        assert np.array_equal(x[0], g[key + "__reset_x"]), key
        assert np.array_equal(prev_a[0], g[key + "__reset_x"][:, 2])
        assert np.random.normal(0, 1) == float(g[key + "__next_normal"]), key
        assert tuple(g[key + "__draws"]) == ((2 + 3 * L, 1 + 4 * L) if mode == "train" else (2, 1 + L)), key


def test_g3_step_object_form_bit_exact():
    g = np.load(os.path.join(G, "g3_step.npz"))
    for key in g["keys"]:
        ep = _ep_for(key)
        x0 = g[key + "__x0"]
        L = x0.shape[0]
        np.random.seed(0)
        p = platoon.RefPlatoon(L, ep)
        for i, f in enumerate(p.followers):
            f.x = x0[i].copy()
            f.prev_x = f.x.copy()
            f.prev_x[2] = g[key + "__prev_a0"][i]
        for k in range(len(g[key + "__actions"])):
            s, r, d = p.step(list(g[key + "__actions"][k]), float(g[key + "__exog"][k]))
            assert np.array_equal(np.array([np.asarray(v, dtype=np.float64).ravel() for v in s]), g[key + "__obs"][k]), (key, k)
            assert np.array_equal(np.array(r, dtype=np.float64), g[key + "__rewards"][k]), (key, k)
            assert d == bool(g[key + "__done"][k])
            assert np.array_equal(np.array(p.get_jerk()).ravel(), g[key + "__jerk"][k])
            assert np.array_equal(np.array([f.velocity for f in p.followers]), g[key + "__velocity"][k])
            assert np.array_equal(np.array([f.headway for f in p.followers]), g[key + "__headway"][k])


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-13), (np.float32, 1e-5)])
def test_g3_step_batched_form(dtype, tol):
    """Batched comparator == reference trajectories: per-step (teacher-forced from the
    golden state) and free-running over the 64-step trace."""
    g = np.load(os.path.join(G, "g3_step.npz"))
    for key in g["keys"]:
        ep = _ep_for(key)
        xs = np.concatenate([g[key + "__x0"][None], g[key + "__x"]], axis=0)  # state before step k
        K, L = g[key + "__actions"].shape
        pa = g[key + "__prev_a0"].copy()
        x_free = xs[0][None].astype(dtype)
        pa_free = pa[None].astype(dtype)
        cum_free = np.zeros((1, L), dtype=dtype)
        cum = np.zeros((1, L))
        for k in range(K):
            o = platoon.batched_step(ep, xs[k][None], pa[None], cum, g[key + "__actions"][k][None],
                                     g[key + "__exog"][k][None], dtype=dtype)
            scale = np.maximum(1.0, np.abs(g[key + "__x"][k]))
            assert np.all(np.abs(o["x"][0] - g[key + "__x"][k]) <= tol * scale), (key, k)
            rew = o["reward_mean"][0] if ep.framework == "centralized" else o["reward"][0]
            assert np.allclose(rew, g[key + "__rewards"][k].ravel() if ep.framework != "centralized" else g[key + "__rewards"][k][0],
                               rtol=tol, atol=tol * 1e-2), (key, k)
            assert bool(o["done"][0]) == bool(g[key + "__done"][k])
            assert np.allclose(o["jerk"][0], g[key + "__jerk"][k], rtol=tol, atol=tol)
            assert np.allclose(o["velocity"][0], g[key + "__velocity"][k], rtol=tol, atol=tol)
            assert np.allclose(o["headway"][0], g[key + "__headway"][k], rtol=tol, atol=tol)
            pa = xs[k][:, 2]
            cum = o["cum_accel"].astype(np.float64)
            f = platoon.batched_step(ep, x_free, pa_free, cum_free, g[key + "__actions"][k][None], g[key + "__exog"][k][None], dtype=dtype)
            x_free, pa_free, cum_free = f["x"], f["prev_a"], f["cum_accel"]
        # free-running drift after 64 steps stays within 64x the per-step tolerance
        scale = np.maximum(1.0, np.abs(g[key + "__x"][-1]))
        assert np.all(np.abs(x_free[0] - g[key + "__x"][-1]) <= 64 * tol * scale), key


def test_g4_ou_noise():
    g = np.load(os.path.join(G, "g4_ou.npz"))
    for seed in (1, 2):
        np.random.seed(seed)
        ou = noise.RefOUNoise(np.zeros(1))
        seq = np.array([ou()[0] for _ in range(256)])
        assert np.array_equal(seq, g[f"seed{seed}"])
        x = np.zeros(1)
        for k in range(256):
            x = noise.batched_ou_step(x, g[f"seed{seed}_normals"][k:k + 1], dtype=np.float64)
            assert abs(x[0] - g[f"seed{seed}"][k]) <= 1e-15
        x = np.zeros(1, dtype=np.float32)
        for k in range(256):
            x = noise.batched_ou_step(x, g[f"seed{seed}_normals"][k:k + 1], dtype=np.float32)
        assert abs(x[0] - g[f"seed{seed}"][-1]) <= 1e-5 * max(1.0, abs(g[f"seed{seed}"][-1]))


def test_g5_replay():
    g = np.load(os.path.join(G, "g5_replay.npz"))
    for key in g["keys"]:
        cap = int(key.split("_")[0][3:])
        B = int(key.split("_")[1][1:])
        seed = int(key.split("seed")[1])
        rb = replay.RefReplayBuffer(cap, B, 4, 1)
        rows = g[key + "__rows"]
        for k in range(len(rows)):
            assert replay.ring_index(rb.buffer_counter, cap) == k % cap
            rb.add((rows[k, 0:4], rows[k, 4:5], rows[k, 5], rows[k, 6:10]))
        assert rb.buffer_counter == int(g[key + "__counter"])
        if cap == 8:
            assert np.array_equal(rb.state_buffer, g[key + "__ring_s"])
            assert np.array_equal(rb.action_buffer, g[key + "__ring_a"])
            assert np.array_equal(rb.reward_buffer, g[key + "__ring_r"])
            assert np.array_equal(rb.next_state_buffer, g[key + "__ring_s2"])
        np.random.seed(seed)
        idx = rb.sample_indices()
        assert idx.dtype == g[key + "__idx"].dtype and np.array_equal(idx, g[key + "__idx"])  # bit-exact ints
        assert idx.max() < replay.sample_range(rb.buffer_counter, cap)
        s, a, r, s2 = rb.gather(idx)
        for got, name in ((s, "s"), (a, "a"), (r, "r"), (s2, "s2")):
            assert np.array_equal(got, g[key + "__" + name])
        assert [str(v.dtype) for v in (s, a, r, s2)] == list(g[key + "__dtypes"])


def test_g6_loop_rng_interleaving():
    """The oracle's reference-shaped loop consumes the global RNG in the same order as the
    reference objects: OU draws, leader exog, sample indices; gate fires at the 65th add."""
    from oracle import mlp, trainer

    g = np.load(os.path.join(G, "g6_loop.npz"))
    for P in (1, 2):
        tr = trainer.RefTrainer(platoon.EnvParams(), P, 3, seed=1, buffer_size=128)
        # stub zero actor: last layer weights & bias zero -> tanh(0)*high = 0
        for p in range(P):
            for m in range(3):
                tr.actors[p][m][12][:] = 0
        # learning must not alter the zero actor's output: freeze by making learn a sampler only
        orig_learn = mlp.learn
        idx_log = []

        def fake_learn(batch, *a, **k):
            return None, None, None

        tr._apply_local = lambda *a, **k: None
        mlp.learn = fake_learn
        try:
            tr.reset_episode()
            assert np.array_equal(np.array([[np.asarray(s) for s in tr.prev_states[p]] for p in range(P)]), g[f"P{P}__reset_obs"])
            n = len(g[f"P{P}__actions"])
            first_gate = None
            for i in range(n):
                before = [[tr.rbufs[p][m].buffer_counter for m in range(3)] for p in range(P)]
                done = tr.step()
                assert np.array_equal(tr.actions, g[f"P{P}__actions"][i]), (P, i)
                assert np.array_equal(np.array([[np.asarray(s) for s in tr.prev_states[p]] for p in range(P)]), g[f"P{P}__states"][i])
                assert done == bool(g[f"P{P}__done"][i].any())
                gate = np.array(before) + 1 > 64
                assert np.array_equal(gate, g[f"P{P}__gate"][i])
                if gate.any() and first_gate is None:
                    first_gate = i
            assert first_gate == int(g[f"P{P}__first_gate_step"]) == 64
            assert np.array_equal(np.array(tr.ep_reward), g[f"P{P}__ep_reward_f32"])
            assert np.random.normal(0, 1) == float(g[f"P{P}__next_normal"])
        finally:
            mlp.learn = orig_learn


def test_g7_federated_table():
    from oracle import federated

    t = json.load(open(os.path.join(G, "g7_federated.json")))
    pl, w = t["grads_list"], t["weights"]
    P, M = 2, 2
    sysu = [[[np.array(pl[p][m][i], dtype=np.float32) for i in range(3)] for p in range(P)] for m in range(M)]
    got = federated.get_avg_params(sysu)
    for m in range(M):
        for i in range(3):
            assert np.allclose(got[m][i], t["interfrl_unweighted"][m][i], rtol=1e-6)
    sysw = [[[np.float32(w[p][m]) * sysu[m][p][i] for i in range(3)] for p in range(P)] for m in range(M)]
    ws = [sum(w[p][m] for p in range(P)) for m in range(M)]
    assert ws == t["interfrl_weight_sums"]
    got = federated.get_weighted_avg_params(sysw, ws)
    for m in range(M):
        for i in range(3):
            assert np.allclose(got[m][i], t["interfrl_weighted"][m][i], rtol=1e-6)
    # hand-computed known answer from SURVEY section 4: model-1 layer-1 = (2*[1,2,3] + 1*[10,11,12]) / 3
    assert np.allclose(got[0][0], (2 * np.array([1, 2, 3.]) + np.array([10, 11, 12.])) / 3)


def test_g8_evaluator_rollout():
    from oracle import evaluator

    g = np.load(os.path.join(G, "g8_evaluator.npz"))
    for L, model, T in ((2, "ModelB", 600), (3, "ModelA", 100)):
        key = f"L{L}_{model}"
        pl_rew, tr = evaluator.run(platoon.EnvParams(model=model), L, None, T)
        assert pl_rew == float(g[key + "__pl_rew"])
        assert np.array_equal(tr["states"], g[key + "__states"]) and np.array_equal(tr["leader"], g[key + "__inputs"])
        assert np.array_equal(tr["jerks"], g[key + "__jerks"]) and np.array_equal(tr["counters"], g[key + "__counters"])


def test_g9_evaluator_rollout_centralized():
    """Centralized framework: one model, 4L-wide observation, platoon-mean reward (environment.py:234-236, 281)."""
    from oracle import evaluator

    g = np.load(os.path.join(G, "g9_evaluator_centralized.npz"))
    for L, T in ((3, 200), (1, 100)):
        key = f"L{L}_ModelB_centralized"
        pl_rew, tr = evaluator.run(platoon.EnvParams(framework="centralized"), L, None, T)
        assert pl_rew == float(g[key + "__pl_rew"]) and tr["states"].shape == (T, 1, 4 * L)
        assert np.array_equal(tr["states"], g[key + "__states"]) and np.array_equal(tr["leader"], g[key + "__inputs"])
        assert np.array_equal(tr["jerks"], g[key + "__jerks"]) and np.array_equal(tr["counters"], g[key + "__counters"])
