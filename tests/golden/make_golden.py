#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Run in the build container only (it needs /root/reference):

    python tests/golden/make_golden.py

The NumPy half of the reference (src/environment.py, src/noise.py,
src/replaybuffer.py, src/util.get_random_val, src/config.py) is imported with
inert stand-in modules for the two third-party imports that are absent from
this image (``tensorflow`` and ``h5py`` -- they are only touched by
ReplayBuffer.sample for ``convert_to_tensor``/``cast``, mapped to numpy here).
Nothing from the reference is copied: the fixtures are *data* -- inputs and
the outputs the reference produced for them.

Fixtures:
  G1 system matrices          g1_matrices.json
  G2 reset states / RNG count g2_reset.npz
  G3 Platoon.step traces      g3_step.npz
  G4 OU noise sequences       g4_ou.npz
  G5 replay ring + samples    g5_replay.npz
  G6 trainer inner-loop trace g6_loop.npz
  G7 federated table          g7_federated.json
  G8 evaluator rollout (stub) g8_evaluator.npz
  G9 same, centralized        g9_evaluator_centralized.npz
(g10_curves.npz is NOT made here and is not a reference-derived vector: it holds the ORACLE loop's long-run reward curves and
 evaluator scores at BASELINE configs[0] on 5 seeds over the reference's full schedule (1666 episodes) -- `python tools/train_curves.py
 oracle --episodes 1666`, 3.6 hours on 5 cores -- which tests/test_gpu_curves.py compares the GPU trainer's curves with. The reference itself cannot
 run here: its trainer imports TensorFlow.)
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    tf = types.ModuleType("tensorflow")
    tf.float32 = np.float32
    tf.convert_to_tensor = lambda x: np.asarray(x)
    tf.cast = lambda x, dtype=None: np.asarray(x).astype(dtype)
    sys.modules["tensorflow"] = tf
    sys.modules["h5py"] = types.ModuleType("h5py")


_install_stubs()
sys.path.insert(0, REF)
import logging  # noqa: E402

logging.disable(logging.CRITICAL)
import contextlib  # noqa: E402
import io  # noqa: E402

from src import config as ref_config  # noqa: E402
from src import environment as ref_env  # noqa: E402
from src import noise as ref_noise  # noqa: E402
from src import replaybuffer as ref_rb  # noqa: E402
from src import util as ref_util  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_platoon(L, conf, idx=0, **kw):
    """Platoon ctor; for L>6 the ValueError is the LAST ctor statement
    (reference src/environment.py:84-85) so the object is fully built."""
    p = ref_env.Platoon.__new__(ref_env.Platoon)
    try:
        quiet(p.__init__, L, conf, idx, **kw)
    except ValueError:
        if L <= 6:
            raise
    return p


class DrawCounter:
    """Counts calls into the global legacy RNG made through util.get_random_val."""

    def __init__(self):
        self.n = 0
        self._orig = ref_util.get_random_val

    def __enter__(self):
        def wrapped(*a, **k):
            self.n += 1
            return self._orig(*a, **k)

        ref_util.get_random_val = wrapped
        return self

    def __exit__(self, *exc):
        ref_util.get_random_val = self._orig


# --------------------------------------------------------------------- G1
def g1():
    cases = []
    for method in ("euler", "exact"):
        for (T, tau, tau_lead, h) in [(0.1, 0.1, 0.1, 1.0), (0.1, 0.25, 0.15, 1.0), (0.05, 0.5, 0.5, 0.8)]:
            conf = ref_config.Config()
            conf.method = method
            conf.sample_rate = T
            conf.dyn_coeff = tau
            conf.timegap = h
            v = quiet(ref_env.Vehicle, 0, conf, tau_lead, 0.0, num_states=4, num_actions=1)
            cases.append(dict(method=method, T=T, tau=tau, tau_lead=tau_lead, h=h,
                              A=[[repr(float(x)) for x in row] for row in v.A],
                              B=[repr(float(x)) for x in v.B], C=[repr(float(x)) for x in v.C]))
    with open(os.path.join(OUT, "g1_matrices.json"), "w") as f:
        json.dump(cases, f, indent=1)


# --------------------------------------------------------------------- G2
def g2():
    out = {}
    meta = []
    for seed in (1, 2, 3):
        for L in (3, 5, 10):
            for rand_gen in ("normal", "uniform"):
                for model in ("ModelA", "ModelB"):
                    for mode in ("train", "evaluator", "fixed"):
                        conf = ref_config.Config()
                        conf.rand_gen = rand_gen
                        conf.model = model
                        conf.pl_size = L
                        kw = {}
                        if mode == "evaluator":
                            kw = dict(evaluator_states_enabled=True)
                        elif mode == "fixed":
                            kw = dict(rand_states=False)
                        np.random.seed(seed)
                        with DrawCounter() as dc0:
                            p = make_platoon(L, conf, 0, **kw)
                        ctor_x = np.array([f.x for f in p.followers], dtype=np.float64)
                        with DrawCounter() as dc1:
                            st = p.reset()
                        full_x = np.array([f.x for f in p.followers], dtype=np.float64)
                        key = f"s{seed}_L{L}_{rand_gen}_{model}_{mode}"
                        out[key + "__ctor_x"] = ctor_x
                        out[key + "__reset_obs"] = np.array([np.asarray(s, dtype=np.float64) for s in st])
                        out[key + "__reset_x"] = full_x
                        out[key + "__draws"] = np.array([dc0.n, dc1.n])
                        # RNG position after ctor+reset: next legacy normal draw
                        out[key + "__next_normal"] = np.array(np.random.normal(0, 1))
                        meta.append(key)
    out["keys"] = np.array(meta)
    np.savez_compressed(os.path.join(OUT, "g2_reset.npz"), **out)


# --------------------------------------------------------------------- G3
def _trace(p, K, rs, L, force_exog=True):
    obs, rew, done, jerk, vel, hw, fullx = [], [], [], [], [], [], []
    acts = rs.uniform(-2.5, 2.5, size=(K, L))
    exog = rs.normal(0, 0.1, size=K)
    for k in range(K):
        s, r, d = p.step(list(acts[k]), float(exog[k]))
        obs.append(np.array([np.asarray(x, dtype=np.float64).ravel() for x in s]))
        rew.append(np.array(r, dtype=np.float64))
        done.append(bool(d))
        jerk.append(np.array(p.get_jerk(), dtype=np.float64).ravel())
        vel.append(np.array([f.velocity for f in p.followers]))
        hw.append(np.array([f.headway for f in p.followers]))
        fullx.append(np.array([f.x for f in p.followers], dtype=np.float64))
    return dict(actions=acts, exog=exog, obs=np.array(obs), rewards=np.array(rew), done=np.array(done),
                jerk=np.array(jerk), velocity=np.array(vel), headway=np.array(hw), x=np.array(fullx))


def g3():
    out = {}
    keys = []
    K = 64

    def run_case(key, L, seed, mutate=None, **confkw):
        conf = ref_config.Config()
        conf.pl_size = L
        for k, v in confkw.items():
            setattr(conf, k, v)
        np.random.seed(seed)
        p = make_platoon(L, conf, 0)
        p.reset()
        if mutate:
            mutate(p)
        x0 = np.array([f.x for f in p.followers], dtype=np.float64)
        pa0 = np.array([f.prev_x[2] for f in p.followers], dtype=np.float64)
        tr = _trace(p, K, np.random.RandomState(1234), L)
        out[key + "__x0"] = x0
        out[key + "__prev_a0"] = pa0
        for k, v in tr.items():
            out[key + "__" + k] = v
        keys.append(key)

    for L in (3, 5, 10):
        for model in ("ModelA", "ModelB"):
            for method in ("euler", "exact"):
                run_case(f"L{L}_{model}_{method}", L, 1, model=model, method=method)

    # forced terminal: x0 of vehicle 1 beyond max_ep
    def blow(p):
        x = p.followers[1].x.copy()
        x[0] = 25.0
        p.followers[1].x = x
        p.followers[1].prev_x = x

    run_case("terminal_L3", 3, 2, mutate=blow)
    run_case("terminal_off_L3", 3, 2, mutate=blow, can_terminate=False)
    # non-degenerate tau != T, h != 1, rescaled reward
    run_case("nondegenerate_L5_B", 5, 3, model="ModelB", method="exact", dyn_coeff=0.25, pl_leader_tau=0.15,
             timegap=0.8, sample_rate=0.05, re_scalar=2.0)
    run_case("nondegenerate_L5_A", 5, 3, model="ModelA", method="euler", dyn_coeff=0.25, pl_leader_tau=0.15,
             timegap=0.8, sample_rate=0.05)
    # centralized framework: concatenated obs, platoon-mean reward
    run_case("centralized_L3", 3, 1, framework="centralized")
    run_case("centralized_L3_A", 3, 1, framework="centralized", model="ModelA")
    out["keys"] = np.array(keys)
    np.savez_compressed(os.path.join(OUT, "g3_step.npz"), **out)


# --------------------------------------------------------------------- G4
def g4():
    out = {}
    for seed in (1, 2):
        conf = ref_config.Config()
        np.random.seed(seed)
        ou = ref_noise.OUActionNoise(mean=np.zeros(1), config=conf)
        out[f"seed{seed}"] = np.array([ou()[0] for _ in range(256)], dtype=np.float64)
        np.random.seed(seed)
        out[f"seed{seed}_normals"] = np.random.normal(0, 1.0, size=256)
    np.savez_compressed(os.path.join(OUT, "g4_ou.npz"), **out)


# --------------------------------------------------------------------- G5
def g5():
    out = {}
    keys = []
    for cap, B, nadd in ((8, 4, 20), (100000, 64, 5000)):
        for seed in (1, 2):
            rb = ref_rb.ReplayBuffer(cap, B, 4, 1, 3)
            rs = np.random.RandomState(100 + seed)
            rows = rs.normal(size=(nadd, 10))
            for k in range(nadd):
                rb.add((rows[k, 0:4], rows[k, 4:5], rows[k, 5], rows[k, 6:10]))
            np.random.seed(seed)
            # indices the reference will draw (same legacy stream)
            idx = np.random.choice(min(rb.buffer_counter, cap), B)
            np.random.seed(seed)
            s, a, r, s2 = rb.sample()
            key = f"cap{cap}_B{B}_seed{seed}"
            keys.append(key)
            out[key + "__rows"] = rows
            out[key + "__counter"] = np.array(rb.buffer_counter)
            out[key + "__idx"] = idx.astype(np.int64)
            out[key + "__s"] = s
            out[key + "__a"] = a
            out[key + "__r"] = r
            out[key + "__s2"] = s2
            out[key + "__dtypes"] = np.array([str(s.dtype), str(a.dtype), str(r.dtype), str(s2.dtype)])
            if cap == 8:
                out[key + "__ring_s"] = rb.state_buffer.copy()
                out[key + "__ring_a"] = rb.action_buffer.copy()
                out[key + "__ring_r"] = rb.reward_buffer.copy()
                out[key + "__ring_s2"] = rb.next_state_buffer.copy()
    out["keys"] = np.array(keys)
    np.savez_compressed(os.path.join(OUT, "g5_replay.npz"), **out)


# --------------------------------------------------------------------- G6
def g6():
    """Hand-driven replica of the Trainer.run inner loop (workers/trainer.py:246-271,
    282-302, 314-322) using the reference's Platoon / OUActionNoise / ReplayBuffer
    objects and a stub actor that outputs 0 (TensorFlow is absent). Pins the RNG
    interleaving, the strict counter>batch gate and the any-terminal break."""
    out = {}
    for P in (1, 2):
        conf = ref_config.Config()
        conf.pl_size = 3
        conf.num_platoons = P
        L = 3
        np.random.seed(1)
        envs = [make_platoon(L, conf, p, rand_states=conf.rand_states) for p in range(P)]
        ous = [[ref_noise.OUActionNoise(mean=np.zeros(1), config=conf) for _ in range(L)] for _ in range(P)]
        rbs = [[ref_rb.ReplayBuffer(128, conf.batch_size, 4, 1, L) for _ in range(L)]
               for _ in range(P)]
        prev = [envs[p].reset() for p in range(P)]
        reset_obs = np.array([[np.asarray(s) for s in prev[p]] for p in range(P)])
        actions = np.zeros((P, L, 1))
        A, R, S, D, IDX, GATE = [], [], [], [], [], []
        eprew = [np.array([0] * L, dtype=np.float32) for _ in range(P)]
        for i in range(70):
            states_all, rew_all, term_all = [], [], []
            for p in range(P):
                for m in range(L):
                    n = ous[p][m]()
                    actions[p][m] = [np.squeeze(np.clip(0.0 + n, conf.action_low, conf.action_high))]
                s, r, d = envs[p].step(actions[p].flatten(),
                                       ref_util.get_random_val(conf.rand_gen, conf.reset_max_u,
                                                               std_dev=conf.reset_max_u, config=conf))
                states_all.append(s)
                rew_all.append(r)
                term_all.append(d)
            idx_step = np.full((P, L, conf.batch_size), -1, dtype=np.int64)
            gate = np.zeros((P, L), dtype=bool)
            for p in range(P):
                for m in range(L):
                    rbs[p][m].add((prev[p][m], actions[p][m], rew_all[p][m], states_all[p][m]))
                    eprew[p][m] += rew_all[p][m]
                    if rbs[p][m].buffer_counter > conf.batch_size:
                        gate[p, m] = True
                        rr = min(rbs[p][m].buffer_counter, rbs[p][m].buffer_capacity)
                        st = np.random.get_state()
                        idx_step[p, m] = np.random.choice(rr, conf.batch_size)
                        np.random.set_state(st)
                        rbs[p][m].sample()  # consumes the same draws the trainer would
            A.append(actions.copy())
            R.append(np.array(rew_all, dtype=np.float64))
            S.append(np.array([[np.asarray(x) for x in states_all[p]] for p in range(P)]))
            D.append(np.array(term_all))
            IDX.append(idx_step)
            GATE.append(gate)
            if True in term_all:
                break
            prev = states_all
        out[f"P{P}__reset_obs"] = reset_obs
        out[f"P{P}__actions"] = np.array(A)
        out[f"P{P}__rewards"] = np.array(R)
        out[f"P{P}__states"] = np.array(S)
        out[f"P{P}__done"] = np.array(D)
        out[f"P{P}__idx"] = np.array(IDX)
        out[f"P{P}__gate"] = np.array(GATE)
        out[f"P{P}__ep_reward_f32"] = np.array(eprew)
        out[f"P{P}__first_gate_step"] = np.array(int(np.argmax(np.array(GATE).any(axis=(1, 2)))))
        out[f"P{P}__next_normal"] = np.array(np.random.normal(0, 1))
    np.savez_compressed(os.path.join(OUT, "g6_loop.npz"), **out)


# --------------------------------------------------------------------- G7
def g7():
    """The input table of src/server/test_federated.py:26-42 (data only) and the
    means it implies, computed here by hand in float64 (TensorFlow is absent, so
    Server.get_*avg_params itself cannot run): interfrl groups by model index."""
    pl = [[[[1, 2, 3], [1, 2], [3, 4]], [[7, 8, 9], [5, 6], [7, 8]]],
          [[[10, 11, 12], [9, 10], [11, 12]], [[13, 14, 15], [13, 14], [15, 16]]]]
    w = [[2, 0.5], [1, 6]]
    res = {"grads_list": pl, "weights": w}
    P, M = 2, 2
    unweighted, weighted, wsums = [], [], []
    for m in range(M):
        layers_u, layers_w = [], []
        ws = sum(w[p][m] for p in range(P))
        for layer in range(3):
            arrs = [np.array(pl[p][m][layer], dtype=np.float64) for p in range(P)]
            layers_u.append(list(np.mean(arrs, axis=0)))
            layers_w.append(list(sum(w[p][m] * arrs[p] for p in range(P)) / ws))
        unweighted.append(layers_u)
        weighted.append(layers_w)
        wsums.append(ws)
    res["interfrl_unweighted"] = unweighted
    res["interfrl_weighted"] = weighted
    res["interfrl_weight_sums"] = wsums
    with open(os.path.join(OUT, "g7_federated.json"), "w") as f:
        json.dump(res, f, indent=1)


# --------------------------------------------------------------------- G8
def g8(cases=((2, "ModelB", 600, "decentralized"), (3, "ModelA", 100, "decentralized")), fname="g8_evaluator.npz"):
    """Hand-driven replica of the evaluator rollout (workers/evaluator.py:40-95, 145) on the reference's own
    Platoon object with a stub actor that outputs 0 (TensorFlow is absent): seeding with evaluation_seed,
    evaluator-mode Platoon, pre-drawn leader input list, reset, noise-free steps, float32 reward counters,
    pl_rew = round(mean, 3)."""
    out = {}
    for L, model, T, framework in cases:
        conf = ref_config.Config()
        conf.pl_size, conf.model, conf.framework = L, model, framework
        np.random.seed(conf.evaluation_seed)  # rand.set_global_seed(conf.evaluation_seed) (src/rand.py:10)
        env = make_platoon(L, conf, 1, evaluator_states_enabled=True)
        inputs = [ref_util.get_random_val(conf.rand_gen, conf.reset_max_u, std_dev=conf.reset_max_u, config=conf)
                  for _ in range(T)]
        counters = np.array([0] * env.num_models, dtype=np.float32)
        states = env.reset()
        S, R, J = [], [], []
        actions = np.zeros((env.num_models, env.num_actions))
        for i in range(T):
            states, rewards, terminal = env.step(actions.flatten(), inputs[i])
            for m in range(env.num_models):
                counters[m] += rewards[m]
            S.append(np.array([np.asarray(s) for s in states]))
            R.append(np.array(rewards))
            J.append(np.array(env.get_jerk()).ravel())
        key = f"L{L}_{model}" + ("" if framework == "decentralized" else "_" + framework)
        out[key + "__inputs"] = np.array(inputs)
        out[key + "__states"] = np.array(S)
        out[key + "__rewards"] = np.array(R)
        out[key + "__jerks"] = np.array(J)
        out[key + "__counters"] = counters
        out[key + "__pl_rew"] = np.array(round(np.average(counters), 3))
    np.savez_compressed(os.path.join(OUT, fname), **out)


def g9():
    """G8 for the centralized framework: one model, the 4L-wide observation, the platoon-mean reward
    (src/environment.py:234-236, 281)."""
    g8(cases=((3, "ModelB", 200, "centralized"), (1, "ModelB", 100, "centralized")),
       fname="g9_evaluator_centralized.npz")


if __name__ == "__main__":
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9):
        fn()
        print("wrote", fn.__name__)
