#!/usr/bin/env python3
"""Long-run training curves: does the GPU trainer TRAIN like the reference-shaped loop? (VERDICT r04 "next round" #1;
north_star: "matching reference episode-reward curves within tolerance"; workers/trainer.py:510-517, src/config.py:84-97,
workers/evaluator.py:47-95, 145.)

  oracle   CPU only (build container): BASELINE configs[0] -- 1 platoon x 3 vehicles, nofrl -- at the reference's schedule shape
           (600-step episodes, 40-episode trailing mean) for N episodes on K seeds through oracle.trainer.RefTrainer, one
           single-thread process per seed; writes tests/golden/g10_curves.npz (the fixture the -m gpu test compares with):
           per-episode rewards and lengths, the first LOCK steps' actions, evaluator scores of the untrained actors and of the
           actors after episodes 40 / 70 / 100 (the global RNG state is saved around every evaluator call, on both sides).
  gpu      GPU box: the same seeds through avddpg_amd.trainer.Trainer(rng="host"); writes the curves in the reference CSV schema
           (workers/trainer.py:598-611) + a summary JSON under --out.
  big      GPU box: 4096 x 5 interfrl with the headline engine (fused3), device RNG, >= --updates updates per weight set, with
           per-platoon auto-reset and in parity (any-terminal) mode; nofrl at --nofrl-platoons platoons; optionally config #5's
           width (hidden 1024) against the exact engine. Curves (platoon-mean episodic reward per window of finished episodes)
           and evaluator scores go to --out.
The oracle is test infrastructure: only the `oracle` sub-command (fixture generation) and tests import it.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

EVAL_AT = (40, 70, 100, 200, 300, 400, 600, 800, 1000, 1300, 1666)  # episodes after which the evaluator scores the actors (workers/evaluator.py:145)
LOCK_STEPS = 1800        # actions recorded from the start of training (3 full episodes when none ends early)
FIXTURE = os.path.join(ROOT, "tests", "golden", "g10_curves.npz")


def initial_weight_lists(conf, S=4, A=1):
    """The Keras-ordered initial weights vec.AgentGroup(seed=conf.random_seed) starts every agent from -- NumPy only."""
    from avddpg_amd import _hip, params

    dims = params.Dims(S, A, conf.actor_layer1_size, conf.actor_layer2_size, conf.critic_act_layer_size)
    lay = _hip.make_layout(S, A, *params.padded_widths(dims.H1, dims.H2, dims.Ha), conf.batch_size)
    th, st = params.init_weights(lay, np.random.RandomState(conf.random_seed),
                                 nominal=(conf.actor_layer1_size, conf.actor_layer2_size), dims=dims)
    return params.unpack(lay, th, st, "actor", dims=dims), params.unpack(lay, th, st, "critic", dims=dims)


def _eval_keep_rng(fn):
    """Evaluator rollouts reseed the global legacy RNG (src/rand.py:10 via evaluator.py:44): keep the training stream intact."""
    state = np.random.get_state()
    try:
        return fn()
    finally:
        np.random.set_state(state)


def _pad_lock(lock):
    """[LOCK_STEPS, 3] float32, NaN where the run was shorter."""
    out = np.full((LOCK_STEPS, 3), np.nan, np.float32)
    if lock:
        out[:len(lock)] = np.array(lock, np.float32)
    return out


def oracle_seed(seed, episodes, q=None, max_steps=None, log=None):
    from threadpoolctl import threadpool_limits

    from avddpg_amd import config
    from oracle import evaluator as oeval
    from oracle import platoon as oplatoon
    from oracle import trainer as otrainer

    with threadpool_limits(limits=1):
        conf = config.Config(num_platoons=1, pl_size=3, random_seed=seed)
        ep_par = oplatoon.EnvParams()
        ref = otrainer.RefTrainer(ep_par, 1, 3, seed=seed, buffer_size=conf.buffer_size, steps_per_episode=conf.steps_per_episode)
        a0, c0 = initial_weight_lists(conf)
        for m in range(3):  # workers/trainer.py:121-131: every agent and target starts from the same weights
            ref.actors[0][m], ref.t_actors[0][m] = [w.copy() for w in a0], [w.copy() for w in a0]
            ref.critics[0][m], ref.t_critics[0][m] = [w.copy() for w in c0], [w.copy() for w in c0]
        steps = conf.steps_per_episode
        score = lambda: float(_eval_keep_rng(lambda: oeval.run(ep_par, 3, ref.actors[0], steps, conf.evaluation_seed)[0]))
        evals = {0: score()}
        rewards, lengths, lock = np.zeros((episodes, 3), np.float32), np.zeros(episodes, np.int32), []
        t0 = time.perf_counter()
        for e in range(episodes):
            ref.reset_episode()
            n = 0
            for i in range(steps):
                done = ref.step(e, i)
                n += 1
                if len(lock) < LOCK_STEPS:
                    lock.append(ref.actions[0, :, 0].astype(np.float32))
                if done:
                    break
            ref.update_reward_list()
            rewards[e], lengths[e] = np.asarray(ref.ep_reward[0], np.float32), n
            if e + 1 in EVAL_AT:
                evals[e + 1] = score()
            if log and (e + 1) % 25 == 0:
                with open(log, "a") as f:
                    f.write(f"seed {seed} ep {e + 1} steps {int(lengths.sum())} t {time.perf_counter() - t0:.0f}s mean_len25 "
                            f"{lengths[e - 24:e + 1].mean():.0f} mean_rew25 {rewards[e - 24:e + 1].mean():.2f} evals {evals}\n")
            if max_steps is not None and lengths.sum() >= max_steps:
                rewards, lengths = rewards[:e + 1], lengths[:e + 1]
                break
        out = dict(seed=seed, rewards=rewards, lengths=lengths, lock=_pad_lock(lock),
                   evals=np.array([evals.get(k, np.nan) for k in (0,) + EVAL_AT], np.float64), seconds=time.perf_counter() - t0,
                   updates=ref.updates)
    if q is not None:
        q.put(out)
    return out


def cmd_oracle(args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=oracle_seed, args=(s, args.episodes, q, args.max_steps, args.log)) for s in args.seeds]
    for p in procs:
        p.start()
    got = {}
    while len(got) < len(procs):
        try:
            r = q.get(timeout=5.0)
            got[r["seed"]] = r
            print(f"seed {r['seed']}: {r['seconds']:.0f} s, {int(r['lengths'].sum())} steps, evaluator {r['evals']}", flush=True)
        except Exception:
            if not any(p.is_alive() for p in procs) and q.empty():
                break
    for p in procs:
        p.join()
    if len(got) != len(procs):
        raise SystemExit("an oracle process died")
    seeds = sorted(got)
    E = min(len(got[s]["lengths"]) for s in seeds)  # a --max-steps run: keep the episodes every seed has
    for s in seeds:
        got[s]["rewards"], got[s]["lengths"] = got[s]["rewards"][:E], got[s]["lengths"][:E]
    np.savez_compressed(args.fixture, seeds=np.array(seeds), eval_at=np.array((0,) + EVAL_AT),
                        rewards=np.stack([got[s]["rewards"] for s in seeds]), lengths=np.stack([got[s]["lengths"] for s in seeds]),
                        lock=np.stack([got[s]["lock"] for s in seeds]), evals=np.stack([got[s]["evals"] for s in seeds]),
                        updates=np.array([got[s]["updates"] for s in seeds]),
                        oracle_env_steps_per_s=np.array([got[s]["lengths"].sum() / got[s]["seconds"] for s in seeds]))
    print("wrote", args.fixture)


# ---- GPU side ---------------------------------------------------------------------------------------------------------
def gpu_seed(seed, episodes, on_episode=None):
    """BASELINE configs[0] on the GPU through the reference-shaped facade (Trainer.initialize / the loop of Trainer.run with the
    evaluator called at EVAL_AT), host RNG in the reference's draw order. Returns the same record as oracle_seed."""
    from avddpg_amd import config, evaluator, trainer

    conf = config.Config(num_platoons=1, pl_size=3, random_seed=seed)
    np.random.seed(seed)  # src/rand.py:10
    tr = trainer.Trainer(None, "curves", False, conf, rng="host")
    tr.initialize()
    eng = tr.engine
    score = lambda: float(_eval_keep_rng(lambda: evaluator.run(conf=conf, actors=eng.agents, pl_idx=1, set_mod=0)[0]))
    evals = {0: score()}
    rewards, lengths, lock = np.zeros((episodes, 3), np.float32), np.zeros(episodes, np.int32), []
    t0 = time.perf_counter()
    for e in range(episodes):
        eng.episode = e
        eng.reset_episode()
        n = 0
        for i in range(conf.steps_per_episode):
            done = eng.step(e, i)
            n += 1
            if len(lock) < LOCK_STEPS:
                lock.append(eng.actions.view(-1).cpu().numpy().copy())
            if done:
                break
        eng.update_reward_list(e)
        rewards[e], lengths[e] = eng.ep_reward.cpu().numpy()[0], n
        if e + 1 in EVAL_AT:
            evals[e + 1] = score()
        if on_episode is not None:
            on_episode(e)
    return dict(seed=seed, rewards=rewards, lengths=lengths, lock=_pad_lock(lock),
                evals=np.array([evals.get(k, np.nan) for k in (0,) + EVAL_AT], np.float64), seconds=time.perf_counter() - t0,
                updates=eng.updates, avg_lists=eng.all_avg_reward_lists, ep_lists=eng.all_ep_reward_lists, conf=conf)


def _gpu_seed_proc(seed, episodes, q):
    r = gpu_seed(seed, episodes)
    r.pop("conf")
    q.put(r)


def gpu_seeds(seeds, episodes, parallel=True):
    """The seeds as concurrent processes on one GPU (each is latency-bound at P = 1: host draws + a dozen launches per step)."""
    if not parallel:
        return {s: gpu_seed(s, episodes) for s in seeds}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_seed_proc, args=(s, episodes, q)) for s in seeds]
    for p in procs:
        p.start()
    got = {}
    while len(got) < len(procs):
        try:
            r = q.get(timeout=5.0)
            got[r["seed"]] = r
        except Exception:
            if not any(p.is_alive() for p in procs) and q.empty():
                break
    for p in procs:
        p.join()
    if len(got) != len(procs):
        raise RuntimeError("a GPU curve process died")
    return got


def trailing_mean(x, window):
    """workers/trainer.py:515: mean of the last `window` episodic rewards, per episode."""
    x = np.asarray(x, np.float64)
    return np.array([x[max(0, i + 1 - window):i + 1].mean(axis=0) for i in range(len(x))])


def part_step(a, b, tol):
    """First recorded step at which two action traces differ by more than tol (len if never)."""
    n = int(min(np.isfinite(np.asarray(a)).all(axis=1).sum(), np.isfinite(np.asarray(b)).all(axis=1).sum()))
    d = np.abs(np.asarray(a[:n], np.float64) - np.asarray(b[:n], np.float64)).max(axis=1)
    bad = np.nonzero(d > tol)[0]
    return int(bad[0]) if len(bad) else n


def compare_with_fixture(got, fx, window=40):
    """Numbers the -m gpu test asserts on and DESIGN section 5 quotes."""
    seeds = [int(s) for s in fx["seeds"]]
    E = min(fx["rewards"].shape[1], min(len(got[s]["rewards"]) for s in seeds))
    o_pl = fx["rewards"][:, :E].astype(np.float64).mean(axis=2)  # platoon-mean episodic reward [seed, ep]
    g_pl = np.stack([got[s]["rewards"][:E].astype(np.float64).mean(axis=1) for s in seeds])
    o_tr = np.stack([trailing_mean(o_pl[k], window) for k in range(len(seeds))])
    g_tr = np.stack([trailing_mean(g_pl[k], window) for k in range(len(seeds))])
    out = dict(seeds=seeds, episodes=E, part_step_1e4=[], locked_episodes=[], at={})
    for k, s in enumerate(seeds):
        ps = part_step(got[s]["lock"], fx["lock"][k], 1e-4 * 2.5)
        out["part_step_1e4"].append(ps)
        same_len = np.nonzero(got[s]["lengths"][:E] != fx["lengths"][k][:E])[0]
        out["locked_episodes"].append(int(same_len[0]) if len(same_len) else E)
    for e in [x for x in EVAL_AT if x <= E]:
        out["at"][e] = dict(oracle_mean=float(o_tr[:, e - 1].mean()), oracle_std=float(o_tr[:, e - 1].std(ddof=1)),
                            gpu_mean=float(g_tr[:, e - 1].mean()), gpu_std=float(g_tr[:, e - 1].std(ddof=1)))
    idx = [i for i, e in enumerate(fx["eval_at"]) if e <= E]
    out["eval_at"] = [int(fx["eval_at"][i]) for i in idx]
    out["eval_oracle"] = fx["evals"][:, idx].tolist()
    out["eval_gpu"] = [[float(got[s]["evals"][i]) for i in idx] for s in seeds]
    out["first_episodes_mean"] = dict(oracle=float(o_pl[:, :10].mean()), gpu=float(g_pl[:, :10].mean()))
    out["last_episodes_mean"] = dict(oracle=float(o_pl[:, E - 10:E].mean()), gpu=float(g_pl[:, E - 10:E].mean()))
    return out


def write_reference_csvs(out_dir, seed, r):
    """ep_reward__seed<N>.csv / avg_ep_reward__seed<N>.csv in the reference's schema (workers/trainer.py:552-568, 598-611; column
    names src/env/env.py:5-11; file names src/config.py:135-138) through the package's own writer (avddpg_amd/artifacts.py)."""
    from avddpg_amd import artifacts, config

    os.makedirs(out_dir, exist_ok=True)
    return artifacts.generate_csvs(out_dir, config.Config(num_platoons=1, pl_size=3, random_seed=seed), r["ep_lists"], r["avg_lists"])


def cmd_gpu(args):
    fx = np.load(args.fixture)
    seeds = [int(s) for s in fx["seeds"]] if args.seeds is None else args.seeds
    t0 = time.perf_counter()
    got = gpu_seeds(seeds, args.episodes, parallel=not args.serial)
    wall = time.perf_counter() - t0
    os.makedirs(args.out, exist_ok=True)
    cmp_ = compare_with_fixture(got, fx)
    cmp_["gpu_wall_s"] = wall
    cmp_["gpu_env_steps_per_s_per_process"] = [float(got[s]["lengths"].sum() / got[s]["seconds"]) for s in seeds]
    cmp_["oracle_env_steps_per_s"] = fx["oracle_env_steps_per_s"].tolist()
    np.savez_compressed(os.path.join(args.out, "config1_gpu_curves.npz"), seeds=np.array(seeds),
                        rewards=np.stack([got[s]["rewards"] for s in seeds]), lengths=np.stack([got[s]["lengths"] for s in seeds]),
                        evals=np.stack([got[s]["evals"] for s in seeds]))
    with open(os.path.join(args.out, "config1_summary.json"), "w") as f:
        json.dump(cmp_, f, indent=1)
    for s_ in seeds:
        write_reference_csvs(os.path.join(args.out, "config1_reference_schema"), s_, got[s_])
    # the curves themselves as text: episode, then per seed oracle / gpu platoon-mean episodic reward
    with open(os.path.join(args.out, "config1_platoon_mean_episodic_reward.csv"), "w") as f:
        f.write("episode," + ",".join(f"oracle_seed{s},gpu_seed{s}" for s in seeds) + "\n")
        for e in range(cmp_["episodes"]):
            f.write(str(e + 1) + "," + ",".join(f"{fx['rewards'][k, e].mean():.5f},{got[s]['rewards'][e].mean():.5f}"
                                                for k, s in enumerate(seeds)) + "\n")
    print(json.dumps(cmp_))


# ---- large runs: the headline engine at BASELINE configs[1] ----------------------------------------------------------------------
def _set_view(agents, lo, n):
    """Weight sets lo .. lo + n of an AgentGroup as a group of their own (views, nothing copied): what evaluator.run addresses."""
    import copy

    v = copy.copy(agents)
    v.theta, v.stats, v.theta_t, v.stats_t = (x[lo:lo + n] for x in (agents.theta, agents.stats, agents.theta_t, agents.stats_t))
    v.n_sets = n
    return v


def _evaluator_scores(conf, vt, platoons=(0,)):
    """workers/evaluator.py:145 score of the CURRENT actors (noise-free rollout from the evaluator's start state, 600 steps)."""
    from avddpg_amd import evaluator

    out = []
    for k in platoons:
        grp = vt.agents if vt.shared else _set_view(vt.agents, k * vt.M, vt.M)
        out.append(float(_eval_keep_rng(lambda: evaluator.run(conf=conf, actors=grp, pl_idx=k + 1, set_mod=vt.M if vt.shared else 0)[0])))
    return out


def big_run(name, conf, steps, out_dir, report=1000, eval_every=5000, auto_reset="platoon", eval_platoons=(0,), hook=None, **kw):
    """One long run of VecTrainer (device RNG). auto_reset="platoon": per-platoon episodes, curve points = means over the episodes
    closed in each reporting window (env.pop_episode_stats). auto_reset="parity": the reference's loop -- every platoon reset
    together, the episode of ALL platoons ends at the first terminal one (workers/trainer.py:246-249, 268-269), host sync per
    step; curve points = means over the window's episodes of the mean episodic reward over all agents."""
    import torch

    from avddpg_amd import trainer

    parity = auto_reset == "parity"
    vt = trainer.VecTrainer(conf, rng="device", auto_reset=False if parity else auto_reset, **kw)
    if hook is not None:
        hook(vt)
    vt.reset_episode()
    rows, t0 = [], time.perf_counter()
    ev = _evaluator_scores(conf, vt, eval_platoons)
    rows.append(dict(step=0, updates_per_set=0, episodes=0, mean_ep_reward=float("nan"), mean_ep_len=float("nan"),
                     reward_per_step=float("nan"), evaluator=ev))
    print(f"[{name}] step 0 evaluator {np.mean(ev):.3f}", flush=True)
    win_ret, win_len, win_n, ep, i = 0.0, 0.0, 0, 0, 0
    first_update = conf.batch_size + 1
    for k in range(1, steps + 1):
        if parity:
            done = vt.step(ep, i)
            i += 1
            if done or i >= conf.steps_per_episode:
                win_ret += float(vt.ep_reward.mean())
                win_len += i
                win_n += 1
                ep, i = ep + 1, 0
                vt.episode = ep
                vt.reset_episode()
        else:
            vt.step()
        if k % report == 0 or k == steps:
            if not parity:
                r, ln, n = vt.env.pop_episode_stats()
            else:
                r, ln, n = (win_ret / win_n, win_len / win_n, win_n) if win_n else (float("nan"), float("nan"), 0)
                win_ret, win_len, win_n = 0.0, 0.0, 0
            row = dict(step=k, updates_per_set=max(0, k - first_update + 1), episodes=n, mean_ep_reward=r, mean_ep_len=ln,
                       reward_per_step=r / ln if n else float("nan"), evaluator=None)
            if k % eval_every == 0 or k == steps:
                row["evaluator"] = _evaluator_scores(conf, vt, eval_platoons)
            rows.append(row)
            e = "" if row["evaluator"] is None else f" evaluator {np.mean(row['evaluator']):.3f}"
            print(f"[{name}] step {k} ({time.perf_counter() - t0:.0f} s) episodes {n} mean episodic reward {r:.3f} mean length {ln:.1f} "
                  f"reward/step {row['reward_per_step']:.4f}{e}", flush=True)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    finite = bool(torch.isfinite(vt.agents.theta).all())
    skipped = vt.nonfinite_updates()
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"{name}_curve.csv"), "w") as f:
        f.write("step,updates_per_set,episodes_closed,mean_episodic_reward,mean_episode_length,reward_per_step,evaluator_score\n")
        for r in rows:
            e = "" if r["evaluator"] is None else f"{np.mean(r['evaluator']):.3f}"
            f.write(f"{r['step']},{r['updates_per_set']},{r['episodes']},{r['mean_ep_reward']:.5f},{r['mean_ep_len']:.2f},"
                    f"{r['reward_per_step']:.6f},{e}\n")
    evs = [(r["step"], float(np.mean(r["evaluator"]))) for r in rows if r["evaluator"] is not None]
    pts = [r for r in rows if r["episodes"]]
    summary = dict(name=name, platoons=conf.num_platoons, pl_size=conf.pl_size, fed_method=conf.fed_method, engine=getattr(vt, "shared_engine", None),
                   auto_reset=auto_reset, steps=steps, updates_per_set=max(0, steps - first_update + 1), wall_s=wall,
                   env_steps_per_s=conf.num_platoons * steps / wall, weights_finite=finite, nonfinite_updates_skipped=skipped,
                   evaluator_first=evs[0][1], evaluator_best=max(e for _, e in evs), evaluator_last=evs[-1][1], evaluator_curve=evs,
                   reward_per_step_first=pts[0]["reward_per_step"], reward_per_step_last=pts[-1]["reward_per_step"],
                   mean_ep_len_first=pts[0]["mean_ep_len"], mean_ep_len_last=pts[-1]["mean_ep_len"],
                   mean_ep_reward_first=pts[0]["mean_ep_reward"], mean_ep_reward_last=pts[-1]["mean_ep_reward"])
    with open(os.path.join(out_dir, f"{name}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    del vt
    torch.cuda.empty_cache()
    return summary


def use_torch_f32_engine(vt):
    """Swap VecTrainer's shared-set learner AND its acting path for the plain-PyTorch float32 reference (tools/torch_set_learn.py):
    the f32 trainer a bf16 engine's curve is compared with at widths where no exact-f32 HIP engine exists (hidden 1024)."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch_set_learn as tsl

    P, M = vt.P, vt.M

    def learn(s, a, r, s2, weights=None):
        if weights is not None:
            raise ValueError("torch f32 engine: unweighted federated mean only")
        if getattr(vt, "set_grads", None) is None:
            vt.set_grads = torch.zeros(M, vt.agents.lay.theta_size, dtype=torch.float32, device=vt.device)
            vt.set_losses = torch.zeros(M, 2, dtype=torch.float32, device=vt.device)
        tsl.learn_sets(vt.agents, s, a, r, s2, P * M, grads=vt.set_grads, losses=vt.set_losses)

    vt._learn_batched = learn
    vt.agents.actor_shared = lambda sm, n: tsl.act_sets(vt.agents, sm.transpose(0, 1).reshape(n, -1), n,
                                                        torch.empty(n, device=vt.device)).view(P, M).transpose(0, 1).contiguous()
    vt.shared_engine_note = "torch-f32"
    return vt


def cmd_big(args):
    from avddpg_amd import config

    out = {}
    base = dict(pl_size=args.pl_size, weighted_average_enabled=False, buffer_size=args.buffer_size)
    runs = args.runs
    if "interfrl" in runs:
        conf = config.Config(num_platoons=args.platoons, fed_method="interfrl", **base)
        out["interfrl_platoon"] = big_run(f"interfrl_{args.platoons}x{args.pl_size}_fused3_per_platoon_episodes", conf, args.steps, args.out,
                                          report=args.report, eval_every=args.eval_every, auto_reset="platoon", shared_engine="fused3")
    if "parity" in runs:
        conf = config.Config(num_platoons=args.platoons, fed_method="interfrl", **base)
        out["interfrl_parity"] = big_run(f"interfrl_{args.platoons}x{args.pl_size}_fused3_any_terminal_episodes", conf, args.parity_steps,
                                         args.out, report=args.report, eval_every=args.eval_every, auto_reset="parity", shared_engine="fused3")
    if "nofrl" in runs:
        conf = config.Config(num_platoons=args.nofrl_platoons, fed_method="normal", **base)
        out["nofrl"] = big_run(f"nofrl_{args.nofrl_platoons}x{args.pl_size}_per_platoon_episodes", conf, args.nofrl_steps, args.out,
                               report=args.report, eval_every=args.eval_every, auto_reset="platoon", fused_update=True,
                               eval_platoons=tuple(range(min(8, args.nofrl_platoons))))
    if "weighted" in runs:  # r06: the Config-default weighted federated mean with the weights computed on the device (per-platoon episodes)
        conf = config.Config(num_platoons=args.platoons, fed_method="interfrl", **{**base, "weighted_average_enabled": True})
        out["interfrl_weighted"] = big_run(f"interfrl_{args.platoons}x{args.pl_size}_fused3_weighted_per_platoon_episodes", conf, args.steps, args.out,
                                           report=args.report, eval_every=args.eval_every, auto_reset="platoon", shared_engine="fused3")
    if "intrafrl" in runs:  # r06: every agent stepping with its platoon's mean gradient (avd_adam_polyak_intra_f32, platoon-chunk pipeline)
        conf = config.Config(num_platoons=args.platoons, fed_method="intrafrl", **base)
        out["intrafrl"] = big_run(f"intrafrl_{args.platoons}x{args.pl_size}_per_platoon_episodes", conf, args.intra_steps, args.out,
                                  report=args.report, eval_every=args.eval_every, auto_reset="platoon", pipeline_chunks=16,
                                  eval_platoons=tuple(range(min(8, args.platoons))))
    if "wide_bf16" in runs:  # the bf16 engine alone at BASELINE configs[4]'s full size (the float32 PyTorch reference is too slow there)
        conf = config.Config(num_platoons=args.wide_platoons, fed_method="interfrl", actor_layer1_size=1024, actor_layer2_size=1024,
                             critic_layer1_size=1024, critic_layer2_size=1024, **base)
        out["wide_bf16"] = big_run(f"interfrl_{args.wide_platoons}x{args.pl_size}_hidden1024_bf16_engine", conf, args.wide_steps, args.out,
                                   report=args.report, eval_every=args.eval_every, auto_reset="platoon", shared_engine="batched")
    if "wide" in runs:
        # BASELINE configs[4]'s width (hidden 1024): the bf16 layer-wise engine (csrc/wide.hip) against the float32 PyTorch reference
        # trainer on the same Philox streams (tie the 6 % gradient tolerance of the bf16 operands to an outcome)
        mk = lambda: config.Config(num_platoons=args.wide_platoons, fed_method="interfrl", actor_layer1_size=1024, actor_layer2_size=1024,
                                   critic_layer1_size=1024, critic_layer2_size=1024, **base)
        kw = dict(report=args.report, eval_every=args.eval_every, auto_reset="platoon", shared_engine="batched")
        out["wide_bf16"] = big_run(f"interfrl_{args.wide_platoons}x{args.pl_size}_hidden1024_bf16_engine", mk(), args.wide_steps, args.out, **kw)
        out["wide_f32"] = big_run(f"interfrl_{args.wide_platoons}x{args.pl_size}_hidden1024_torch_f32_reference", mk(), args.wide_steps, args.out,
                                  hook=use_torch_f32_engine, **kw)
    path = os.path.join(args.out, "big_summary.json")  # (runs of earlier calls stay in the file)
    merged = json.load(open(path)) if os.path.exists(path) else {}
    merged.update(out)
    with open(path, "w") as f:
        json.dump(merged, f, indent=1)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    o = sub.add_parser("oracle")
    o.add_argument("--episodes", type=int, default=100)
    o.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3, 4, 5])
    o.add_argument("--fixture", default=FIXTURE)
    o.add_argument("--max-steps", type=int, default=None)
    o.add_argument("--log", default=None)
    g = sub.add_parser("gpu")
    g.add_argument("--episodes", type=int, default=100)
    g.add_argument("--seeds", type=int, nargs="+", default=None)
    g.add_argument("--fixture", default=FIXTURE)
    g.add_argument("--serial", action="store_true")
    g.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_training_curves"))
    b = sub.add_parser("big")
    b.add_argument("--runs", nargs="+", default=["interfrl", "parity", "nofrl"])
    b.add_argument("--platoons", type=int, default=4096)
    b.add_argument("--pl-size", type=int, default=5)
    b.add_argument("--buffer-size", type=int, default=100000)
    b.add_argument("--steps", type=int, default=50100, help="interfrl, per-platoon episodes: >= 50 k updates per weight set")
    b.add_argument("--parity-steps", type=int, default=20000)
    b.add_argument("--nofrl-platoons", type=int, default=512)
    b.add_argument("--nofrl-steps", type=int, default=50100)
    b.add_argument("--intra-steps", type=int, default=30000)
    b.add_argument("--wide-platoons", type=int, default=256)
    b.add_argument("--wide-steps", type=int, default=30000)
    b.add_argument("--report", type=int, default=1000)
    b.add_argument("--eval-every", type=int, default=5000)
    b.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_training_curves"))
    args = ap.parse_args()
    {"oracle": cmd_oracle, "gpu": cmd_gpu, "big": cmd_big}[args.cmd](args)


if __name__ == "__main__":
    main()
