#!/usr/bin/env python3
"""Round-4 measurement behind the split-operand learner's test tolerances (tests/test_gpu_fsplit.py, test_gpu_configs_full.py):
  part A  unconditioned inputs at the parity-test sizes: per-tensor error of avd_learn_set_split_f16x3 and of the float32 oracle
          against the float64 oracle;
  part B  4096 x 5 (BASELINE configs[1]): the same for two whole sets (262 144 rows each);
  part C  VecTrainer 4096 x 5 interfrl on device Philox streams, fused3 vs per_agent: episodic rewards and weights after N steps.
usage: r04_precision_probe.py [A] [B] [C[:steps]]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from avddpg_amd import config, trainer, vec
from oracle import mlp as omlp
from tests.gpu_util import t
from tests.test_gpu_fset import NAMES, _batch
from tests.test_gpu_mlp import _nets, _perturbed_group, _relerr

parts = sys.argv[1:] or ["A", "B", "C"]


def table(grp, g, s, a, r, s2, P, M, sets, B=64):
    worst = {}
    for k in sets:
        sel = np.arange(P) * M + k
        cat = lambda x: x[sel].reshape(P * B, *x.shape[2:])
        batch = (cat(s), cat(a), cat(r)[:, None], cat(s2))
        cg, ag, _ = omlp.learn(batch, *_nets(grp, k, np.float64))
        cg32, ag32, _ = omlp.learn(batch, *_nets(grp, k, np.float32))
        gcg, gag = grp.grads_as_lists(g[k])
        for name, got, ref, r32 in zip(NAMES, gcg + gag, cg + ag, cg32 + ag32):
            e, e32 = _relerr(got, ref), _relerr(r32, ref)
            w = worst.setdefault(name, [0.0, 0.0])
            w[0], w[1] = max(w[0], e), max(w[1], e32)
    return worst


if "A" in parts:
    for S, P, M in [(4, 6, 2), (3, 5, 3), (4, 1, 1), (4, 70, 5)]:
        conf, grp = _perturbed_group(M, S=S, seed=61)
        s, a, r, s2 = _batch(np.random.RandomState(62), P * M, S)
        g = grp.learn_set_split(t(s), t(a), t(r), t(s2), P * M)
        w = table(grp, g, s, a, r, s2, P, M, range(M))
        print(f"A S={S} P={P} M={M}: worst split err {max(v[0] for v in w.values()):.2e} ({max(w, key=lambda n: w[n][0])}), "
              f"worst f32-oracle err {max(v[1] for v in w.values()):.2e}; tensors over 2e-5: "
              + str({n: (f'{v[0]:.1e}', f'{v[1]:.1e}') for n, v in w.items() if v[0] > 2e-5}), flush=True)

if "B" in parts:
    P, M, B, S = 4096, 5, 64, 4
    conf, grp = _perturbed_group(M, S=S, seed=91)
    gen = torch.Generator(device="cuda").manual_seed(92)
    rn = lambda *sh: torch.randn(*sh, device="cuda", generator=gen)
    n = P * M
    s, a, r, s2 = 1.5 * rn(n, B, S), 2.5 * (2 * torch.rand(n, B, 1, device="cuda", generator=gen) - 1), -rn(n, B).abs() * 0.3, 1.5 * rn(n, B, S)
    g = grp.learn_set_split(s, a, r, s2, n).clone()
    exact = vec.fed_mean(grp.learn(s, a, r, s2, M), P, M, method=conf.interfrl)
    t0 = time.time()
    sn, an, rn_, s2n = (x.cpu().numpy() for x in (s, a, r, s2))
    for name, gg in (("split", g), ("per_agent f32 engine", exact)):
        w = table(grp, gg, sn, an, rn_, s2n, P, M, (0, 4))
        print(f"B {name}: " + " ".join(f"{n}:{v[0]:.1e}/{v[1]:.1e}" for n, v in w.items()), flush=True)
    print(f"B oracle time {time.time() - t0:.1f} s", flush=True)

def run_engine(engine, can_term, steps):
    """engine 'per_agent_t': the exact-f32 engine with its OTHER learn kernel (learn_kernel_t instead of learn_kernel_l: the same
    exact f32 products in another summation order; a switch of the diagnostic library build) = the float32 noise floor."""
    import contextlib
    from avddpg_amd import _hip
    ctx = _hip.diag_library() if engine == "per_agent_t" else contextlib.nullcontext()
    if engine == "per_agent_t":
        os.environ["AVD_LEARN_KERNEL"] = "fast"
    with ctx:
        conf = config.Config(num_platoons=4096, pl_size=5, buffer_size=2048, fed_method="interfrl", weighted_average_enabled=False,
                             can_terminate=can_term, episode_sim_time=20.0)
        vt = trainer.VecTrainer(conf, rng="device", shared_sets=True, shared_engine=engine.replace("_t", ""), seed=5)
        eps, ends = [], []
        vt.reset_episode()
        i = 0
        t0 = time.time()
        for n_ in range(steps):
            done = vt.step(vt.episode, i)
            i += 1
            if done or i >= conf.steps_per_episode:
                eps.append(vt.ep_reward.cpu().numpy().copy())
                ends.append(n_)
                vt.episode += 1
                vt.reset_episode()
                i = 0
        torch.cuda.synchronize()
        out = (np.array(eps), ends, vt.agents.theta.cpu().numpy().copy(), int(vt.agents.step[0]), time.time() - t0, conf)
    os.environ.pop("AVD_LEARN_KERNEL", None)
    return out


for p in parts:
    if not p.startswith("C"):
        continue
    steps = int(p.split(":")[1]) if ":" in p else 2100
    for can_term in (True, False):
        res = {e: run_engine(e, can_term, steps) for e in ("per_agent", "fused3", "per_agent_t")}
        for other in ("fused3", "per_agent_t"):
            (ra, ea, tha, ua, ta, conf), (rb, eb, thb, ub, tb, _) = res["per_agent"], res[other]
            print(f"C can_terminate={can_term} steps={steps} {other} vs per_agent: updates per set {ua}/{ub}, episodes {len(ea)}/{len(eb)}, "
                  f"same episode ends: {ea == eb}; {ta:.0f} s / {tb:.0f} s", flush=True)
            ne = min(len(ea), len(eb))
            same = next((k for k in range(ne) if ea[k] != eb[k]), ne)
            if same:
                d = np.abs(ra[:same] - rb[:same])  # [episodes, P, M]
                rel_agent = d.max(axis=(1, 2)) / np.abs(ra[:same]).mean(axis=(1, 2))
                rel_mean = np.abs(ra[:same].mean(axis=1) - rb[:same].mean(axis=1)).max(axis=1) / np.abs(ra[:same].mean(axis=1)).max(axis=1)
                print(f"  episodes compared {same}: max over agents |dR| / mean|R| per episode: first {rel_agent[0]:.2e} last {rel_agent[-1]:.2e} "
                      f"max {rel_agent.max():.2e}; platoon-mean curve per vehicle index: max rel {rel_mean.max():.2e}", flush=True)
            dth = np.abs(tha - thb)
            print(f"  weights: mean |dtheta| {dth.mean():.3e} max {dth.max():.3e} (actor lr {conf.actor_lr}, critic lr {conf.critic_lr})", flush=True)
