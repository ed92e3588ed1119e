"""GPU parity of the whole loop: the vectorised trainer vs the oracle's reference-shaped Python loop
(same global legacy RNG stream, same initial weights), the shared-vs-per-agent interfrl equivalence,
and the reference-shaped object API (Platoon / OUActionNoise / ReplayBuffer / get_actor / policy /
update_target / Trainer.learn)."""
import numpy as np
import pytest
import torch

from avddpg_amd import config, ddpgagent, environment, model, noise, replaybuffer, trainer
from oracle import mlp as omlp
from oracle import platoon as oplatoon
from oracle import trainer as otrainer
from tests.gpu_util import need_gpu

pytestmark = pytest.mark.gpu


def _copy_weights_to_oracle(vt, ref):
    for p in range(vt.P):
        for m in range(vt.M):
            k = m if vt.shared else p * vt.M + m
            ref.actors[p][m] = vt.agents.get_weights(k, "actor")
            ref.critics[p][m] = vt.agents.get_weights(k, "critic")
            ref.t_actors[p][m] = vt.agents.get_weights(k, "actor", target=True)
            ref.t_critics[p][m] = vt.agents.get_weights(k, "critic", target=True)


@pytest.mark.parametrize("fed_method", ["normal", "interfrl"])
def test_vectorised_loop_matches_reference_shaped_oracle_loop(fed_method):
    P, L, steps = 2, 3, 72
    conf = config.Config(num_platoons=P, pl_size=L, buffer_size=128, fed_method=fed_method,
                         weighted_average_enabled=False)
    # --- oracle run (records the trajectory) ---
    np.random.seed(1)
    vt0 = trainer.VecTrainer(conf, rng="host", shared_sets=False)  # only to obtain identical initial weights
    ref = otrainer.RefTrainer(oplatoon.EnvParams(), P, L, seed=1, buffer_size=128, fed_method=fed_method)
    _copy_weights_to_oracle(vt0, ref)
    ref.reset_episode()
    traj = []
    for i in range(steps):
        done = ref.step()
        traj.append((ref.actions.copy(), np.array([[np.asarray(s) for s in ref.prev_states[p]] for p in range(P)]), done))
        assert not done
    next_draw_ref = np.random.normal(0, 1)
    # --- product run on the GPU, same seed ---
    np.random.seed(1)
    vt = trainer.VecTrainer(conf, rng="host", shared_sets=False)
    assert torch.equal(vt.agents.theta, vt0.agents.theta)
    vt.reset_episode()
    for i in range(steps):
        done = vt.step(0, i)
        act = vt.actions.cpu().numpy()
        x = vt.env.x.cpu().numpy()
        tol = 1e-5 if i < 65 else 2e-3  # after the first Adam steps f32 rounding is amplified by m/sqrt(v)
        assert np.allclose(act, traj[i][0], rtol=0, atol=tol * 2.5), (i, np.abs(act - traj[i][0]).max())
        assert np.allclose(x, traj[i][1], rtol=0, atol=tol * np.maximum(1.0, np.abs(traj[i][1]))), i
        assert done == traj[i][2]
    assert np.random.normal(0, 1) == next_draw_ref  # identical consumption of the global RNG stream
    assert vt.replay.buffer_counter == steps and vt.updates == (steps - 64) * P * L == ref.updates
    assert np.allclose(vt.ep_reward.cpu().numpy(), np.array(ref.ep_reward), rtol=1e-4)
    # weights after 8 updates: close to the oracle's (Adam turns 1e-4 gradient differences on tiny-gradient
    # elements into O(lr) differences, hence the two-level bound)
    for p in range(P):
        for m in range(L):
            k = p * L + m
            for which, refw, lr in (("actor", ref.actors[p][m], conf.actor_lr), ("critic", ref.critics[p][m], conf.critic_lr)):
                for got, want in zip(vt.agents.get_weights(k, which), refw):
                    d = np.abs(got - want)
                    assert d.max() <= 2 * lr * (steps - 64) and d.mean() <= 0.05 * lr * (steps - 64), (which, d.max(), d.mean())
    if fed_method == "interfrl":  # every platoon's vehicle-m agent stays bit-identical
        th = vt.agents.theta.reshape(P, L, -1)
        assert torch.equal(th[0], th[1])
        assert not torch.equal(th[0, 0], th[0, 1])


def test_shared_sets_equal_per_agent_sets_under_interfrl():
    """interfrl + gradients with every step federated: ONE weight set per vehicle index gives bit-identical
    weights and trajectories to the reference's P x M separate agents."""
    P, L, steps = 4, 3, 70
    conf = config.Config(num_platoons=P, pl_size=L, buffer_size=128, fed_method="interfrl", weighted_average_enabled=False)
    runs = []
    for shared in (False, True):
        np.random.seed(3)
        vt = trainer.VecTrainer(conf, rng="host", shared_sets=shared)
        vt.reset_episode()
        for i in range(steps):
            vt.step(0, i)
        runs.append(vt)
    a, b = runs
    assert b.agents.n_sets == L and a.agents.n_sets == P * L
    assert torch.equal(a.env.x, b.env.x) and torch.equal(a.actions, b.actions)
    assert torch.equal(a.agents.theta.reshape(P, L, -1)[2], b.agents.theta)
    assert torch.equal(a.agents.theta_t.reshape(P, L, -1)[0], b.agents.theta_t)
    assert int(b.agents.step[0]) == steps - 64


def test_device_rng_throughput_mode_runs_and_learns_shapes():
    conf = config.Config(num_platoons=64, pl_size=5, buffer_size=256)
    vt = trainer.VecTrainer(conf, rng="device", auto_reset=True)
    vt.reset_episode()
    for _ in range(80):
        vt.step()
    torch.cuda.synchronize()
    assert vt.updates == (80 - 64) * 64 * 5 and vt.env_steps == 80 * 64
    assert torch.isfinite(vt.agents.theta).all() and torch.isfinite(vt.env.x).all()
    assert not torch.equal(vt.agents.theta[0], vt.agents.theta[1])  # independent agents diverge
    assert (vt.losses[:, 0] >= 0).all()


def test_reference_object_api_roundtrip():
    """A trainer written against the reference's object API (workers/trainer.py:71-171, 282-356) runs
    unchanged against the avddpg_amd objects, and matches the oracle objects step by step."""
    conf = config.Config(pl_size=3, buffer_size=100)
    ep = oplatoon.EnvParams()
    np.random.seed(5)
    env = environment.Platoon(conf.pl_size, conf, 0, rand_states=conf.rand_states)
    ou = [noise.OUActionNoise(mean=np.zeros(1), config=conf) for _ in range(env.num_models)]
    actor = model.get_actor(env.num_states, env.num_actions, conf.action_high, seed_int=conf.random_seed,
                            hidd_mult=env.hidden_multiplier, layer1_size=conf.actor_layer1_size,
                            layer2_size=conf.actor_layer2_size)
    critic = model.get_critic(env.num_states, env.num_actions, hidd_mult=env.hidden_multiplier,
                              layer1_size=conf.critic_layer1_size, layer2_size=conf.critic_layer2_size,
                              action_layer_size=conf.critic_act_layer_size)
    t_actor = model.get_actor(env.num_states, env.num_actions, conf.action_high, seed_int=conf.random_seed,
                              layer1_size=conf.actor_layer1_size, layer2_size=conf.actor_layer2_size)
    t_critic = model.get_critic(env.num_states, env.num_actions, layer1_size=conf.critic_layer1_size,
                                layer2_size=conf.critic_layer2_size, action_layer_size=conf.critic_act_layer_size)
    t_actor.set_weights(actor.get_weights())
    t_critic.set_weights(critic.get_weights())
    assert [w.shape for w in actor.get_weights()] == [(4, 256), (256,), (256,), (256,), (256,), (256,), (256, 128),
                                                      (128,), (128,), (128,), (128,), (128,), (128, 1), (1,)]
    assert len(critic.weights) == 20 and len(critic.trainable_variables) == 14 and len(actor.trainable_variables) == 10
    rb = replaybuffer.ReplayBuffer(conf.buffer_size, conf.batch_size, env.num_states, env.num_actions, conf.pl_size)
    # oracle twins on the same RNG stream
    st = np.random.get_state()
    np.random.seed(5)
    oenv = oplatoon.RefPlatoon(conf.pl_size, ep)
    np.random.set_state(st)
    prev = env.reset()
    np.random.set_state(st)
    oprev = oenv.reset()
    assert np.allclose(np.array(prev), np.array(oprev), rtol=1e-6)
    aw = [w.astype(np.float64) for w in actor.get_weights()]
    actions = np.zeros((env.num_models, env.num_actions))
    for step in range(70):
        for m in range(env.num_models):
            st = np.random.get_state()
            actions[m] = ddpgagent.policy(actor(np.asarray(prev[m])[None]), ou[m], conf.action_low, conf.action_high)[0]
            np.random.set_state(st)
            nz = np.random.normal(0, 1.0, size=1)  # the draw OUActionNoise made
            assert abs(actions[m, 0]) <= 2.5
        ex = np.random.normal(0, conf.reset_max_u)
        states, rewards, done = env.step(actions.flatten(), ex)
        ostates, orewards, odone = oenv.step(actions.flatten(), ex)
        assert np.allclose(np.array(states), np.array(ostates), rtol=1e-5, atol=1e-5) and done == odone
        assert np.allclose(rewards, orewards, rtol=1e-5, atol=1e-7)
        assert np.allclose(np.array(env.get_jerk()), np.array(oenv.get_jerk()), rtol=1e-3, atol=1e-4)
        rb.add((prev[0], actions[0], rewards[0], states[0]))
        prev = states
    assert rb.buffer_counter == 70 > conf.batch_size
    # the fused equivalents the north star names (avddpg_amd.ddpgagent.act / .learn): act == the per-model policy(actor(state), noise)
    # loop of workers/trainer.py:286-289 -- one actor launch + one clip launch --, with and without noise, same draws in the same order
    M_ = env.num_models
    loop = np.array([ddpgagent.policy(actor(np.asarray(prev[m])[None]), None, conf.action_low, conf.action_high)[0] for m in range(M_)])
    assert np.array_equal(ddpgagent.act([actor] * M_, prev, None, conf.action_low, conf.action_high), loop)
    twins = [noise.OUActionNoise(mean=np.zeros(1), x_init=np.array([0.01 * (m % M_ + 1)]), config=conf) for m in range(2 * M_)]
    st = np.random.get_state()
    loop = np.array([ddpgagent.policy(actor(np.asarray(prev[m])[None]), twins[m], conf.action_low, conf.action_high)[0] for m in range(M_)])
    np.random.set_state(st)
    fused = ddpgagent.act([actor] * M_, prev, twins[M_:], conf.action_low, conf.action_high)
    assert np.array_equal(fused, loop) and not np.array_equal(fused, ddpgagent.act([actor] * M_, prev, None, conf.action_low, conf.action_high))
    others = [t_actor, actor, t_actor]  # different models per row: the per-set form of the same launch
    t_actor.set_weights([w * 1.5 for w in actor.get_weights()])
    mixed = ddpgagent.act(others[:M_], prev, None, conf.action_low, conf.action_high)
    assert np.array_equal(mixed, np.array([ddpgagent.policy(others[m](np.asarray(prev[m])[None]), None, conf.action_low, conf.action_high)[0]
                                           for m in range(M_)]))
    t_actor.set_weights(actor.get_weights())
    np.random.seed(11)
    cg, ag = trainer.Trainer.learn(rb, actor, critic, t_actor, t_critic)
    np.random.seed(11)
    cg2, ag2 = ddpgagent.learn(rb, actor, critic, t_actor, t_critic)
    assert all(np.array_equal(x, y) for x, y in zip(cg + ag, cg2 + ag2))
    np.random.seed(11)
    idx = np.random.choice(70, 64)
    ring = rb._v.ring.cpu().numpy()[0]
    batch = (ring[idx, 0:4], ring[idx, 4:5], ring[idx, 5:6], ring[idx, 6:10])
    nets = [[w.astype(np.float64) for w in n.get_weights()] for n in (actor, critic, t_actor, t_critic)]
    ocg, oag, _ = omlp.learn(batch, *nets)
    for got, ref in zip(cg + ag, ocg + oag):
        assert got.shape == ref.shape and np.max(np.abs(got - ref)) <= 1e-4 * max(1e-12, np.max(np.abs(ref)))
    tc, ta = ddpgagent.update_target(conf.tau, t_critic.weights, critic.weights, t_actor.weights, actor.weights)
    otc, ota = omlp.update_target(conf.tau, t_critic.weights, critic.weights, t_actor.weights, actor.weights)
    for got, ref in zip(tc + ta, otc + ota):
        assert np.array_equal(got, ref)
    t_critic.set_weights(tc)
    assert np.array_equal(t_critic.get_weights()[0], tc[0])
    with pytest.raises(ValueError):  # reference refuses platoons longer than its colour table, after construction
        environment.Platoon(7, conf, 0)


def test_pipelined_learn_apply_is_bitwise_learn_then_apply():
    """nofrl: the two-stream agent-slice pipeline (learn || Adam+Polyak) gives exactly the serial result."""
    conf = config.Config(num_platoons=37, pl_size=3, buffer_size=128)
    runs = []
    for chunks in (1, 5):
        vt = trainer.VecTrainer(conf, rng="device", auto_reset=True, pipeline_chunks=chunks, seed=4)
        vt.reset_episode()
        for _ in range(70):
            vt.step()
        torch.cuda.synchronize()
        runs.append(vt)
    a, b = runs
    assert torch.equal(a.agents.theta, b.agents.theta) and torch.equal(a.agents.theta_t, b.agents.theta_t)
    assert torch.equal(a.agents.m, b.agents.m) and torch.equal(a.agents.v, b.agents.v)
    assert torch.equal(a.env.x, b.env.x) and torch.equal(a.grads, b.grads)
    assert int(a.agents.step[0]) == int(b.agents.step[-1]) == 6


def test_evaluator_rollout_matches_reference_golden_and_oracle():
    """Noise-free deterministic-start rollout (workers/evaluator.py): with a zero policy it reproduces the golden
    captured from the reference env (G8); with real actors it follows the oracle's rollout."""
    import os

    from avddpg_amd import evaluator, vec
    from oracle import evaluator as oeval

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_evaluator.npz"))
    for L, model, T in ((2, "ModelB", 600), (3, "ModelA", 100)):
        conf = config.Config(pl_size=L, model=model)
        grp = vec.AgentGroup(L, 3 if model == "ModelA" else 4, 1, conf)
        lay = grp.lay
        grp.theta[:, lay.aW3:lay.aW3 + lay.H2].zero_()  # zero last actor layer -> tanh(0)*high = 0
        pl_rew, tr = evaluator.run(conf=conf, actors=grp, pl_idx=1, manual_timestep_override=T)
        key = f"L{L}_{model}"
        assert abs(pl_rew - float(g[key + "__pl_rew"])) <= 2e-3 and np.all(tr["inputs"] == 0)
        ref = g[key + "__states"]
        assert np.all(np.abs(tr["states"] - ref) <= 1e-5 * T * np.maximum(1.0, np.abs(ref)))
        assert np.allclose(tr["counters"], g[key + "__counters"], rtol=2e-5)
    # trained-looking (random) actors vs the oracle rollout
    conf = config.Config(pl_size=3)
    grp = vec.AgentGroup(3, 4, 1, conf, seed=3)
    grp.theta[:, grp.lay.aW3:grp.lay.aW3 + grp.lay.H2] *= 40  # visible, unsaturated control
    actors = [[w.astype(np.float64) for w in grp.get_weights(m, "actor")] for m in range(3)]
    pl_rew, tr = evaluator.run(conf=conf, actors=grp, pl_idx=1, manual_timestep_override=100)
    o_rew, o_tr = oeval.run(oplatoon.EnvParams(), 3, actors, 100)
    assert abs(pl_rew - o_rew) <= 2e-3 and np.abs(o_tr["inputs"]).max() > 0.05
    assert np.allclose(tr["inputs"], o_tr["inputs"], atol=5e-5) and np.allclose(tr["states"], o_tr["states"], atol=2e-4, rtol=1e-4)
    assert np.allclose(tr["jerks"], o_tr["jerks"], atol=5e-3)


def test_model_a_and_weighted_interfrl_trainer_paths_run():
    """Model A (3 observed states, 4-wide internal state) through the whole loop, and the weighted interfrl
    branch (weights |1/mean(last 10 episodic rewards)|, trainer.py:385-398) once episode >= weighted_window."""
    conf = config.Config(num_platoons=6, pl_size=3, buffer_size=128, model="ModelA")
    vt = trainer.VecTrainer(conf, rng="device", auto_reset=True)
    assert vt.S == 3 and vt.agents.lay.S == 3
    vt.reset_episode()
    for _ in range(70):
        vt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(vt.agents.theta).all() and int(vt.agents.step[0]) == 6
    conf = config.Config(num_platoons=4, pl_size=2, buffer_size=128, fed_method="interfrl", weighted_average_enabled=True,
                         weighted_window=2, episode_sim_time=3.0)  # 30-step episodes
    np.random.seed(2)
    vt = trainer.VecTrainer(conf, rng="host")
    assert vt.shared
    vt.run(number_of_episodes=4)
    assert len(vt.all_ep_reward_lists[3][1]) == 4 and vt.fed_weights is not None and vt.fed_weights[0] == 3
    w = vt.fed_weights[1].cpu().numpy()
    ref = np.array([[abs(1 / np.mean(vt.all_ep_reward_lists[p][m][-2 - 1:-1])) for m in range(2)] for p in range(4)])
    assert np.allclose(w, ref, rtol=1e-6)  # weights of episode 3 come from episodes 1-2
    assert torch.isfinite(vt.agents.theta).all() and int(vt.agents.step[0]) == 4 * 30 - 64


@pytest.mark.parametrize("model,kernel", [("ModelB", "lean"), ("ModelA", "lean"), ("ModelB", "fast"), ("ModelA", "fast"),
                                          ("ModelB", "general"), ("ModelB", "centralized")])
def test_fused_learn_update_is_bitwise_learn_then_apply(model, kernel, monkeypatch, request):
    """avd_learn_update_f32 (Adam + Polyak applied where each gradient is produced, theta ping-pong) gives exactly
    the weights, targets and moments of avd_learn_f32 followed by avd_adam_polyak_f32 -- for learn_kernel_l (default),
    learn_kernel_t (AVD_LEARN_KERNEL=fast) and the general kernel (AVD_LEARN_GENERAL=1 at the reference widths; the
    centralized framework: S = 12, A = 3, widths 307/153/57 padded to 320/160/64)."""
    if kernel in ("fast", "general"):  # kernel-variant switches exist in the diagnostic build only
        request.getfixturevalue("diag_lib")
    if kernel == "fast":
        monkeypatch.setenv("AVD_LEARN_KERNEL", "fast")
    if kernel == "general":
        monkeypatch.setenv("AVD_LEARN_GENERAL", "1")
    conf = config.Config(num_platoons=21, pl_size=3, buffer_size=128, model=model,
                         framework="centralized" if kernel == "centralized" else "decentralized")
    runs = []
    for fused in (False, True):
        vt = trainer.VecTrainer(conf, rng="device", auto_reset=True, fused_update=fused, seed=4)
        vt.reset_episode()
        for _ in range(70):
            vt.step()
        torch.cuda.synchronize()
        runs.append(vt)
    a, b = runs
    assert int(a.agents.step[0]) == int(b.agents.step[-1]) == 6
    assert torch.equal(a.agents.m, b.agents.m) and torch.equal(a.agents.v, b.agents.v)
    assert torch.equal(a.agents.theta, b.agents.theta) and torch.equal(a.agents.theta_t, b.agents.theta_t)
    assert torch.equal(a.agents.stats_t, b.agents.stats_t) and torch.equal(a.env.x, b.env.x)
    assert torch.equal(a.losses, b.losses)
    lay = b.agents.lay
    assert torch.all(b.agents.theta[:, lay.ab3 + lay.A:lay.actor_size] == 0)  # padding untouched in both slabs
    assert torch.all(b.agents.theta_alt[:, lay.ab3 + lay.A:lay.actor_size] == 0)


def test_intrafrl_directional_and_weights_aggregation_semantics():
    """intrafrl gradients with intra_directional_averaging (lead vehicle skipped on federated steps,
    trainer.py:417-418) and aggregation_method='weights' (group [0]'s average written into every model and target,
    :442-456)."""
    P, M = 3, 3
    # --- intrafrl + gradients, directional: followers of a platoon share the platoon-mean gradient, leader frozen
    conf = config.Config(num_platoons=P, pl_size=M, buffer_size=128, fed_method="intrafrl", weighted_average_enabled=False,
                         intra_directional_averaging=True)
    np.random.seed(7)
    vt = trainer.VecTrainer(conf, rng="host")
    vt.reset_episode()
    th0 = vt.agents.theta.clone()
    for i in range(66):
        vt.step(0, i)
    th = vt.agents.theta.view(P, M, -1)
    assert torch.equal(th[:, 0], th0.view(P, M, -1)[:, 0])  # lead vehicles untouched
    assert int(vt.agents.step.view(P, M)[0, 0]) == 0 and int(vt.agents.step.view(P, M)[0, 1]) == 2
    assert torch.equal(th[0, 1], th[0, 2]) and not torch.equal(th[0, 1], th[1, 1])  # same mean grad inside a platoon
    g = vt.grads.view(P, M, -1)
    # (r06: the platoon's mean is formed inside the Adam pass, avd_adam_polyak_intra_f32 -- the slab keeps the agents' OWN gradients;
    #  the followers' identical moments show that they stepped with one and the same mean)
    assert not torch.equal(g[1, 0], g[1, 2])
    mm = vt.agents.m.view(P, M, -1)
    assert torch.equal(mm[1, 1], mm[1, 2]) and mm[1, 1].abs().max() > 0 and (mm[:, 0] == 0).all()
    # --- interfrl + weights aggregation with delay 2: odd steps train locally, even steps overwrite everything
    conf = config.Config(num_platoons=P, pl_size=M, buffer_size=128, fed_method="interfrl", weighted_average_enabled=False,
                         aggregation_method="weights", fed_update_delay=0.2)
    assert conf.fed_update_delay_steps == 2
    np.random.seed(7)
    vt = trainer.VecTrainer(conf, rng="host")
    assert not vt.shared
    vt.reset_episode()
    for i in range(65):  # steps 0..64; first learn at i=64 (even: federated weights step, no local update before it)
        vt.step(0, i)
    assert torch.allclose(vt.agents.theta, th0, rtol=2e-7, atol=0)  # mean of identical weights: (3w)/3 rounds, no training yet
    vt.step(0, 65)  # odd step: local Adam update -> agents diverge
    local = vt.agents.theta.clone().view(P, M, -1)
    assert not torch.equal(local[0, 0], local[1, 0])
    vt.step(0, 66)  # even step: weights aggregation
    after = vt.agents.theta.view(P, M, -1)
    want = local[:, 0].mean(dim=0)  # vehicle 0 averaged over platoons == group [0]
    assert torch.allclose(after[2, 1], want, rtol=1e-6, atol=1e-8)
    assert torch.equal(after[0, 0], after[2, 2]) and torch.equal(vt.agents.theta_t, vt.agents.theta)


def test_episode_reward_curve_matches_oracle_loop_config1():
    """BASELINE configs[0] (1 platoon x 3 vehicles, DDPG): several full episodes of training through the
    reference-shaped facade (Trainer.initialize/run, host RNG) against the oracle's per-object loop: the per-episode
    cumulative rewards (the `ep_reward` curve of the reference) agree to 1e-4 relative after ~1000 updates/agent."""
    P, L, EPS, STEPS = 1, 3, 3, 150
    conf = config.Config(num_platoons=P, pl_size=L, buffer_size=1000, episode_sim_time=STEPS * 0.1)
    np.random.seed(conf.random_seed)
    tr = trainer.Trainer(None, "t0", False, conf, rng="host")
    tr.initialize()
    ref = otrainer.RefTrainer(oplatoon.EnvParams(), P, L, seed=conf.random_seed, buffer_size=1000)
    _copy_weights_to_oracle(tr.engine, ref)
    curve = []
    for ep in range(EPS):
        ref.reset_episode()
        for i in range(STEPS):
            if ref.step():
                break
        curve.append(np.array(ref.ep_reward).copy())
    np.random.seed(conf.random_seed)
    tr = trainer.Trainer(None, "t0", False, conf, rng="host")
    tr.initialize()
    ep_lists, avg_lists = tr.run(number_of_episodes=EPS)
    for ep in range(EPS):
        for m in range(L):
            want = float(curve[ep][0][m])
            assert abs(float(ep_lists[0][m][ep]) - want) <= 1e-4 * abs(want), (ep, m)
    assert np.isclose(avg_lists[0][1][-1], np.mean([c[0][1] for c in curve]), rtol=1e-4)  # trailing-mean curve
    assert tr.engine.updates == ref.updates


@pytest.mark.parametrize("framework", ["decentralized", "centralized"])
def test_cli_train_then_esim_roundtrip(tmp_path, capsys, framework):
    """`python -m avddpg_amd tr` writes reward CSVs, conf.json and Keras-ordered checkpoints; `esim` reloads the
    actors from them and reproduces the evaluator reward of the in-memory agents. Centralized runs save ONE model per
    platoon (S = 4L, A = L, widths x1.2) and esim must rebuild exactly that shape."""
    import glob
    import os

    from avddpg_amd import __main__ as cli
    from avddpg_amd import artifacts, evaluator
    from avddpg_amd.config import Config

    cent = framework == "centralized"
    # the reference CLI has no --framework flag (src/cmd/api.py:60-81): it is a Config field
    cli.main(["tr", "--pl_num", "2", "--pl_size", "2", "--total_time_steps", "1200", "--buffer_size", "500", "--out",
              str(tmp_path)], conf=Config(framework=framework))
    base = capsys.readouterr().out.strip().splitlines()[-1]
    assert os.path.exists(os.path.join(base, "ep_reward__seed1.csv")) and os.path.exists(os.path.join(base, "conf.json"))
    assert len(glob.glob(os.path.join(base, "*.npz"))) == 2 * (1 if cent else 2) * 4
    w = artifacts.load_actor_weights(base, 2, 1)
    assert [x.shape for x in w][:3] == ([(8, 307), (307,), (307,)] if cent else [(4, 256), (256,), (256,)]) and len(w) == 14
    cli.main(["esim", base, "--n_timesteps", "60"])
    out = capsys.readouterr().out
    assert "platoon 1: cumulative platoon reward" in out and "platoon 2:" in out
    conf = artifacts.config_loader(os.path.join(base, "conf.json"), Config)
    assert conf.number_of_episodes == 2 and conf.num_platoons == 2


@pytest.mark.parametrize("L", [1, 3])
def test_centralized_loop_matches_oracle_loop(L):
    """Centralized framework: one model per platoon, S = 4L, A = L, widths x1.2 (307/153/57, padded
    to 320/160/64 in HBM), platoon-mean reward, one scalar OU process broadcast over the L actions. L = 1 is the
    case the reference trainer itself completes (trainer.py:45 iterates pl_size models); L = 3 follows its
    evaluator's env.num_models loop (evaluator.py:48-91)."""
    P, steps = 2, 70
    conf = config.Config(num_platoons=P, pl_size=L, buffer_size=128, framework="centralized")
    np.random.seed(5)
    vt0 = trainer.VecTrainer(conf, rng="host")
    assert (vt0.M, vt0.S, vt0.A) == (1, 4 * L, L) and tuple(vt0.agents.dims)[2:] == (307, 153, 57)
    ref = otrainer.RefTrainer(oplatoon.EnvParams(framework="centralized"), P, L, seed=5, buffer_size=128)
    _copy_weights_to_oracle(vt0, ref)
    assert ref.actors[0][0][0].shape == (4 * L, 307) and ref.critics[0][0][18].shape == (153, L)
    ref.reset_episode()
    traj = []
    for i in range(steps):
        done = ref.step()
        traj.append((ref.actions.copy(), np.array([np.asarray(ref.prev_states[p][0]) for p in range(P)]), done,
                     np.array([ref.ep_reward[p][0] for p in range(P)])))
        assert not done
    next_draw_ref = np.random.normal(0, 1)
    np.random.seed(5)
    vt = trainer.VecTrainer(conf, rng="host")
    assert torch.equal(vt.agents.theta, vt0.agents.theta)
    vt.reset_episode()
    for i in range(steps):
        done = vt.step(0, i)
        act = vt.actions.cpu().numpy()
        x = vt.env.x.cpu().numpy().reshape(P, 4 * L)
        tol = 1e-5 if i < 65 else 2e-3
        assert np.allclose(act, traj[i][0], rtol=0, atol=tol * 2.5), (i, np.abs(act - traj[i][0]).max())
        assert np.allclose(x, traj[i][1], rtol=0, atol=tol * np.maximum(1.0, np.abs(traj[i][1]))), i
        assert done == traj[i][2]
        assert np.allclose(vt.ep_reward.cpu().numpy()[:, 0], traj[i][3], rtol=1e-4, atol=1e-6)
    assert np.random.normal(0, 1) == next_draw_ref
    assert vt.updates == (steps - 64) * P == ref.updates
    for p in range(P):
        for which, refw, lr in (("actor", ref.actors[p][0], conf.actor_lr), ("critic", ref.critics[p][0], conf.critic_lr)):
            for got, want in zip(vt.agents.get_weights(p, which), refw):
                assert got.shape == want.shape
                d = np.abs(got - want)
                assert d.max() <= 2 * lr * (steps - 64) and d.mean() <= 0.05 * lr * (steps - 64), (which, d.max(), d.mean())
    # the zero padding of the slabs survives Adam and Polyak
    lay = vt.agents.lay
    for slab in (vt.agents.theta, vt.agents.theta_t):
        w1 = slab[:, lay.aW1:lay.aW1 + 4 * L * 320].reshape(P, 4 * L, 320)
        assert torch.all(w1[:, :, 307:] == 0) and torch.any(w1[:, :, :307] != 0)
    with pytest.raises(ValueError, match="Model A"):
        trainer.VecTrainer(config.Config(num_platoons=1, pl_size=2, framework="centralized", model="ModelA"), rng="device")


def test_centralized_evaluator_rollout_matches_reference_golden_and_oracle():
    """workers/evaluator.py with env.num_models = 1: zero policy vs the golden captured from the reference env (G9),
    a non-trivial centralized actor vs the oracle rollout."""
    import os

    from avddpg_amd import evaluator, vec
    from oracle import evaluator as oeval

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_evaluator_centralized.npz"))
    for L, T in ((3, 200), (1, 100)):
        conf = config.Config(pl_size=L, framework="centralized")
        grp = vec.AgentGroup(1, 4 * L, L, conf, hidd_mult=conf.centrl_hidd_mult)
        lay = grp.lay
        grp.theta[:, lay.aW3:lay.aW3 + lay.H2 * L].zero_()
        pl_rew, tr = evaluator.run(conf=conf, actors=grp, pl_idx=1, manual_timestep_override=T)
        key = f"L{L}_ModelB_centralized"
        assert abs(pl_rew - float(g[key + "__pl_rew"])) <= 2e-3 and np.all(tr["inputs"] == 0)
        ref = g[key + "__states"].reshape(T, L, 4)
        assert np.all(np.abs(tr["states"] - ref) <= 1e-5 * T * np.maximum(1.0, np.abs(ref)))
        assert np.allclose(tr["counters"], g[key + "__counters"], rtol=2e-5)
    L = 3
    conf = config.Config(pl_size=L, framework="centralized")
    grp = vec.AgentGroup(1, 4 * L, L, conf, seed=3, hidd_mult=conf.centrl_hidd_mult)
    grp.theta[:, grp.lay.aW3:grp.lay.aW3 + grp.lay.H2 * L] *= 40
    actors = [[w.astype(np.float64) for w in grp.get_weights(0, "actor")]]
    pl_rew, tr = evaluator.run(conf=conf, actors=grp, pl_idx=1, manual_timestep_override=100)
    o_rew, o_tr = oeval.run(oplatoon.EnvParams(framework="centralized"), L, actors, 100)
    assert abs(pl_rew - o_rew) <= 2e-3 and np.abs(o_tr["inputs"]).max() > 0.05
    assert np.allclose(tr["inputs"], o_tr["inputs"], atol=5e-5)
    assert np.allclose(tr["states"], o_tr["states"].reshape(100, L, 4), atol=2e-4, rtol=1e-4)


def test_initial_weights_do_not_depend_on_the_stream_seed():
    """Multi-GPU interfrl: every rank passes its own stream seed (env / noise / replay) but must start from the SAME
    weights -- the reference starts every agent from agent (0,0)'s (workers/trainer.py:121-131) and the federated
    arithmetic assumes identical sets on all ranks. The init seed is conf.random_seed, independent of `seed`."""
    conf = config.Config(num_platoons=6, pl_size=3, buffer_size=128, fed_method="interfrl", weighted_average_enabled=False)
    a = trainer.VecTrainer(conf, rng="device", auto_reset=True, seed=1)
    b = trainer.VecTrainer(conf, rng="device", auto_reset=True, seed=2)
    assert torch.equal(a.agents.theta, b.agents.theta) and torch.equal(a.agents.theta_t, b.agents.theta_t)
    assert torch.equal(a.agents.stats, b.agents.stats) and a.total_platoons == 6.0
    a.reset_episode(), b.reset_episode()
    assert not torch.equal(a.env.x, b.env.x)  # the streams do differ
    c = trainer.VecTrainer(conf, rng="device", auto_reset=True, seed=1, init_seed=7)
    assert not torch.equal(a.agents.theta, c.agents.theta)


@pytest.mark.parametrize("mode,engine", [(True, "fused3"), (True, "per_agent"), ("platoon", "fused3"), (True, "batched")])
def test_weighted_federated_averaging_with_the_episode_bookkeeping_on_the_device(mode, engine):
    """`weighted_average_enabled` -- the Config default (src/config.py:28) -- in the throughput modes. The weights
    |1 / mean(last `weighted_window` episodic rewards)| (workers/trainer.py:385-398) come from a device ring of closed-episode
    rewards (avd_fed_history_push_f32 + avd_fed_weights_f32), no host synchronisation per step. A host shadow follows the run step
    by step -- float32 episodic counters as the reference's (:249, 321), closed where the device closes episodes -- and must find:
    the ring holds exactly (bitwise) the last closed episodes' rewards; before `weighted_window` episodes the update is the plain
    mean (all factors 1, like `training_episode < weighted_window`, :694); from then on w, the per-set sums and the learners'
    factors w P / sum(w) equal the reference formula on the shadow's lists to float32 rounding (1e-6: the mean's summation order);
    and the learn call fed from the device equals the one fed the host's weights (1e-5 of each block's max)."""
    need_gpu()
    P, L, W = 6, 3, 3
    conf = config.Config(num_platoons=P, pl_size=L, buffer_size=256, fed_method="interfrl", weighted_average_enabled=True,
                         weighted_window=W, episode_sim_time=2.5)  # 25-step episodes
    vt = trainer.VecTrainer(conf, rng="device", auto_reset=mode, shared_engine=engine, shared_sets=True)
    assert vt._dev_weighted and vt.shared_engine == engine
    vt.reset_episode()
    acc = np.zeros((P, L), dtype=np.float32)
    lists = [[[] for _ in range(L)] for _ in range(P)]
    cnt_prev = np.zeros(P, dtype=np.int64)
    checked = 0
    for i in range(140):
        vt.step()
        acc += vt.env.reward.cpu().numpy()  # float32 adds in step order: what the fused step's counters do
        cnt = vt._hist_cnt.cpu().numpy().astype(np.int64)
        for p in np.nonzero(cnt != cnt_prev)[0]:
            assert cnt[p] == cnt_prev[p] + 1
            for m in range(L):
                lists[p][m].append(acc[p, m])
            acc[p] = 0.0
        if mode is True:
            assert len(set(cnt.tolist())) == 1  # the all-platoons rule: every platoon is in the same episode
        cnt_prev = cnt
        ring = vt._hist_ring.cpu().numpy().reshape(P, L, W)
        for p in range(P):
            for m in range(L):
                for k, val in enumerate(lists[p][m][-W:]):
                    e = len(lists[p][m]) - len(lists[p][m][-W:]) + k
                    assert ring[p, m, e % W] == val, (i, p, m, e)
        aw, ws, wr = vt._aw.cpu().numpy().reshape(P, L), vt._wsum.cpu().numpy(), vt._w_raw.cpu().numpy().reshape(P, L)
        enabled = (cnt.min() >= W) if mode is True else (vt.steps_total >= W * conf.steps_per_episode)
        if not enabled:
            assert (aw == 1).all() and (ws == P).all() and (wr == 1).all()
        else:
            want = np.array([[abs(1 / np.mean(lists[p][m][-W:])) for m in range(L)] for p in range(P)], dtype=np.float32)
            assert np.allclose(wr, want, rtol=2e-6) and np.allclose(ws, want.sum(axis=0), rtol=2e-6)
            assert np.allclose(aw, want * (P / want.sum(axis=0)), rtol=3e-6)
            checked += 1
    assert checked >= 30 and cnt_prev.min() >= W + 1
    assert torch.isfinite(vt.agents.theta).all() and int(vt.agents.step[0]) >= 70
    if engine in ("fused3", "batched"):  # the same learn call from the device factors and from the host's weights
        s_, a_, r_, s2_ = vt.replay.sample()
        vt._learn_batched(s_, a_, r_, s2_, "device")
        g_dev = vt.set_grads.clone()
        vt._learn_batched(s_, a_, r_, s2_, vt._w_raw.view(P, L).clone())
        A = vt.agents.lay.actor_size
        for lo, hi in ((0, A), (A, vt.agents.lay.theta_size)):
            d = (g_dev[:, lo:hi] - vt.set_grads[:, lo:hi]).abs().max().item()
            assert d <= 1e-5 * vt.set_grads[:, lo:hi].abs().max().item(), (lo, d)


def test_host_episode_loop_still_needs_recorded_episodes_for_its_weights():
    """The host episode loop (auto_reset=False) computes the weights from its per-episode lists: without any recorded episode it
    must fail loudly, not poison the weights with NaN."""
    need_gpu()
    conf = config.Config(num_platoons=4, pl_size=2, buffer_size=128, fed_method="interfrl", weighted_average_enabled=True)
    vt = trainer.VecTrainer(conf, rng="device", auto_reset=False)
    with pytest.raises(RuntimeError, match="episodic rewards"):
        vt._weights_for_fed(conf.weighted_window)


def test_device_rng_leader_exog_honours_rand_gen():
    """util.get_random_val(conf.rand_gen, reset_max_u) (trainer.py:291-295): the device-RNG path draws U(-u, u) when
    rand_gen == 'uniform' (avd_uniform_f32, bit-checked against the oracle's Philox restatement), N(0, u) otherwise."""
    from oracle import philox as ophilox

    for gen in ("uniform", "normal"):
        conf = config.Config(num_platoons=8192, pl_size=2, buffer_size=128, rand_gen=gen)
        vt = trainer.VecTrainer(conf, rng="device", auto_reset=True, seed=9)
        vt.reset_episode()
        vt.step()
        ex = vt.leader_exog.cpu().numpy()
        u = conf.reset_max_u
        if gen == "uniform":
            assert np.abs(ex).max() <= u and abs(ex.std() - u / np.sqrt(3)) < 0.02 * u and abs(ex.mean()) < 0.05 * u
            want = ophilox.uniforms(8192, 9, 0, ophilox.STREAM_NORMAL, u)
            assert np.array_equal(ex, want)
        else:
            assert np.abs(ex).max() > u and abs(ex.std() / u - 1) < 0.05


def test_fused_update_next_action_epilogue_survives_episode_resets():
    """fused_update has the learn kernel leave actor(next state) of the UPDATED weights in actor_out, and the next step's
    actor launch runs only if the episode ended in between (device flag). With tight termination bounds episodes end
    every few steps: the trajectory must stay bit-identical to the two-kernel path, which calls the actor every step."""
    conf = config.Config(num_platoons=33, pl_size=3, buffer_size=128, max_ep=5.5, max_ev=5.5)
    runs = []
    for fused in (False, True):
        vt = trainer.VecTrainer(conf, rng="device", auto_reset=True, fused_update=fused, seed=11)
        vt.reset_episode()
        ends = 0
        for _ in range(140):
            vt.step()
            ends += int(vt.env.done.any().item())
        torch.cuda.synchronize()
        runs.append((vt, ends))
    (a, ea), (b, eb) = runs
    assert ea == eb and 5 <= ea <= 135, ea          # episodes do end, but not on every step: both branches are exercised
    assert b._act_ready and not a._act_ready
    assert torch.equal(a.env.x, b.env.x) and torch.equal(a.actions, b.actions)  # same actions taken on every step
    assert torch.equal(a.agents.theta, b.agents.theta) and torch.equal(a.agents.theta_t, b.agents.theta_t)
    # b.actor_out already holds the NEXT step's actor outputs (unless that step must recompute them after a reset)
    nxt = a.agents.actor(a.env.x.view(-1, a.x_stride), 0, x_stride=a.x_stride)
    assert bool(b.env.done.any()) or torch.equal(b.actor_out.view(-1), nxt)


FRL_VARIANTS = {
    # SURVEY 8 f-2, each against the oracle loop (oracle/trainer.py) on the same host RNG stream
    "interfrl_weighted_gradients": dict(fed_method="interfrl", weighted_average_enabled=True, weighted_window=2),
    "interfrl_weights_delay2": dict(fed_method="interfrl", weighted_average_enabled=False, aggregation_method="weights",
                                    fed_update_delay=0.2),
    "intrafrl_directional_gradients": dict(fed_method="intrafrl", weighted_average_enabled=False, intra_directional_averaging=True),
    "intrafrl_weighted_weights_delay3": dict(fed_method="intrafrl", weighted_average_enabled=True, weighted_window=2,
                                             aggregation_method="weights", fed_update_delay=0.3),
    "interfrl_update_count2_cutoff": dict(fed_method="interfrl", weighted_average_enabled=False, fed_update_count=2,
                                          fed_update_delay=0.2),
}


@pytest.mark.parametrize("variant", sorted(FRL_VARIANTS))
def test_frl_variants_match_the_oracle_loop(variant):
    """Weighted FRL (workers/trainer.py:385-398: |1 / mean(last `window` episodic rewards)|, from episode >= window), weights
    aggregation (:433-456: group [0]'s average into every model and target), intrafrl + directional (:417-418), delays and
    update counts with the local-update-gate quirk (:345 vs :680): VecTrainer's episode loop against the oracle's
    per-object loop, four 45-step episodes (updates from episode 1 on), identical consumption of the global RNG stream."""
    kw = FRL_VARIANTS[variant]
    P, L, episodes = 2, 2, 4
    conf = config.Config(num_platoons=P, pl_size=L, buffer_size=256, episode_sim_time=4.5, **kw)
    T = conf.steps_per_episode
    assert T == 45
    np.random.seed(4)
    vt0 = trainer.VecTrainer(conf, rng="host", shared_sets=False)  # only to obtain identical initial weights
    ref = otrainer.RefTrainer(oplatoon.EnvParams(), P, L, seed=4, buffer_size=256, fed_method=conf.fed_method,
                              aggregation_method=conf.aggregation_method, weighted_average_enabled=conf.weighted_average_enabled,
                              weighted_window=conf.weighted_window, fed_update_count=conf.fed_update_count,
                              fed_cutoff_episode=conf.fed_cutoff_episode, fed_update_delay_steps=conf.fed_update_delay_steps,
                              intra_directional_averaging=conf.intra_directional_averaging, steps_per_episode=T)
    _copy_weights_to_oracle(vt0, ref)
    traj = []
    for ep in range(episodes):
        ref.reset_episode()
        for i in range(T):
            done = ref.step(ep, i)
            traj.append((ep, i, ref.actions.copy(), np.array([[np.asarray(s) for s in ref.prev_states[p]] for p in range(P)]), done))
            if done:
                break
        ref.update_reward_list()
    next_draw_ref = np.random.normal(0, 1)
    np.random.seed(4)
    vt = trainer.VecTrainer(conf, rng="host", shared_sets=False)
    k = 0
    for ep in range(episodes):
        vt.episode = ep
        vt.reset_episode()
        for i in range(T):
            done = vt.step(ep, i)
            e_, i_, act, x, d_ = traj[k]
            k += 1
            assert (e_, i_, d_) == (ep, i, done)
            tol = 1e-5 if vt.updates == 0 or (ep == 1 and i <= 19) else 5e-3
            assert np.abs(vt.actions.cpu().numpy() - act).max() <= tol * 2.5, (ep, i, np.abs(vt.actions.cpu().numpy() - act).max())
            got = vt.env.x.cpu().numpy()
            assert np.all(np.abs(got - x) <= tol * np.maximum(1.0, np.abs(x))), (ep, i)
            if done:
                break
        vt.update_reward_list(ep)
    assert k == len(traj) and np.random.normal(0, 1) == next_draw_ref
    assert vt.updates == ref.updates
    # Adam step counters: who was updated how often (directional: leaders never; update_count 2: odd episodes stand still)
    steps = vt.agents.step.view(P, L).cpu().numpy()
    for p in range(P):
        for m in range(L):
            assert steps[p, m] == ref.a_opts[p][m].t == ref.c_opts[p][m].t, (p, m, steps[p, m], ref.a_opts[p][m].t)
    n_upd = max(1, int(steps.max()))
    for p in range(P):
        for m in range(L):
            for which, refw, lr in (("actor", ref.actors[p][m], conf.actor_lr), ("critic", ref.critics[p][m], conf.critic_lr)):
                for got, want in zip(vt.agents.get_weights(p * L + m, which), refw):
                    d = np.abs(got - want)
                    assert d.max() <= 2 * lr * n_upd + 1e-6 and d.mean() <= 0.05 * lr * n_upd + 1e-7, (variant, which, d.max(), d.mean())
    if conf.weighted_average_enabled:
        assert vt.fed_weights is not None and vt.fed_weights[0] == episodes - 1
        w = vt.fed_weights[1].cpu().numpy()  # [P, M]
        want = np.array(ref.fed_weights).T if conf.fed_method == "interfrl" else np.array(ref.fed_weights)
        assert np.allclose(w, want, rtol=1e-3)


@pytest.mark.parametrize("kw", [dict(), dict(model="ModelA", pl_size=3), dict(rand_gen="uniform", pl_size=7, num_platoons=70),
                                dict(fed_method="interfrl", weighted_average_enabled=False)])
def test_fused_step_launch_is_bitwise_the_seven_separate_kernels(kw):
    """avd_step_fused_f32 (OU noise, policy clip, leader exog, platoon step, replay add, reward counters in one launch, platoons
    inside one wavefront, shuffle / ballot reductions) against ou_step + policy + normal/uniform + env_step + replay_add
    + the reward update as separate launches: same Philox draws, same unfused arithmetic -> identical bits in every state
    tensor, through episode ends (20-step episodes; any-terminal resets) and the first updates. Also covers
    avd_replay_sample_f32 against indices + gather (the fused path samples with it, the other path is forced onto the two
    kernels)."""
    base = dict(num_platoons=130, pl_size=5, buffer_size=96, episode_sim_time=2.0)
    base.update(kw)
    runs = []
    for fused in (True, False):
        conf = config.Config(**base)
        vt = trainer.VecTrainer(conf, rng="device", auto_reset=True, seed=9, fused_step=fused,
                                fused_update=(conf.fed_method == "normal"))
        assert vt.fused_step == fused
        if not fused:  # the unfused reference path: index kernel + gather kernel
            rp = vt.replay
            rp.sample = lambda host_idx=None, rp=rp: (rp.draw_indices(host_idx), rp.gather())[1]
        vt.reset_episode()
        for _ in range(75):
            vt.step()
        torch.cuda.synchronize()
        runs.append(vt)
    a, b = runs
    assert a.replay.buffer_counter == b.replay.buffer_counter == 75 and a.ou.calls == b.ou.calls and a.exog_calls == b.exog_calls
    for name in ("x", "x_prev", "prev_a", "reward", "term", "done"):
        assert torch.equal(getattr(a.env, name), getattr(b.env, name)), name
    assert torch.equal(a.actions, b.actions) and torch.equal(a.leader_exog, b.leader_exog) and torch.equal(a.ou.state, b.ou.state)
    assert torch.equal(a.replay.ring, b.replay.ring) and torch.equal(a.ep_reward, b.ep_reward)
    assert torch.equal(a.replay.idx, b.replay.idx) and torch.equal(a.replay.s, b.replay.s) and torch.equal(a.replay.s2, b.replay.s2)
    assert torch.equal(a.replay.a, b.replay.a) and torch.equal(a.replay.r, b.replay.r)
    assert torch.equal(a.agents.theta, b.agents.theta) and torch.equal(a.agents.theta_t, b.agents.theta_t)
    assert a.episode == b.episode >= 3 and int(a.env.any_done.item()) == int(b.env.any_done.item())
    assert a.updates == (75 - 64) * a.n_agents


def test_per_platoon_auto_reset_runs_every_platoon_in_its_own_episodes():
    """auto_reset="platoon" (vec.VecPlatoon.episode_end): platoons close their own episodes (own terminal step or own step limit)
    and the episodic rewards arrive as per-platoon running sums; the fused nofrl update's "next action from the epilogue" must be
    recomputed for the steps after a reset (same trajectory as the unfused update, bit for bit)."""
    P, L, steps, limit = 96, 3, 150, 20
    runs = []
    for fused in (False, True):
        conf = config.Config(num_platoons=P, pl_size=L, buffer_size=256, episode_sim_time=limit * 0.1)
        vt = trainer.VecTrainer(conf, rng="device", auto_reset="platoon", fused_update=fused, seed=3)
        vt.reset_episode()
        seen = torch.zeros(P, dtype=torch.int32, device="cuda")
        for k in range(steps):
            vt.step()
            seen = torch.maximum(seen, vt.env.ep_len)
        torch.cuda.synchronize()
        runs.append(vt)
        assert int(seen.max()) <= limit - 1 and int(vt.env.ep_len.min()) >= 0
        cnt = vt.env.ep_stats["count"].cpu().numpy()
        lens = vt.env.ep_stats["len_sum"].cpu().numpy()
        assert cnt.min() >= steps // limit and np.all(lens + vt.env.ep_len.cpu().numpy() == steps)  # every step belongs to an episode
        ret = vt.env.ep_stats["ret_sum"].cpu().numpy()
        assert np.all(ret < 0) and np.all(np.isfinite(ret))
        # rewards are in (-0.5 * steps, 0): the closed episodes' platoon-mean sums + the running counters account for every step
        total = ret + vt.ep_reward.cpu().numpy().mean(axis=1)
        assert np.all(total > -0.5 * steps - 1e-3) and np.all(total < 0)
        mean_ret, mean_len, n = vt.env.pop_episode_stats()
        assert n == int(cnt.sum()) and 1 <= mean_len <= limit and mean_ret < 0
    a, b = runs
    assert torch.equal(a.env.x, b.env.x) and torch.equal(a.agents.theta, b.agents.theta) and torch.equal(a.actions, b.actions)
    with pytest.raises(ValueError, match="auto_reset"):
        trainer.VecTrainer(config.Config(num_platoons=2, pl_size=2), rng="device", auto_reset="vehicle")


def test_cli_throughput_mode_with_per_platoon_episodes(tmp_path, capsys):
    """`python -m avddpg_amd tr --rng device --episodes platoon`: the fast path from the command line -- device RNG, per-platoon
    episodes, no host synchronisation per step; curve.csv + conf.json + checkpoints of the first platoons, reloadable by `esim`."""
    import os

    from avddpg_amd import __main__ as cli

    cli.main(["tr", "--pl_num", "32", "--pl_size", "3", "--total_time_steps", "300", "--buffer_size", "500", "--fed_method", "interfrl",
              "--engine", "fused3", "--rng", "device", "--episodes", "platoon", "--report_every", "100", "--save_platoons", "2",
              "--out", str(tmp_path)])
    base = capsys.readouterr().out.strip().splitlines()[-1]
    rows = open(os.path.join(base, "curve.csv")).read().strip().splitlines()
    assert rows[0].startswith("step,episodes_closed") and len(rows) == 1 + 1 + 3
    last = rows[-1].split(",")
    assert int(last[0]) == 300 and int(last[1]) > 0 and float(last[2]) < 0 and 1 <= float(last[3]) <= 600 and float(last[4]) < 0
    assert os.path.exists(os.path.join(base, "conf.json")) and os.path.exists(os.path.join(base, "actor2_3.npz"))
    assert not os.path.exists(os.path.join(base, "actor3_1.npz"))
    cli.main(["esim", base, "--n_timesteps", "50"])
    out = capsys.readouterr().out
    assert "platoon 1:" in out and "platoon 2:" in out and "platoon 3:" not in out
    with pytest.raises(SystemExit):
        cli.main(["tr", "--episodes", "platoon", "--out", str(tmp_path)])  # host RNG: refused


@pytest.mark.parametrize("weighted,directional", [(False, False), (False, True), (True, True)])
def test_intrafrl_mean_inside_the_adam_pass_equals_the_four_kernel_path(weighted, directional):
    """intrafrl + gradients (workers/trainer.py:189-190, 417-431) with the platoon's mean gradient formed where Adam
    consumes it (avd_adam_polyak_intra_f32: the gradient slab is read once) against fed_sum + fed_finalize + fed_scatter + apply
    with the lead vehicle's slabs saved and restored (the r05 path; VecTrainer.intra_fused = False), and the two-stream pipeline
    over platoon chunks against both: 75 steps = 11 updates from identical Philox streams. Same summation order and scaling ->
    the SAME BITS in every weight, moment, target and Adam step count (lead vehicles: untouched under directional averaging)."""
    need_gpu()
    P, L = 37, 5
    conf = config.Config(num_platoons=P, pl_size=L, buffer_size=256, fed_method="intrafrl", weighted_average_enabled=weighted,
                         weighted_window=2, intra_directional_averaging=directional, episode_sim_time=1.5)
    runs = []
    for fused, chunks in ((False, 1), (True, 1), (True, 6)):
        vt = trainer.VecTrainer(conf, rng="device", auto_reset=True, seed=21, pipeline_chunks=chunks)
        vt.intra_fused = fused
        vt.reset_episode()
        for _ in range(75):
            vt.step()
        torch.cuda.synchronize()
        runs.append(vt)
    ref = runs[0]
    assert int(ref.agents.step.view(P, L)[0, 1]) == 11 and not torch.equal(ref.agents.theta, ref.agents.theta_t)
    if directional:
        assert int(ref.agents.step.view(P, L)[:, 0].max()) == 0  # lead vehicles never stepped
    if weighted:
        assert (ref._w_raw != 1).any()  # the weights did switch on (episode >= window)
    for vt in runs[1:]:
        for name in ("theta", "theta_t", "stats_t", "m", "v", "step"):
            assert torch.equal(getattr(vt.agents, name), getattr(ref.agents, name)), name
        assert torch.equal(vt.env.x, ref.env.x)
