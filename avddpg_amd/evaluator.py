"""Evaluator rollout over the HIP kernels: the compute part of reference ``workers/evaluator.py:16-158``
(deterministic-start, noise-free rollout of trained actors; returns the mean episodic reward rounded to 3
digits, :145, :158). Figures / LaTeX are out of scope of the hot path; the traces they were drawn from are returned."""
import numpy as np
import torch

from . import vec
from ._hip import call, ptr, stream_handle


def get_number_of_timesteps_for_plot(conf, manual_timestep_override=None):
    return conf.steps_per_episode if manual_timestep_override is None else manual_timestep_override


def run(conf=None, actors=None, pl_idx=None, seed=True, manual_timestep_override=None, set_mod=None, **_ignored):
    """actors: an ``AgentGroup`` whose first ``pl_size`` weight sets are the platoon's vehicle actors (or whose
    sets are addressed with ``set_mod``). Host-RNG parity mode: the global legacy RNG is seeded with
    ``conf.evaluation_seed`` and consumed in the reference's order. Returns (pl_rew, traces)."""
    if seed:
        np.random.seed(conf.evaluation_seed)  # rand.set_global_seed (src/rand.py:10)
    L = conf.pl_size
    env = vec.VecPlatoon(1, L, conf, evaluator_states_enabled=True, rng="host", track_aux=True)  # evaluator.py:47
    steps = get_number_of_timesteps_for_plot(conf, manual_timestep_override)
    rand = (lambda: np.random.uniform(-conf.reset_max_u, conf.reset_max_u)) if conf.rand_gen == conf.uniform else \
        (lambda: np.random.normal(0, conf.reset_max_u))
    inputs = np.array([rand() for _ in range(steps)], dtype=np.float32)  # :55-56
    d_inputs = torch.from_numpy(inputs).to(env.device)
    M, A = env.num_models, env.num_actions  # centralized: one model with L actions and a 4L-wide observation (:48, :58)
    xs = 4 * L // M
    counters = torch.zeros(M, dtype=torch.float32, device=env.device)  # float32 counters (:67)
    env.reset()
    act = torch.zeros(1, L, dtype=torch.float32, device=env.device)
    raw = torch.zeros(M * A, dtype=torch.float32, device=env.device)
    sm = M if set_mod is None else set_mod
    states = torch.zeros(steps, L, env.obs_width, device=env.device)
    ctrl = torch.zeros(steps, L, device=env.device)
    jerks = torch.zeros(steps, L, device=env.device)
    for i in range(steps):
        actors.actor(env.x.view(M, xs), sm, x_stride=xs, out=raw)
        call("avd_policy_f32", L, ptr(raw), None, conf.action_low, conf.action_high, ptr(act), stream_handle())  # no noise
        pa_before = env.prev_a.clone()
        env.step(act, d_inputs[i:i + 1])
        counters += env.reward[0] if M == L else env.reward_mean
        states[i] = env.observations()[0]
        ctrl[i] = act[0]
        jerks[i] = env.get_jerk_from(env.x_prev, pa_before)[0]
    pl_rew = round(np.average(counters.cpu().numpy()), 3)  # np.float32 rounded in float32, as the reference (:145)
    return pl_rew, dict(states=states.cpu().numpy(), inputs=ctrl.cpu().numpy(), jerks=jerks.cpu().numpy(),
                        counters=counters.cpu().numpy(), leader=inputs)
