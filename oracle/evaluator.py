"""Oracle: evaluator rollout (TEST INFRASTRUCTURE).

Restates the compute part of reference ``workers/evaluator.py:40-95, 145``: seed with ``evaluation_seed``,
evaluator-mode platoon (every vehicle starts at (1, 1, 0.03, a_lead)), pre-drawn leader input list, reset,
noise-free policy, float32 episodic reward counters, ``pl_rew = round(mean(counters), 3)``. Plotting is out of scope.
"""
import numpy as np

from . import mlp, platoon


def run(ep: platoon.EnvParams, pl_size, actors, steps, evaluation_seed=6, high=2.5, low=-2.5):
    """actors: list (one per vehicle) of Keras-ordered weight lists, or None for a zero policy."""
    np.random.seed(evaluation_seed)  # src/rand.py:10
    env = platoon.RefPlatoon(pl_size, ep, evaluator_states=True)  # evaluator.py:47
    inputs = [platoon.get_random_val(ep.rand_gen, ep.reset_max_u, std_dev=ep.reset_max_u) for _ in range(steps)]  # :55-56
    counters = np.array([0] * env.num_models, dtype=np.float32)
    states = env.reset()
    actions = np.zeros((env.num_models, env.num_actions))  # :58
    S, U, J = [], [], []
    for i in range(steps):
        for m in range(env.num_models):
            out = 0.0 if actors is None else mlp.actor_forward(actors[m], np.asarray(states[m])[None, :], high)
            actions[m] = mlp.policy(out, None, low, high)[0]  # :80, no noise
        states, rewards, _ = env.step(actions.flatten(), inputs[i])
        for m in range(env.num_models):
            counters[m] += rewards[m]
        S.append(np.array([np.asarray(s) for s in states]))
        U.append(actions.copy().ravel())
        J.append(np.array(env.get_jerk()).ravel())
    return round(np.average(counters), 3), dict(  # np.float32, rounded in float32 like the reference (:145)
        states=np.array(S), inputs=np.array(U), jerks=np.array(J),
                                                       counters=counters, leader=np.array(inputs))
