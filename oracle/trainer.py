"""Oracle: the reference training loop, per-object Python loops (TEST INFRASTRUCTURE).

Restates ``workers/trainer.py``: initialize :71-179, run inner loop :246-271,
advance_environment :282-302, train_all_models :304-356 (nofrl local updates) and
the interfrl+gradients branch :400-431.  The centralized framework follows the
loop shape of workers/evaluator.py:48-91 (``env.num_models`` = 1 model per platoon
with S = 4L, A = L, widths x centrl_hidd_mult; one scalar OU process broadcast
over the L actions, agent/ddpgagent.py:22): the trainer itself iterates
``conf.pl_size`` models (trainer.py:45) and so only runs centralized for pl_size = 1,
where both readings coincide.  Structured like the reference -- one
Platoon / OU / replay / actor / critic / Adam object per (platoon, vehicle), global
legacy numpy RNG, float64 env, float32 networks -- so that timing it on one core
is the CPU baseline ``bench.py`` reports beside the GPU number (kind "port":
TensorFlow is not installable here, so the NN arithmetic is NumPy, not TF-eager).
"""
import numpy as np

from . import federated, mlp, noise, platoon, replay


class RefTrainer:
    def __init__(self, ep: platoon.EnvParams, num_platoons=1, pl_size=3, seed=1, buffer_size=100000, batch_size=64,
                 gamma=0.99, tau=0.001, critic_lr=5e-4, actor_lr=5e-5, fed_method="normal", H1=256, H2=128, Ha=48,
                 hidd_mult=1.2):
        np.random.seed(seed)  # src/rand.py:10
        self.ep, self.P = ep, num_platoons
        self.batch_size, self.gamma, self.tau, self.fed_method = batch_size, gamma, tau, fed_method
        centralized = ep.framework == "centralized"
        self.M = 1 if centralized else pl_size  # environment.py:35-42
        S, A = ep.num_obs * (pl_size if centralized else 1), (pl_size if centralized else 1)
        if centralized and fed_method != "normal":
            raise ValueError("FRL is decentralized-only (trainer.py:632)")
        hm = hidd_mult if centralized else 1  # environment.py:37, 41
        wrs = np.random.RandomState(seed + 7919)  # TF initialiser stream is not reproducible; own stream
        self.envs, self.ous, self.actors, self.critics, self.t_actors, self.t_critics = [], [], [], [], [], []
        self.a_opts, self.c_opts, self.rbufs = [], [], []
        init_a = mlp.init_actor(wrs, S, A, H1, H2, hidd_mult=hm)
        init_c = mlp.init_critic(wrs, S, A, H1, H2, Ha, hidd_mult=hm)
        for p in range(self.P):  # trainer.py:71-171
            self.envs.append(platoon.RefPlatoon(pl_size, ep))
            self.ous.append([noise.RefOUNoise(np.zeros(1)) for _ in range(self.M)])
            self.actors.append([[w.copy() for w in init_a] for _ in range(self.M)])  # :121-128 same init everywhere
            self.critics.append([[w.copy() for w in init_c] for _ in range(self.M)])
            self.t_actors.append([[w.copy() for w in init_a] for _ in range(self.M)])
            self.t_critics.append([[w.copy() for w in init_c] for _ in range(self.M)])
            self.c_opts.append([mlp.RefAdam(critic_lr) for _ in range(self.M)])
            self.a_opts.append([mlp.RefAdam(actor_lr) for _ in range(self.M)])
            self.rbufs.append([replay.RefReplayBuffer(buffer_size, batch_size, S, A) for _ in range(self.M)])
        self.actions = np.zeros((self.P, self.M, A))
        self.high, self.low = ep.action_high, ep.action_low
        self.prev_states = None
        self.ep_reward = None
        self.updates = 0

    def reset_episode(self):  # trainer.py:244-249
        self.prev_states = [self.envs[p].reset() for p in range(self.P)]
        self.ep_reward = [np.array([0] * self.M, dtype=np.float32) for _ in range(self.P)]

    def _apply_local(self, p, m, cg, ag):  # :348-356
        c, a = self.critics[p][m], self.actors[p][m]
        self.c_opts[p][m].apply_gradients(cg, [c[i] for i in mlp.CRITIC_TRAINABLE])
        self.a_opts[p][m].apply_gradients(ag, [a[i] for i in mlp.ACTOR_TRAINABLE])
        tc, ta = mlp.update_target(self.tau, self.t_critics[p][m], c, self.t_actors[p][m], a)
        self.t_critics[p][m], self.t_actors[p][m] = tc, ta

    def step(self):
        """One iteration of the loop at trainer.py:251-271. Returns any-terminal."""
        ep = self.ep
        all_states, all_rewards, terms = [], [], []
        for p in range(self.P):  # advance_environment :282-302
            for m in range(self.M):
                out = mlp.actor_forward(self.actors[p][m], np.asarray(self.prev_states[p][m])[None, :], self.high)
                self.actions[p][m] = mlp.policy(out, self.ous[p][m](), self.low, self.high)[0]
            s, r, t = self.envs[p].step(self.actions[p].flatten(),
                                        platoon.get_random_val(ep.rand_gen, ep.reset_max_u, std_dev=ep.reset_max_u))
            all_states.append(s), all_rewards.append(r), terms.append(t)
        grads = {}
        for p in range(self.P):  # train_all_models :314-356
            for m in range(self.M):
                rb = self.rbufs[p][m]
                rb.add((self.prev_states[p][m], self.actions[p][m], all_rewards[p][m], all_states[p][m]))
                self.ep_reward[p][m] += all_rewards[p][m]
                if rb.buffer_counter > self.batch_size:  # :322 strict
                    cg, ag, _ = mlp.learn(rb.sample(), self.actors[p][m], self.critics[p][m], self.t_actors[p][m],
                                          self.t_critics[p][m], self.gamma, self.high)
                    self.updates += 1
                    if self.fed_method == "interfrl":
                        grads[(p, m)] = (cg, ag)
                    else:
                        self._apply_local(p, m, cg, ag)
        if self.fed_method == "interfrl" and len(grads) == self.P * self.M:  # :400-431, unweighted
            a_avg = federated.get_avg_params([[grads[(p, m)][1] for p in range(self.P)] for m in range(self.M)])
            c_avg = federated.get_avg_params([[grads[(p, m)][0] for p in range(self.P)] for m in range(self.M)])
            for p in range(self.P):
                for m in range(self.M):
                    self._apply_local(p, m, c_avg[m], a_avg[m])
        self.prev_states = all_states
        return True in terms
