"""Oracle: the reference training loop, per-object Python loops (TEST INFRASTRUCTURE).

Restates ``workers/trainer.py``: initialize :71-179, run inner loop :246-271,
advance_environment :282-302, train_all_models :304-359 (local updates and parameter aggregation),
the federated branches -- gradients :400-431, weights :433-456, weighted FRL :361-398, intrafrl and its
directional form :417-418 -- and the schedule predicates :631-695 with their quirk (the local-update gate tests the step
only, :345, so a valid step of a non-update episode updates nothing).  The centralized framework follows the
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
                 hidd_mult=1.2, aggregation_method="gradients", weighted_average_enabled=False, weighted_window=10,
                 fed_update_count=1, fed_cutoff_episode=10 ** 9, fed_update_delay_steps=1, intra_directional_averaging=False,
                 steps_per_episode=600):
        np.random.seed(seed)  # src/rand.py:10
        self.ep, self.P = ep, num_platoons
        self.batch_size, self.gamma, self.tau, self.fed_method = batch_size, gamma, tau, fed_method
        # src/config.py:25-37 (FRL knobs) and :89 (steps_per_episode)
        self.aggregation_method, self.weighted_average_enabled = aggregation_method, weighted_average_enabled
        self.weighted_window, self.fed_update_count, self.fed_cutoff_episode = weighted_window, fed_update_count, fed_cutoff_episode
        self.fed_update_delay_steps, self.intra_directional_averaging = fed_update_delay_steps, intra_directional_averaging
        self.steps_per_episode = steps_per_episode
        centralized = ep.framework == "centralized"
        self.M = 1 if centralized else pl_size  # environment.py:35-42
        S, A = ep.num_obs * (pl_size if centralized else 1), (pl_size if centralized else 1)
        if centralized and fed_method != "normal":
            raise ValueError("FRL is decentralized-only (trainer.py:632)")
        self.fed_enabled = fed_method in ("interfrl", "intrafrl") and not centralized  # :631-632
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
        self.all_ep_reward_lists = [[[] for _ in range(self.M)] for _ in range(self.P)]  # :173-177
        self.rbuffers_filled = [[False] * self.M for _ in range(self.P)]
        # aggregation lists [idx1][idx2] (:141-171): interfrl [vehicle][platoon], intrafrl [platoon][vehicle]
        n1, n2 = (self.M, self.P) if fed_method == "interfrl" else (self.P, self.M)
        mk = lambda: [[None] * n2 for _ in range(n1)]
        self.a_grad_list, self.c_grad_list, self.a_weight_list, self.c_weight_list = mk(), mk(), mk(), mk()
        self.fed_weights = [[1.0] * n2 for _ in range(n1)]
        self.fed_weight_sums = None

    # ---- schedule predicates (workers/trainer.py:631-695) ------------------------------------------------------
    def _weighted(self, episode):
        return self.weighted_average_enabled and episode >= self.weighted_window

    def _valid_update_episode(self, episode):
        return self.fed_enabled and episode % self.fed_update_count == 0 and episode <= self.fed_cutoff_episode

    def _valid_update_step(self, step):
        return step % self.fed_update_delay_steps == 0

    def _fed_step(self, episode, step, method):
        return (self.fed_enabled and self._valid_update_episode(episode) and self._valid_update_step(step)
                and self.aggregation_method == method)

    def reset_episode(self):  # trainer.py:244-249
        self.prev_states = [self.envs[p].reset() for p in range(self.P)]
        self.ep_reward = [np.array([0] * self.M, dtype=np.float32) for _ in range(self.P)]

    def _apply_local(self, p, m, cg, ag):  # :348-356
        c, a = self.critics[p][m], self.actors[p][m]
        self.c_opts[p][m].apply_gradients(cg, [c[i] for i in mlp.CRITIC_TRAINABLE])
        self.a_opts[p][m].apply_gradients(ag, [a[i] for i in mlp.ACTOR_TRAINABLE])
        self._update_targets(p, m)

    def _update_targets(self, p, m):
        tc, ta = mlp.update_target(self.tau, self.t_critics[p][m], self.critics[p][m], self.t_actors[p][m], self.actors[p][m])
        self.t_critics[p][m], self.t_actors[p][m] = tc, ta

    def _aggregate(self, p, m, factor, cg, ag):
        """aggregate_params (:361-383): weighted -> the parameters are stored PRE-multiplied by the factor."""
        i1, i2 = (m, p) if self.fed_method == "interfrl" else (p, m)
        aw, cw = self.actors[p][m], self.critics[p][m]  # `.weights`: trainables AND BN statistics, live references (:330-331)
        if factor is not None:
            f = np.float32(factor)
            mul = lambda ws: [(w * f).astype(np.float32) for w in ws]
            self.a_grad_list[i1][i2], self.c_grad_list[i1][i2] = mul(ag), mul(cg)
            self.a_weight_list[i1][i2], self.c_weight_list[i1][i2] = mul(aw), mul(cw)
            self.fed_weights[i1][i2] = float(factor)
        else:
            self.a_grad_list[i1][i2], self.c_grad_list[i1][i2] = ag, cg
            self.a_weight_list[i1][i2], self.c_weight_list[i1][i2] = aw, cw

    def _all_filled(self):  # :458-470
        return all(all(row) for row in self.rbuffers_filled)

    def _federated_gradients(self, episode):
        """train_all_models_federated_gradients (:400-431)."""
        if not self._all_filled():
            return
        if self._weighted(episode):
            a_avg = federated.get_weighted_avg_params(self.a_grad_list, self.fed_weight_sums)
            c_avg = federated.get_weighted_avg_params(self.c_grad_list, self.fed_weight_sums)
        else:
            a_avg, c_avg = federated.get_avg_params(self.a_grad_list), federated.get_avg_params(self.c_grad_list)
        for p in range(self.P):
            for m in range(self.M):
                if self.fed_method == "intrafrl" and m == 0 and self.intra_directional_averaging:
                    continue  # :417-418: the lead vehicle takes no step at all, not even the soft update
                g = m if self.fed_method == "interfrl" else p
                a, c = self.actors[p][m], self.critics[p][m]
                self.a_opts[p][m].apply_gradients(a_avg[g], [a[i] for i in mlp.ACTOR_TRAINABLE])  # actor first here (:420-425)
                self.c_opts[p][m].apply_gradients(c_avg[g], [c[i] for i in mlp.CRITIC_TRAINABLE])
                self._update_targets(p, m)

    def _federated_weights(self, episode):
        """train_all_models_federated_weights (:433-456): group [0]'s average goes into EVERY model and target."""
        if not self._all_filled():
            return
        if self._weighted(episode):
            a_avg = federated.get_weighted_avg_params(self.a_weight_list, self.fed_weight_sums)[0]
            c_avg = federated.get_weighted_avg_params(self.c_weight_list, self.fed_weight_sums)[0]
        else:
            a_avg, c_avg = federated.get_avg_params(self.a_weight_list)[0], federated.get_avg_params(self.c_weight_list)[0]
        for p in range(self.P):
            for m in range(self.M):
                if self.fed_method == "intrafrl" and m == 0 and self.intra_directional_averaging:
                    continue
                self.actors[p][m] = [w.copy() for w in a_avg]
                self.critics[p][m] = [w.copy() for w in c_avg]
                self.t_actors[p][m] = [w.copy() for w in a_avg]
                self.t_critics[p][m] = [w.copy() for w in c_avg]

    def step(self, episode=0, i=0):
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
        for p in range(self.P):  # train_all_models :314-359
            for m in range(self.M):
                rb = self.rbufs[p][m]
                rb.add((self.prev_states[p][m], self.actions[p][m], all_rewards[p][m], all_states[p][m]))
                self.ep_reward[p][m] += all_rewards[p][m]
                if rb.buffer_counter > self.batch_size:  # :322 strict
                    self.rbuffers_filled[p][m] = True
                    cg, ag, _ = mlp.learn(rb.sample(), self.actors[p][m], self.critics[p][m], self.t_actors[p][m],
                                          self.t_critics[p][m], self.gamma, self.high)
                    self.updates += 1
                    factor = (federated.frl_weight(self.all_ep_reward_lists[p][m], self.weighted_window)
                              if self._weighted(episode) else None)  # :334-339, :385-395
                    if self.fed_method in ("interfrl", "intrafrl"):
                        self._aggregate(p, m, factor, cg, ag)
                    # local updates only when no global update can occur: the gate tests the STEP, not the episode (:345)
                    if not self.fed_enabled or not self._valid_update_step(i):
                        self._apply_local(p, m, cg, ag)
        if self._weighted(episode):
            self.fed_weight_sums = [float(np.sum(row)) for row in self.fed_weights]  # :358-359
        if self._fed_step(episode, i, "gradients"):
            self._federated_gradients(episode)
        if self._fed_step(episode, i, "weights"):
            self._federated_weights(episode)
        self.prev_states = all_states
        return True in terms

    def update_reward_list(self):
        """trainer.py:510-517 (the episodic rewards the FRL weights are computed from)."""
        for p in range(self.P):
            for m in range(self.M):
                self.all_ep_reward_lists[p][m].append(self.ep_reward[p][m])

    def run(self, number_of_episodes):
        """trainer.py:232-273."""
        for episode in range(number_of_episodes):
            self.reset_episode()
            for i in range(self.steps_per_episode):
                if self.step(episode, i):
                    break
            self.update_reward_list()
