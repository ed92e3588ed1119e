"""Hyper-parameters of the hot path, same attribute names and defaults as the reference
``src/config.py:14-117`` (reporting / file-name / LaTeX fields are out of scope and omitted)."""


class Config:
    modelA = "ModelA"
    modelB = "ModelB"
    weights = "weights"
    gradients = "gradients"

    def __init__(self, **overrides):
        self.model = self.modelB  # config.py:7
        self.dcntrl, self.cntrl = "decentralized", "centralized"
        # federated learning (config.py:21-37)
        self.interfrl, self.intrafrl, self.nofrl = "interfrl", "intrafrl", "normal"
        self.fed_method = self.nofrl
        self.framework = self.dcntrl
        self.weighted_average_enabled = True
        self.weighted_window = 10
        self.fed_update_count = 1
        self.fed_cutoff_ratio = 1.0
        self.fed_update_delay = 0.1
        self.aggregation_method = self.gradients
        self.intra_directional_averaging = False
        # environment (config.py:39-72)
        self.num_platoons = 1
        self.pl_size = 2
        self.pl_leader_reset_a = 0
        self.reset_max_u = 0.100
        self.pl_leader_tau = 0.1
        self.exact, self.euler = "exact", "euler"
        self.method = self.euler
        self.timegap = 1.0
        self.dyn_coeff = 0.1
        self.reward_ep_coeff, self.reward_ev_coeff, self.reward_u_coeff, self.reward_jerk_coeff = 0.4, 0.2, 0.2, 0.2
        self.max_ep = self.max_ev = 20
        self.reset_ep_max, self.reset_max_ev, self.reset_max_a = 1.5, 1.5, 0.05
        self.reset_ep_eval_max, self.reset_ev_eval_max, self.reset_a_eval_max = 1, 1, 0.03
        self.action_high, self.action_low = 2.5, -2.5
        self.re_scalar = 1
        self.terminal_reward = 0.5
        # trainer (config.py:75-107)
        self.can_terminate = True
        self.random_seed = 1
        self.evaluation_seed = 6
        self.normal, self.uniform = "normal", "uniform"
        self.rand_gen = self.normal
        self.rand_states = True
        self.total_time_steps = 1000000
        self.sample_rate = 0.1
        self.episode_sim_time = 60
        self.gamma = 0.99
        self.centrl_hidd_mult = 1.2
        self.reward_averaging_window = 40
        self.critic_lr, self.actor_lr = 0.0005, 0.00005
        self.std_dev, self.theta, self.ou_dt = 0.02, 0.15, 1e-2
        self.tau = 0.001
        self.batch_size = 64
        self.buffer_size = 100000
        # models (config.py:112-117)
        self.actor_layer1_size, self.actor_layer2_size = 256, 128
        self.critic_layer1_size, self.critic_act_layer_size, self.critic_layer2_size = 256, 48, 128
        for k, v in overrides.items():
            if not hasattr(self, k):
                raise AttributeError(f"unknown Config field {k!r}")
            setattr(self, k, v)
        self.refresh()

    def refresh(self):
        """Derived counts (config.py:27, 88-94)."""
        self.fed_enabled = self.fed_method in (self.interfrl, self.intrafrl) and self.framework == self.dcntrl
        self.steps_per_episode = int(self.episode_sim_time / self.sample_rate)
        self.fed_update_delay_steps = int(self.fed_update_delay / self.sample_rate)
        self.number_of_episodes = int(self.total_time_steps / self.steps_per_episode)
        self.fed_cutoff_episode = int(self.fed_cutoff_ratio * self.number_of_episodes)
        return self
