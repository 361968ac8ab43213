"""Hyper-parameter table of the reference's agents (utilities/data_structures/Config.py:18-109):
same keys and values, so `self.hyper_parameters[self.agent][...]` reads in agent code carry over."""
import random


def _table():
    return {
        "DA3C": dict(learning_rate=3e-4, discount_rate=0.99, num_episodes_to_run=1200, gradient_clipping_norm=1.0,
                     clip_rewards=True, normalise_rewards=True, epsilon_decay_rate_denominator=1.0,
                     exploration_worker_difference=2.0),
        "MP_PPO": dict(actor_number=5, policy_update_round=10, num_episodes_to_run=1000, tau=0.005, learning_rate=3e-4,
                       discount_rate=0.99, buffer_size=10000, batch_size=256, episodes_per_learning_round=10,
                       learning_iterations_per_round=10, learning_iterations_per_round_actor=10,
                       learning_iterations_per_round_critic=10, clip_epsilon=0.2, mu=0, theta=0.15, sigma=0.2,
                       epsilon_decay_rate_denominator=10, clip_rewards=False, normalized_rewards=True,
                       standardized_rewards=True, gradient_clipping_norm=1.0),
        "HMP_SAC": dict(num_episodes_to_run=2000, learning_rate=3e-4, discount_rate=0.99, buffer_size=10000,
                        batch_size=256, gradient_clipping_norm=1.0, min_steps_before_learning=10000, tau=0.005,
                        learning_updates_per_learning_session=10, update_every_n_steps=1000, add_extra_noise=False,
                        do_evaluation_iterations=False, entropy_term_weight=0, normalized_rewards=True,
                        standardized_rewards=True, automatically_tune_entropy_hyper_parameter=True),
        "DDQN": dict(num_episodes_to_run=1000, gradient_clipping_norm=5.0, tau=0.005, buffer_size=100000,
                     batch_size=1280, learning_iterations=1, update_every_n_steps=10, epsilon_decay_rate_denominator=10,
                     learning_rate=1e-6, discount_rate=1),
    }


class Config(object):
    def __init__(self):
        self.seed = random.randint(1, 100)
        self.environment_test = None
        self.requirements_to_solve_game = None
        self.num_episodes_to_run = None
        self.file_to_save_data_results = None
        self.file_to_save_results_graph = None
        self.use_GPU = None
        self.overwrite_existing_results_file = None
        self.save_model = True
        self.hyper_parameters = self.hyper_parameter()

    def hyper_parameter(self):
        return _table()
