"""A trained Stable-Baselines3 model as an Agent (reference: mbt_gym/agents/SbAgent.py:8-26).

A consumer of the environment, not part of the accelerated path: `get_action` is `model.predict` on the (optionally
column-reduced) observation.  stable_baselines3 itself is not imported here - any object with `predict(obs,
deterministic=True) -> (actions, state)`, `action_space` and (for `train`) `learn(total_timesteps=...)` serves."""
import numpy as np

from mbt_gym_amd.agents.Agent import Agent


class SbAgent(Agent):
    def __init__(self, model, reduced_training_indices: list = None, num_trajectories: int = None):
        self.model = model
        self.num_trajectories = num_trajectories or model.env.num_trajectories
        self.num_actions = model.action_space.shape[0]
        self.reduced_training = reduced_training_indices is not None
        if self.reduced_training:
            self.reduced_training_indices = reduced_training_indices

    def get_action(self, state: np.ndarray) -> np.ndarray:
        observed = state[:, self.reduced_training_indices] if self.reduced_training else state
        actions, _ = self.model.predict(observed, deterministic=True)
        return np.asarray(actions).reshape(observed.shape[0], self.num_actions)

    def train(self, total_timesteps: int = 100000):
        self.model.learn(total_timesteps=total_timesteps)
