"""The canonical rollout loop and its output layout (reference: mbt_gym/gym/helpers/generate_trajectory.py:8-38):
observations (N, D, n_steps + 1), actions (N, A, n_steps), rewards (N, 1, n_steps)."""
import numpy as np


def generate_trajectory(env, agent, seed: int = None, include_log_probs: bool = False):
    if include_log_probs:
        raise NotImplementedError("log-probabilities belong to the learning agent, not to the environment path")
    if seed is not None:
        env.seed(seed)
    n, horizon = env.num_trajectories, env.n_steps
    observations = np.zeros((n, env.observation_space.shape[0], horizon + 1), dtype=np.float32)
    actions = np.zeros((n, env.action_space.shape[0], horizon), dtype=np.float32)
    rewards = np.zeros((n, 1, horizon), dtype=np.float32)
    obs = env.reset()
    observations[:, :, 0] = obs
    for k in range(horizon):
        action = agent.get_action(obs)
        obs, reward, done, _ = env.step(action)
        actions[:, :, k] = action
        observations[:, :, k + 1] = obs
        rewards[:, 0, k] = np.asarray(reward).reshape(-1)
        if (n > 1 and done[0]) or (n == 1 and done):  # GT:32
            break
    return observations, actions, rewards
