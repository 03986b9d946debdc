"""The canonical rollout loop and its output layout (reference: mbt_gym/gym/helpers/generate_trajectory.py:8-38):
observations (N, D, n_steps + 1), actions (N, A, n_steps), rewards (N, 1, n_steps).

Two execution paths with identical results:
  * the reference's loop - agent.get_action(obs) on the host, one env.step() launch per time step;
  * `fused=True` (default whenever the agent can describe itself to the device, `agent.device_policy()`): the whole
    episode in ONE kernel launch (csrc/step_kernel.hpp: rollout_kernel).  The device records time-major; the arrays
    returned are transposed views with the reference's shapes.
"""
import numpy as np


def generate_trajectory(env, agent, seed: int = None, include_log_probs: bool = False, fused: bool = None):
    if include_log_probs:
        raise NotImplementedError("log-probabilities belong to the learning agent, not to the environment path")
    if seed is not None:
        env.seed(seed)
    n, horizon = env.num_trajectories, env.n_steps
    if fused is None:
        fused = hasattr(agent, "device_policy") and getattr(env, "noise", "philox") == "philox"
    if fused:
        env.reset()
        obs_t, act_t, rew_t, steps, _ = env.rollout(agent, max_steps=horizon, record=True)
        observations = np.zeros((n, obs_t.shape[2], horizon + 1), dtype=np.float32)
        actions = np.zeros((n, act_t.shape[2], horizon), dtype=np.float32)
        rewards = np.zeros((n, 1, horizon), dtype=np.float32)
        observations[:, :, : steps + 1] = np.transpose(obs_t, (1, 2, 0))
        actions[:, :, :steps] = np.transpose(act_t, (1, 2, 0))
        rewards[:, 0, :steps] = rew_t.T
        return observations, actions, rewards
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
