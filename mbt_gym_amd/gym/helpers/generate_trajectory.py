"""The canonical rollout loop and its output layout (reference: mbt_gym/gym/helpers/generate_trajectory.py:8-38):
observations (N, D, n_steps + 1), actions (N, A, n_steps), rewards (N, 1, n_steps).

Two execution paths with identical results:
  * the reference's loop - agent.get_action(obs) on the host, one env.step() launch per time step;
  * `fused=True` (default whenever the agent can describe itself to the device, `agent.device_policy()`): the whole
    episode in ONE kernel launch (csrc/step_kernel.hpp: rollout_kernel).  The device records time-major (every store
    a coalesced row); the arrays returned are transposed VIEWS of that recording with the reference's shapes - no
    host-side transpose pass over what is 5.9 GB at 2^20 trajectories x 200 steps.  Reductions over the time axis
    (episode returns, plotting.py:96-108) run at full speed on them; `np.ascontiguousarray` gives the reference's
    memory order when a consumer needs it.
"""
import numpy as np


def generate_trajectory_on_device(env, agent, seed: int = None):
    """The fused path with the recording left in HBM: returns three torch CUDA tensors with the reference's shapes -
    observations (N, D, n_steps + 1), actions (N, A, n_steps), rewards (N, 1, n_steps) - as transposed views of the
    time-major buffers the kernel wrote (no copy, nothing crosses PCIe).  At 2^20 trajectories x 200 steps the host
    version moves those 5.9 GB over PCIe (0.1 s into pooled pinned arrays); this one is the 20 ms the kernel needs.  PyTorch only
    owns the memory here (and is what an on-GPU consumer would hand the tensors to)."""
    import torch

    if not hasattr(agent, "device_policy"):
        raise ValueError("the fused rollout needs an agent that can describe itself to the device (device_policy())")
    if seed is not None:
        env.seed(seed)
    n, n_pad, horizon = env.num_trajectories, env.padded_lanes, env.n_steps
    device = torch.device("cuda", env.device)
    obs = torch.empty((horizon + 1, n_pad, env.observation_dim), dtype=torch.float32, device=device)
    act = torch.empty((horizon, n_pad, env.action_dim), dtype=torch.float32, device=device)
    rew = torch.empty((horizon, n_pad), dtype=torch.float32, device=device)
    env.set_stream(torch.cuda.current_stream(device).cuda_stream)  # torch's allocator and the kernel share a stream
    env.reset_device()
    steps, _ = env.rollout_device(agent, max_steps=horizon, obs_ptr=obs.data_ptr(), act_ptr=act.data_ptr(), rew_ptr=rew.data_ptr())
    if steps < horizon:  # an episode that starts late leaves the tail zero, as GT:11-13 allocates it
        obs[steps + 1:].zero_()
        act[steps:].zero_()
        rew[steps:].zero_()
    return obs[:, :n].permute(1, 2, 0), act[:, :n].permute(1, 2, 0), rew[:, :n].t().unsqueeze(1)


def generate_trajectory(env, agent, seed: int = None, include_log_probs: bool = False, fused: bool = None):
    if seed is not None:
        env.seed(seed)
    n, horizon = env.num_trajectories, env.n_steps
    if include_log_probs:
        fused = False  # the agent's own sampling carries the autograd graph (GT:16-17, GT:22-23): the reference's loop
    if fused is None:
        fused = hasattr(agent, "device_policy") and getattr(agent, "has_device_policy", True) and getattr(env, "noise", "philox") == "philox"
    if fused:
        env.reset()
        obs_t, act_t, rew_t, steps, _ = env.rollout(agent, max_steps=horizon, record=True)
        if steps < horizon:  # an episode that starts late (start_time > 0) leaves the tail zero, as GT:11-13 allocates it
            obs_t = np.concatenate([obs_t, np.zeros((horizon - steps,) + obs_t.shape[1:], np.float32)])
            act_t = np.concatenate([act_t, np.zeros((horizon - steps,) + act_t.shape[1:], np.float32)])
            rew_t = np.concatenate([rew_t, np.zeros((horizon - steps, n), np.float32)])
        return np.transpose(obs_t, (1, 2, 0)), np.transpose(act_t, (1, 2, 0)), rew_t.T[:, None, :]
    observations = np.zeros((n, env.observation_space.shape[0], horizon + 1), dtype=np.float32)
    actions = np.zeros((n, env.action_space.shape[0], horizon), dtype=np.float32)
    rewards = np.zeros((n, 1, horizon), dtype=np.float32)
    log_probs = None
    if include_log_probs:
        import torch

        log_probs = torch.zeros((n, env.action_space.shape[0], horizon))
    obs = env.reset()
    observations[:, :, 0] = obs
    for k in range(horizon):
        if include_log_probs:
            action, log_prob = agent.get_action(obs, include_log_probs=True)
            log_probs[:, :, k] = log_prob.to(log_probs.device)
        else:
            action = agent.get_action(obs)
        obs, reward, done, _ = env.step(action)
        actions[:, :, k] = action
        observations[:, :, k + 1] = obs
        rewards[:, 0, k] = np.asarray(reward).reshape(-1)
        if (n > 1 and done[0]) or (n == 1 and done):  # GT:32
            break
    if include_log_probs:
        return observations, actions, rewards, log_probs
    return observations, actions, rewards
