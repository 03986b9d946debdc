#!/usr/bin/env python3
"""The reference's policy-gradient training loop (agents/PolicyGradientAgent.py:49-73: sample a trajectory with
a ~ N(policy(obs), std), weight log-probabilities with the rewards-to-go, take an optimiser step) with the data collection
moved INTO the kernel: each epoch is ONE fused rollout launch in which the current network is evaluated on the matrix cores
and its exploration noise drawn from Philox (csrc/policy_mlp.hpp), recorded straight into torch tensors; PyTorch-ROCm only
does what needs gradients - the log-probabilities of the recorded (observation, action) pairs and the update.

    python examples/policy_gradient_on_device.py [log2_lanes] [epochs]

PyTorch is the consumer here, not the product.  (The reference spends 0.2 ms per env.step at N = 1000 and its agent one torch
forward per step on the host; here an epoch at 2^14 lanes x 50 steps is a ~1 ms launch plus the backward pass.)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mbt_gym_amd import _native  # noqa: E402
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment  # noqa: E402
from mbt_gym_amd.rewards.RewardFunctions import RunningInventoryPenalty  # noqa: E402


def device_policy(net, action_std):
    """The torch actor [Linear, Tanh, Linear, Tanh, Linear] -> the in-kernel policy (weights copied, 25 KB)."""
    layers = [(m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy()) for m in net if isinstance(m, torch.nn.Linear)]
    return _native.mlp_policy(layers, "tanh", action_std=action_std, clip=False)  # PolicyGradientAgent does not clip (PG:34-47)


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    n, horizon, action_std = 1 << log2n, 50, 0.05
    env = TradingEnvironment(num_trajectories=n, n_steps=horizon, seed=1, max_inventory=50,
                             reward_function=RunningInventoryPenalty(0.1, 0.5))  # the default market, normalised spaces
    dev = torch.device("cuda", 0)
    env.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 2)).to(dev)
    optimiser = torch.optim.Adam(net.parameters(), lr=3e-3)
    n_pad = env.padded_lanes
    obs = torch.empty((horizon + 1, n_pad, 4), device=dev)
    act = torch.empty((horizon, n_pad, 2), device=dev)
    rew = torch.empty((horizon, n_pad), device=dev)
    history = []
    t0 = time.perf_counter()
    for epoch in range(epochs):
        env.reset_device()
        steps, done = env.rollout_device(device_policy(net, action_std), obs_ptr=obs.data_ptr(), act_ptr=act.data_ptr(), rew_ptr=rew.data_ptr())
        assert steps == horizon and done
        o, a, r = obs[:-1, :n], act[:, :n], rew[:, :n]
        to_go = torch.flip(torch.cumsum(torch.flip(r, dims=(0,)), dim=0), dims=(0,))  # PG:70-73
        advantage = to_go - to_go.mean(dim=1, keepdim=True)
        log_prob = torch.distributions.Normal(net(o), action_std).log_prob(a).sum(dim=-1)  # of the actions the KERNEL sampled
        loss = -(log_prob * advantage).mean()
        optimiser.zero_grad()
        loss.backward()
        optimiser.step()
        history.append(float(r.sum(dim=0).mean()))
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    env.close()
    print(json.dumps({"lanes": n, "horizon": horizon, "epochs": epochs, "mean_episode_return_first_3": history[:3], "mean_episode_return_last_3": history[-3:],
                      "seconds_per_epoch": seconds / epochs, "env_steps_per_s_including_training": n * horizon * epochs / seconds}))
    return history


if __name__ == "__main__":
    main()
