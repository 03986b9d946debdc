"""The reference's REINFORCE learner (mbt_gym/agents/PolicyGradientAgent.py:14-73) as a consumer of the device path.

Same interface - `PolicyGradientAgent(policy, action_std, optimizer, env, lr_scheduler)`, `get_action(state, deterministic,
include_log_probs)`, `train(num_epochs, reporting_freq) -> (losses, mean rewards)` - and the same objective (log-probabilities
of the sampled actions weighted by the rewards-to-go, PG:49-73).  What differs is where an epoch's data comes from: when the
actor is a network the kernels can evaluate ([Linear, act, Linear, act, Linear], hidden width <= 64, tanh or relu) the whole
episode is sampled by ONE fused rollout launch - the network on the matrix cores, a ~ N(mean, std) drawn from Philox in the
kernel, observations / actions / rewards recorded straight into torch tensors - and PyTorch only computes what needs
gradients: log N(a | policy(obs), std) of the recorded pairs.  Any other torch module takes the reference's loop
(generate_trajectory with include_log_probs=True: one host forward pass and one env.step per time step).  PyTorch is the
consumer here, not the product."""
from typing import Callable, Tuple, Union

import numpy as np

from mbt_gym_amd.agents.Agent import Agent
from mbt_gym_amd.gym.helpers.generate_trajectory import generate_trajectory


def _actor_layers(policy):
    """([(W, b)] * 3, activation) when the kernels can evaluate the module, else None."""
    import torch

    modules = list(policy) if isinstance(policy, torch.nn.Sequential) else None
    if modules is None or len(modules) != 5:
        return None
    first, act1, second, act2, last = modules
    if not all(isinstance(m, torch.nn.Linear) for m in (first, second, last)) or type(act1) is not type(act2):
        return None
    activation = {torch.nn.Tanh: "tanh", torch.nn.ReLU: "relu"}.get(type(act1))
    hidden = first.out_features
    if activation is None or hidden > 64 or second.in_features != hidden or second.out_features != hidden or last.in_features != hidden:
        return None
    if any(m.bias is None for m in (first, second, last)):
        return None
    return [(m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy()) for m in (first, second, last)], activation


class PolicyGradientAgent(Agent):
    def __init__(self, policy, action_std: Union[float, Callable] = 0.01, optimizer=None, env=None, lr_scheduler=None):
        import torch
        from torch.optim.lr_scheduler import StepLR

        if env is None:
            from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment

            env = TradingEnvironment()
        self.env = env
        self.input_size = env.observation_space.shape[0]
        self.action_size = env.action_space.shape[0]
        first = policy[0] if hasattr(policy, "__getitem__") else None
        assert first is None or self.input_size == first.in_features
        self.policy_net = policy
        self.action_std = action_std
        self.optimizer = optimizer or torch.optim.SGD(self.policy_net.parameters(), lr=1e-1)
        self.lr_scheduler = lr_scheduler or StepLR(self.optimizer, step_size=1, gamma=0.995)
        self.noise_dist = torch.distributions.Normal
        self.proportion_completed: float = 0.0

    # ---- the policy -------------------------------------------------------------------------------------------------------
    def _std(self) -> float:
        return float(self.action_std(self.proportion_completed)) if isinstance(self.action_std, Callable) else float(self.action_std)

    def _module_device(self):
        import torch

        return next((p.device for p in self.policy_net.parameters()), torch.device("cpu"))

    def get_action(self, state: np.ndarray, deterministic: bool = False, include_log_probs: bool = False):
        import torch

        assert not (deterministic and include_log_probs), "include_log_probs is only an option for deterministic output"
        mean_value = self.policy_net(torch.as_tensor(np.asarray(state), dtype=torch.float, device=self._module_device()))
        if deterministic:
            return mean_value.detach().cpu().numpy()
        action_dist = self.noise_dist(loc=mean_value, scale=self._std() * torch.ones_like(mean_value))
        action = action_dist.sample()
        if include_log_probs:
            return action.detach().cpu().numpy(), action_dist.log_prob(action)
        return action.detach().cpu().numpy()

    @property
    def has_device_policy(self) -> bool:
        """The fused rollout can sample this actor on this environment: a [Linear, act, Linear, act, Linear] network, Philox
        noise, and observations the matrix cores can read as fp16 (normalised, or bounded by 4: SbAgent.observations_fit_fp16) -
        otherwise `train` takes the reference's host loop."""
        from mbt_gym_amd.agents.SbAgent import observations_fit_fp16

        return (_actor_layers(self.policy_net) is not None and hasattr(self.env, "rollout_device") and getattr(self.env, "noise", "philox") == "philox"
                and observations_fit_fp16(self.env))

    def kernel_mean(self, obs):
        """The actor's mean AS THE KERNEL EVALUATES IT (csrc/policy_mlp.hpp), in torch and differentiable: the observation row,
        [W1 | b1], the hidden activations, W2 and W3 rounded to fp16 - fp32 accumulation, b2 / b3 in fp32 - with a
        straight-through estimator for the rounding of the weights.  The kernel samples a ~ N(kernel mean, std); scoring those
        actions against the float32 `policy_net(obs)` instead would add (kernel mean - float32 mean) / std^2 x grad(mean) x
        reward-to-go to the REINFORCE gradient - a bias the reference's loop, which samples and scores with ONE mean, does not
        have, and which grows as the exploration std is annealed."""
        import torch

        first, act1, second, act2, last = list(self.policy_net)

        def half(t):  # value rounded to fp16, gradient passed straight through
            return t + (t.detach().half().float() - t.detach())

        x = obs.half().float()
        h = act1(torch.nn.functional.linear(x, half(first.weight), half(first.bias)))
        h = act2(torch.nn.functional.linear(half(h), half(second.weight), second.bias))
        return torch.nn.functional.linear(half(h), half(last.weight), last.bias)

    def device_policy(self, deterministic: bool = False):
        """The actor as the kernels evaluate it: sampling with the current std unless `deterministic`; never clipped (PG:34-47)."""
        from mbt_gym_amd import _native

        layers, activation = _actor_layers(self.policy_net)
        std = None if deterministic else [self._std()] * self.action_size
        return _native.mlp_policy(layers, activation, action_std=std, clip=False)

    # ---- one epoch of data ----------------------------------------------------------------------------------------------------
    def _sample_on_device(self):
        """(rewards (N, 1, T) tensor, log-probabilities (N, A, T) tensor with gradients), from one fused rollout launch."""
        import torch

        env = self.env
        n, n_pad, horizon = env.num_trajectories, env.padded_lanes, env.n_steps
        cuda = torch.device("cuda", env.device)
        obs = torch.empty((horizon + 1, n_pad, env.observation_dim), dtype=torch.float32, device=cuda)
        act = torch.empty((horizon, n_pad, env.action_dim), dtype=torch.float32, device=cuda)
        rew = torch.empty((horizon, n_pad), dtype=torch.float32, device=cuda)
        env.set_stream(torch.cuda.current_stream(cuda).cuda_stream)  # torch's allocator and the kernel share a stream
        env.reset_device()
        steps, _ = env.rollout_device(self.device_policy(), max_steps=horizon, obs_ptr=obs.data_ptr(), act_ptr=act.data_ptr(), rew_ptr=rew.data_ptr())
        where = self._module_device()
        o, a, r = obs[:steps, :n].to(where), act[:steps, :n].to(where), rew[:steps, :n].to(where)
        log_prob = self.noise_dist(self.kernel_mean(o), self._std()).log_prob(a)  # of the actions the KERNEL sampled, around ITS mean: (T, N, A)
        return r.t().unsqueeze(1), log_prob.permute(1, 2, 0)

    def train(self, num_epochs: int = 1, reporting_freq: int = 100):
        import torch

        learning_losses, learning_rewards = [], []
        self.proportion_completed = 0.0
        for epoch in range(num_epochs):
            if self.has_device_policy:
                rewards, log_probs = self._sample_on_device()
            else:
                _, _, rewards, log_probs = generate_trajectory(self.env, self, include_log_probs=True)
                rewards = torch.as_tensor(rewards, device=log_probs.device)
            learning_rewards.append(float(rewards.mean()))
            future_rewards = self._calculate_future_rewards(rewards)
            loss = -torch.mean(log_probs * future_rewards)  # PG:58-59
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
            if epoch % reporting_freq == 0:
                print(loss.item())
            learning_losses.append(loss.item())
            self.proportion_completed += 1 / max(1, num_epochs - 1)
            self.lr_scheduler.step()
        return learning_losses, learning_rewards

    @staticmethod
    def _calculate_future_rewards(rewards):
        import torch

        return torch.flip(torch.cumsum(torch.flip(rewards, dims=(-1,)), dim=-1), dims=(-1,))  # rewards-to-go (PG:69-73)
