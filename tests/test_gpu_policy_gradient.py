"""PolicyGradientAgent (the reference's REINFORCE learner, agents/PolicyGradientAgent.py:14-73) on the device path: the
reference's interface and objective, an epoch's data from ONE fused rollout when the actor is a network the kernels evaluate,
the reference's own loop (generate_trajectory with include_log_probs) otherwise."""
import numpy as np
import pytest

from mbt_gym_amd.agents.PolicyGradientAgent import PolicyGradientAgent
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
from mbt_gym_amd.gym.helpers.generate_trajectory import generate_trajectory
from mbt_gym_amd.rewards.RewardFunctions import RunningInventoryPenalty

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _env(n, horizon=40, seed=3):
    return TradingEnvironment(num_trajectories=n, n_steps=horizon, seed=seed, max_inventory=50, reward_function=RunningInventoryPenalty(0.1, 0.5))


def _actor(hidden=64, act=torch.nn.Tanh):
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(4, hidden), act(), torch.nn.Linear(hidden, hidden), act(), torch.nn.Linear(hidden, 2))


def test_training_on_fused_rollouts_improves_the_return():
    env = _env(1 << 13)
    net = _actor().to("cuda")
    agent = PolicyGradientAgent(net, action_std=0.05, optimizer=torch.optim.Adam(net.parameters(), lr=3e-3), env=env)
    assert agent.has_device_policy
    losses, rewards = agent.train(num_epochs=40, reporting_freq=1000)
    assert len(losses) == len(rewards) == 40 and np.all(np.isfinite(losses))
    assert np.mean(rewards[-5:]) > np.mean(rewards[:5]) + 0.02, (rewards[:5], rewards[-5:])  # mean reward per step goes up
    # the trained actor, evaluated by the kernels deterministically, beats the untrained one
    trained = env.reset() is not None and env.rollout(agent.device_policy(deterministic=True))
    fresh = PolicyGradientAgent(_actor().to("cuda"), env=env)
    env.reset()
    untrained = env.rollout(fresh.device_policy(deterministic=True))
    assert trained[2].sum(axis=0).mean() > untrained[2].sum(axis=0).mean()
    env.close()


def test_reference_loop_for_actors_the_kernels_do_not_evaluate_and_the_two_paths_agree_in_law():
    """A three-hidden-layer actor takes the reference's loop (host forward pass + env.step per time step, log-probabilities
    with their autograd graph from the agent's own sampling).  For an actor both paths can run, the sampled data have the
    same law: same mean reward per step and same mean log-probability within Monte-Carlo error."""
    env = _env(4096, horizon=20)
    deep = torch.nn.Sequential(torch.nn.Linear(4, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                               torch.nn.Linear(32, 2))
    agent = PolicyGradientAgent(deep, action_std=0.05, env=env)
    assert not agent.has_device_policy
    losses, rewards = agent.train(num_epochs=2, reporting_freq=1000)
    assert len(losses) == 2 and np.all(np.isfinite(losses))
    obs, act, rew, log_probs = generate_trajectory(env, agent, include_log_probs=True)
    assert obs.shape == (4096, 4, 21) and act.shape == (4096, 2, 20) and rew.shape == (4096, 1, 20) and tuple(log_probs.shape) == (4096, 2, 20)
    assert log_probs.requires_grad
    # same actor, both paths
    net = _actor()
    both = PolicyGradientAgent(net, action_std=0.05, env=env)
    assert both.has_device_policy
    r_dev, lp_dev = both._sample_on_device()
    _, _, r_host, lp_host = generate_trajectory(env, both, include_log_probs=True)
    assert tuple(r_dev.shape) == r_host.shape and tuple(lp_dev.shape) == tuple(lp_host.shape)
    assert float(r_dev.mean()) == pytest.approx(float(r_host.mean()), abs=0.02)
    assert float(lp_dev.detach().mean()) == pytest.approx(float(lp_host.detach().mean()), abs=0.02)  # E log N(a | mu, 0.05) = -log(0.05 sqrt(2 pi e)) either way
    expected = -np.log(0.05 * np.sqrt(2 * np.pi * np.e))
    assert float(lp_dev.detach().mean()) == pytest.approx(expected, abs=0.02)
    env.close()


def test_sampled_actions_are_scored_around_the_mean_the_kernel_used():
    """The fused rollout samples a ~ N(kernel mean, std) with the network on fp16 operands; REINFORCE scores those actions
    with log N(a | mean, std).  `kernel_mean` restates the kernel's evaluation in torch (differentiable, straight-through for
    the weight rounding): it reproduces the kernel's deterministic actions several times closer than the float32 network does,
    so the score has no systematic (kernel mean - float32 mean) / std^2 term - at std = 0.01 that term would be O(1)."""
    env = _env(2048, horizon=10)
    net = _actor().to("cuda")
    agent = PolicyGradientAgent(net, action_std=0.01, env=env)
    obs = env.reset()
    for _ in range(4):
        obs, _, _, _ = env.step(np.zeros((2048, 2), np.float32))
    env.policy_device(agent.device_policy(deterministic=True))
    env.synchronize()
    kernel = torch.as_tensor(env.action_device, device="cuda")
    o = torch.as_tensor(obs, device="cuda")
    restated, plain = agent.kernel_mean(o), net(o)
    err_restated, err_plain = float((restated - kernel).abs().max()), float((plain - kernel).abs().max())
    assert err_restated <= 1e-4 and err_restated < 0.5 * err_plain, (err_restated, err_plain)  # (what is left: tanh implementations, fp16 rounding ties)
    restated.sum().backward()  # gradients reach the float32 parameters through the rounding
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0 for p in net.parameters())
    # and the score of sampled actions is centred: (a - kernel mean) / std has mean ~ 0 (it had a bias of ~ err_plain / std)
    r, lp = agent._sample_on_device()
    z_mean = float(((-2 * (lp.detach() + np.log(0.01 * np.sqrt(2 * np.pi)))).clamp(min=0).sqrt()).mean())  # log N = -z^2 / 2 - log(std sqrt(2 pi)); E|z| = 0.798
    assert z_mean == pytest.approx(np.sqrt(2 / np.pi), abs=0.02)
    env.close()


def test_rewards_to_go():
    r = torch.tensor([[[1.0, 2.0, 3.0]], [[0.5, 0.0, -1.0]]])
    np.testing.assert_allclose(PolicyGradientAgent._calculate_future_rewards(r).numpy(), [[[6.0, 5.0, 3.0]], [[-0.5, -1.0, -1.0]]])
