#!/usr/bin/env python3
"""Plugin classes written against the reference - NumPy only - running unchanged on the HIP path (the host-callback route).

A user of mbt_gym subclasses its plugin base classes: a fill model states `_get_fill_probabilities(depths)` (FILL:22-34), a
reward function `calculate(current_state, action, next_state, is_terminal_step)` (RW:10-13), a price impact model
`get_impact(action)` (IMP:25-27) ... Nothing below knows about mbt_gym_amd beyond the import lines: swap `mbt_gym_amd` for
`mbt_gym` and the same file runs on the reference.  Here the user's methods keep running on the host, between kernel launches;
everything else of `env.step()` - draws, inventory mask, cash / inventory, clip, midprice, normalisation - is the fused kernel
(include/mbt_env.h, "host-callback plugins").  It is the slow path (a HostCallbackWarning says so); stating the same formula as a
device expression (tests/user_plugins.py) puts it into the kernel.

    python examples/bring_your_own_numpy_plugins.py [lanes]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mbt_gym_amd.gym.ModelDynamics import LimitOrderModelDynamics, TradinghWithSpeedModelDynamics  # noqa: E402
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment  # noqa: E402
from mbt_gym_amd.gym.index_names import ASSET_PRICE_INDEX, CASH_INDEX, INVENTORY_INDEX, TIME_INDEX  # noqa: E402
from mbt_gym_amd.rewards.RewardFunctions import RewardFunction  # noqa: E402
from mbt_gym_amd.stochastic_processes.arrival_models import PoissonArrivalModel  # noqa: E402
from mbt_gym_amd.stochastic_processes.fill_probability_models import FillProbabilityModel  # noqa: E402
from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel  # noqa: E402
from mbt_gym_amd.stochastic_processes.price_impact_models import PriceImpactModel  # noqa: E402


class PowerLawFill(FillProbabilityModel):
    """p(depth) = 1 / (1 + (scale depth)^power): a heavier tail than the exponential fill function."""

    def __init__(self, scale, power, step_size, num_trajectories, seed=None):
        self.scale, self.power = scale, power
        super().__init__(min_value=np.array([[]]), max_value=np.array([[]]), step_size=step_size, terminal_time=0.0,
                         initial_state=np.array([[]]), num_trajectories=num_trajectories, seed=seed)

    def _get_fill_probabilities(self, depths):
        return 1.0 / (1.0 + (self.scale * depths) ** self.power)

    @property
    def max_depth(self):
        return 99.0 ** (1.0 / self.power) / self.scale

    def update(self, arrivals, fills, actions, state=None):
        pass


class ExponentialInventoryCost(RewardFunction):
    """PnL - dt phi (exp(eta |q'|) - 1)."""

    def __init__(self, phi, eta):
        self.phi, self.eta = phi, eta

    def calculate(self, current_state, action, next_state, is_terminal_step=False):
        value = lambda s: s[:, CASH_INDEX] + s[:, INVENTORY_INDEX] * s[:, ASSET_PRICE_INDEX]  # noqa: E731
        dt = next_state[:, TIME_INDEX] - current_state[:, TIME_INDEX]
        return value(next_state) - value(current_state) - dt * self.phi * (np.exp(self.eta * np.abs(next_state[:, INVENTORY_INDEX])) - 1.0)

    def reset(self, initial_state):
        pass


class SquareRootImpact(PriceImpactModel):
    """impact = c sign(v) sqrt(|v|): the square-root law of market impact, stateless."""

    def __init__(self, coefficient, num_trajectories):
        self.coefficient = coefficient
        super().__init__(min_value=np.array([[]]), max_value=np.array([[]]), step_size=None, terminal_time=0.0,
                         initial_state=np.array([[]]), num_trajectories=num_trajectories, seed=None)

    def get_impact(self, action):
        return self.coefficient * np.sign(action) * np.sqrt(np.abs(action))

    def update(self, arrivals, fills, actions, state=None):
        pass

    @property
    def max_speed(self):
        return 10.0


def episode(env, action):
    env.reset()
    total, t0 = np.zeros(env.num_trajectories), time.perf_counter()
    for _ in range(env.n_steps):
        _, rewards, dones, _ = env.step(action)
        total += rewards
    assert dones[0]
    return total, (time.perf_counter() - t0) / env.n_steps


def main():
    n, ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 200
    market_making = TradingEnvironment(
        terminal_time=1.0, n_steps=ns, num_trajectories=n, seed=7, max_inventory=20, reward_function=ExponentialInventoryCost(0.05, 0.3),
        model_dynamics=LimitOrderModelDynamics(
            midprice_model=BrownianMotionMidpriceModel(volatility=2.0, step_size=1 / ns, num_trajectories=n),
            arrival_model=PoissonArrivalModel(intensity=np.array([140.0, 140.0]), step_size=1 / ns, num_trajectories=n),
            fill_probability_model=PowerLawFill(1.25, 1.5, step_size=1 / ns, num_trajectories=n), num_trajectories=n),
        normalise_action_space=False, normalise_observation_space=False)
    returns, per_step = episode(market_making, np.full((n, 2), 0.6, np.float32))
    print(f"market making, user fill model + user reward: mean episode return {returns.mean():.3f} over {n} lanes, {per_step * 1e6:.0f} us per step")
    execution = TradingEnvironment(
        terminal_time=1.0, n_steps=ns, num_trajectories=n, seed=7, initial_inventory=10, max_inventory=100,
        model_dynamics=TradinghWithSpeedModelDynamics(
            midprice_model=BrownianMotionMidpriceModel(volatility=1.0, step_size=1 / ns, num_trajectories=n),
            price_impact_model=SquareRootImpact(0.05, n), num_trajectories=n),
        normalise_action_space=False, normalise_observation_space=False)
    returns, per_step = episode(execution, np.full((n, 1), -10.0, np.float32))  # sell the ten units at a constant speed
    print(f"optimal execution, user price impact model: mean episode return {returns.mean():.3f}, {per_step * 1e6:.0f} us per step")
    print(f"  (the square-root law costs 10 x 0.05 x sqrt(10) = {10 * 0.05 * np.sqrt(10):.3f} on average)")


if __name__ == "__main__":
    main()
