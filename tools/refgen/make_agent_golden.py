#!/usr/bin/env python3
"""Generate tests/golden/agents_*.npz by running the REAL reference agents (read-only at /root/reference).

Build-container only (see make_golden.py).  Only DATA is written: the states fed to the reference's
`get_action` / `calculate_true_value_function` and what it returned.

  agents_cj_asymmetric.npz   CarteaJaimungalMmAgent with intensity (140, 60): for symmetric intensities h(t, q)
                             is even in q and cannot tell an index convention from its mirror image; this one can
                             (mbt_gym/agents/BaselineAgents.py:112-136, :151-170).
  agents_reward_scaling.npz  TradingEnvironment(normalise_rewards=True, num_trajectories=1).reward_scaling (the reference's calibration only
                             broadcasts for num_trajectories 1 or 100 000: MD:109 against TE:173-178) for the default
                             (normalised actions) and an un-normalised environment, seeded - a Monte-Carlo mean over
                             100 000 lanes x n_steps (mbt_gym/gym/TradingEnvironment.py:90-94, :329-343): compared
                             statistically, with its standard error stored beside it.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/refgen/make_agent_golden.py
"""
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "gym_standin"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

from mbt_gym.agents.BaselineAgents import CarteaJaimungalMmAgent  # noqa: E402
from mbt_gym.gym.ModelDynamics import LimitOrderModelDynamics  # noqa: E402
from mbt_gym.gym.TradingEnvironment import TradingEnvironment  # noqa: E402
from mbt_gym.rewards.RewardFunctions import CjMmCriterion  # noqa: E402
from mbt_gym.stochastic_processes.arrival_models import PoissonArrivalModel  # noqa: E402
from mbt_gym.stochastic_processes.fill_probability_models import ExponentialFillFunction  # noqa: E402
from mbt_gym.stochastic_processes.midprice_models import BrownianMotionMidpriceModel  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def market(n, ns, lam, kappa=1.5, sigma=2.0):
    dt = 1.0 / ns
    return LimitOrderModelDynamics(
        midprice_model=BrownianMotionMidpriceModel(volatility=sigma, initial_price=100, terminal_time=1.0, step_size=dt, num_trajectories=n),
        arrival_model=PoissonArrivalModel(intensity=np.array(lam, dtype=float), step_size=dt, num_trajectories=n),
        fill_probability_model=ExponentialFillFunction(fill_exponent=kappa, step_size=dt, num_trajectories=n),
        num_trajectories=n)


def cj_asymmetric():
    q_max, ns, lam = 6, 50, (140.0, 60.0)
    inventories = np.arange(-q_max - 1, q_max + 2, dtype=np.float64)  # one lane per inventory, incl. one beyond each limit
    n = len(inventories)
    env = TradingEnvironment(
        terminal_time=1.0, n_steps=ns, seed=3, initial_inventory=0, max_inventory=q_max, num_trajectories=n,
        reward_function=CjMmCriterion(per_step_inventory_aversion=0.05, terminal_inventory_aversion=0.02, terminal_time=1.0),
        model_dynamics=market(n, ns, lam), normalise_action_space=False, normalise_observation_space=False)
    agent = CarteaJaimungalMmAgent(env=env)
    steps = np.array([0, 1, 17, 25, 49, 50])
    actions, h_rows = [], []
    for k in steps:
        state = np.zeros((n, 4))
        state[:, 1] = inventories
        state[:, 2] = k / ns
        state[:, 3] = 100.0
        actions.append(np.array(agent.get_action(state), dtype=np.float64))
        h_rows.append(np.array(agent._calculate_ht(k / ns), dtype=np.float64)[:, 0])
    np.savez_compressed(
        os.path.join(OUT, "agents_cj_asymmetric.npz"), inventories=inventories, time_steps=steps, actions=np.stack(actions),
        h=np.stack(h_rows), intensity=np.array(lam), kappa=1.5, phi=0.05, alpha=0.02, max_inventory=q_max, n_steps=ns)
    a0 = actions[0][q_max + 1]
    print(f"agents_cj_asymmetric: q=0, t=0 -> (bid, ask) = ({a0[0]:.6f}, {a0[1]:.6f})")


def reward_scaling():
    out = {}
    for tag, kw in (("default", dict()), ("raw_actions", dict(normalise_action_space=False, normalise_observation_space=False))):
        with contextlib.redirect_stdout(io.StringIO()):
            env = TradingEnvironment(terminal_time=1.0, n_steps=40, seed=7, num_trajectories=1, max_inventory=100,
                                     model_dynamics=market(1, 40, (100.0, 100.0)), normalise_rewards=True, **kw)
        out[tag] = float(env.reward_scaling)
        print(f"agents_reward_scaling[{tag}]: reward_scaling = {env.reward_scaling:.6f}  (1 / {1 / env.reward_scaling:.6f})")
    np.savez_compressed(os.path.join(OUT, "agents_reward_scaling.npz"), n_steps=40, intensity=np.array([100.0, 100.0]), kappa=1.5, sigma=2.0,
                        max_inventory=100, lanes=100_000, **out)


if __name__ == "__main__":
    cj_asymmetric()
    reward_scaling()
