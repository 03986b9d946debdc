"""Risk statistics of rolled-out trajectories (reference: mbt_gym/gym/backtesting.py:11-62 - Sharpe and Sortino ratios,
maximum drawdown of the marked-to-market portfolio value).

The reference computes them for ONE trajectory (it asserts `env.num_trajectories == 1`).  Here an episode of every lane is
one fused rollout, so the same three formulas are evaluated per lane over the whole batch: with one trajectory the
functions return the reference's scalar, with N > 1 an (N,) array - a distribution of backtests instead of a single draw.
Host-side arithmetic on the recording; nothing here is on the step path."""
import warnings

import numpy as np

from mbt_gym_amd.gym.helpers.generate_trajectory import generate_trajectory
from mbt_gym_amd.gym.index_names import ASSET_PRICE_INDEX, CASH_INDEX, INVENTORY_INDEX


def portfolio_returns(env, agent):
    """(N, n_steps) relative changes of cash + inventory x midprice along each trajectory, as the reference forms them:
    the difference of consecutive values over the LATER value (backtesting.py:23)."""
    obs, _, _ = generate_trajectory(env, agent)
    obs = np.asarray(obs, dtype=np.float64)
    value = obs[:, CASH_INDEX, :] + obs[:, INVENTORY_INDEX, :] * obs[:, ASSET_PRICE_INDEX, :]
    if np.min(np.abs(value)) < 1e-6:
        warnings.warn("Runtime Warning: Division by Zero")
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.diff(value, axis=-1) / value[:, 1:]


def _per_lane(values):
    return float(values[0]) if values.shape[0] == 1 else values


def get_sharpe_ratio(env, agent, risk_free_rate: float = 0.099):
    """sqrt(n_steps)-annualised: (mean return x n_steps - risk-free rate) / (std of returns x sqrt(n_steps))."""
    returns = portfolio_returns(env, agent)
    mean = returns.mean(axis=-1)
    if np.any(mean < 0):
        warnings.warn("Warning: Mean Return % is negative. Sharpe Ratio may not be appropriate.")
    return _per_lane((mean * env.n_steps - risk_free_rate) / (returns.std(axis=-1) * np.sqrt(env.n_steps)))


def get_sortino_ratio(env, agent, risk_free_rate: float = 0.099):
    """The Sharpe ratio with the spread of the LOSSES only in the denominator."""
    returns = portfolio_returns(env, agent)
    mean = returns.mean(axis=-1)
    if np.any(mean < 0):
        warnings.warn("Warning: Mean Return % is negative. Sortino Ratio may not be appropriate.")
    losses = np.where(returns < 0, returns, np.nan)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)  # a lane without a single loss has no downside deviation: NaN, as numpy's empty std
        downside = np.nanstd(losses, axis=-1)
    return _per_lane((mean * env.n_steps - risk_free_rate) / (downside * np.sqrt(env.n_steps)))


def get_maximum_drawdown(env, agent):
    """The largest relative fall of the compounded return path from its running peak (<= 0)."""
    growth = np.cumprod(portfolio_returns(env, agent) + 1.0, axis=-1)
    peak = np.maximum.accumulate(growth, axis=-1)
    return _per_lane((growth / peak - 1.0).min(axis=-1))
