"""Midprice models with a device implementation (reference: mbt_gym/stochastic_processes/midprice_models.py).

BrownianMotionMidpriceModel (MID:36-68):  S <- S + mu dt + sigma sqrt(dt) Z
OuMidpriceModel            (MID:114-146): S <- S - theta (S - level) + sigma sqrt(dt) Z
    The mean-reversion term is NOT multiplied by dt in the reference (MID:140-143); this is reproduced.

The other reference classes (GBM, jump models, short-term-alpha, Heston, CEV) have no kernel yet: constructing
a TradingEnvironment with an unknown process raises - see SURVEY.md section 2.1 for which of them are broken
upstream.
"""
from typing import Optional

import numpy as np

from mbt_gym_amd import _native
from mbt_gym_amd.stochastic_processes.StochasticProcessModel import StochasticProcessModel

MidpriceModel = StochasticProcessModel


class _SymmetricBandMidprice(MidpriceModel):
    """A midprice whose observation bounds are initial_price -/+ half_width."""

    def __init__(self, initial_price, half_width, terminal_time, step_size, num_trajectories, seed):
        self.terminal_time = terminal_time
        top = initial_price + half_width
        super().__init__(
            min_value=np.array([[initial_price - (top - initial_price)]]),
            max_value=np.array([[top]]),
            step_size=step_size,
            terminal_time=terminal_time,
            initial_state=np.array([[initial_price]]),
            num_trajectories=num_trajectories,
            seed=seed,
        )

    @property
    def initial_price(self) -> float:
        return float(self.initial_state[0, 0])


class BrownianMotionMidpriceModel(_SymmetricBandMidprice):
    device_kind = _native.MID_BROWNIAN

    def __init__(
        self,
        drift: float = 0.0,
        volatility: float = 2.0,
        initial_price: float = 100,
        terminal_time: float = 1.0,
        step_size: float = 0.01,
        num_trajectories: int = 1,
        seed: Optional[int] = None,
    ):
        self.drift = drift
        self.volatility = volatility
        # four standard deviations of S_T (MID:67-68)
        super().__init__(initial_price, 4 * volatility * np.sqrt(terminal_time), terminal_time, step_size, num_trajectories, seed)

    def device_params(self):
        return dict(midprice_kind=self.device_kind, drift=self.drift, volatility=self.volatility, initial_price=self.initial_price)


class OuMidpriceModel(_SymmetricBandMidprice):
    device_kind = _native.MID_OU

    def __init__(
        self,
        mean_reversion_level: float = 0.0,
        mean_reversion_speed: float = 1.0,
        volatility: float = 2.0,
        initial_price: float = 100.0,
        terminal_time: float = 1.0,
        step_size: float = 0.01,
        num_trajectories: int = 1,
        seed: Optional[int] = None,
    ):
        self.mean_reversion_level = mean_reversion_level
        self.mean_reversion_speed = mean_reversion_speed
        self.volatility = volatility
        # the reference bounds OU by 4 sigma T, not 4 sigma sqrt(T) (MID:145-146)
        super().__init__(initial_price, 4 * volatility * terminal_time, terminal_time, step_size, num_trajectories, seed)

    def device_params(self):
        return dict(
            midprice_kind=self.device_kind, volatility=self.volatility, initial_price=self.initial_price,
            ou_level=self.mean_reversion_level, ou_speed=self.mean_reversion_speed,
        )
