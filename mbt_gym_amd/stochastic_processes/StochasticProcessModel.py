"""Plugin base class of the stochastic-process family (midprice / arrival / fill-probability models).

Same constructor surface and attributes as the reference's `StochasticProcessModel`
(mbt_gym/stochastic_processes/StochasticProcessModel.py:8-53) - `min_value`, `max_value` (shape (1, d)),
`step_size`, `terminal_time`, `num_trajectories`, `initial_state` (1, d), `current_state` (N, d), `seed_` - but
here a process is a DESCRIPTOR: it names a device implementation (`device_kind`) and carries its parameters
(`device_params()`).  The numerics run inside the fused HIP step kernel (csrc/step_kernel.hpp); nothing is
evaluated on the host, so `update()` on a descriptor raises instead of silently running a CPU path.
"""
import abc
from typing import Optional

import numpy as np


class DeviceResidentError(NotImplementedError):
    """Raised when host code asks a descriptor to run numerics that only exist inside the HIP kernel."""


class StochasticProcessModel(metaclass=abc.ABCMeta):
    #: kind code of the HIP implementation (include/mbt_env.h), None when the class has no device kernel
    device_kind: Optional[int] = None

    def __init__(
        self,
        min_value: np.ndarray,
        max_value: np.ndarray,
        step_size: float,
        terminal_time: float,
        initial_state: np.ndarray,
        num_trajectories: int = 1,
        seed: int = None,
    ):
        self.min_value = np.asarray(min_value, dtype=np.float64)
        self.max_value = np.asarray(max_value, dtype=np.float64)
        self.initial_state = np.asarray(initial_state, dtype=np.float64)
        for name in ("initial_state", "min_value", "max_value"):
            value = getattr(self, name)
            # same contract as SP:41-46
            assert value.ndim == 2 and value.shape[0] == 1, f"Attribute {name} must be a vector of shape (1, state_size)."
        self.step_size = step_size
        self.terminal_time = terminal_time
        self.num_trajectories = num_trajectories
        self.seed_ = seed
        # API parity only (SP:27): the reference draws from this generator, the device draws from Philox keyed by the
        # environment's seed - nothing on the path consumes it
        self.rng = np.random.default_rng(seed)
        self._env = None  # set by TradingEnvironment: (env, first column, last column)
        self._columns = None

    # ---- descriptor side --------------------------------------------------------------------------------
    def device_params(self) -> dict:
        """mbt_config fields this process contributes."""
        return {}

    @property
    def state_dim(self) -> int:
        return int(self.initial_state.shape[1])

    def _attach(self, env, lo: int, hi: int):
        self._env, self._columns = env, (lo, hi)

    # ---- reference surface ------------------------------------------------------------------------------
    @property
    def initial_vector_state(self) -> np.ndarray:
        """(N, d) tiling of `initial_state` (SP:48-53)."""
        return np.repeat(self.initial_state, self.num_trajectories, axis=0)

    @property
    def current_state(self) -> np.ndarray:
        """The process's columns of the device-resident state matrix (host copy)."""
        if self._env is None or not self._env.has_device_state:
            return self.initial_vector_state
        lo, hi = self._columns
        return self._env.state[:, lo:hi]

    def reset(self):
        """The device state is re-initialised by TradingEnvironment.reset(); nothing to do on the host."""

    def seed(self, seed: int = None):
        # The reference gives process i its own generator seeded seed+i+1 (TE:345-348).  On the device all
        # processes read disjoint words of ONE Philox stream keyed by the environment seed; the per-process
        # number is kept for API parity only.
        self.rng = np.random.default_rng(seed)
        self.seed_ = seed

    def update(self, arrivals, fills, action, state=None):
        raise DeviceResidentError(
            f"{type(self).__name__}.update runs inside the fused HIP step kernel; call TradingEnvironment.step()."
        )
