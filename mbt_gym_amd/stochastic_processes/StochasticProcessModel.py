"""Plugin base class of the stochastic-process family (midprice / arrival / fill-probability models).

Same constructor surface and attributes as the reference's `StochasticProcessModel`
(mbt_gym/stochastic_processes/StochasticProcessModel.py:8-53) - `min_value`, `max_value` (shape (1, d)),
`step_size`, `terminal_time`, `num_trajectories`, `initial_state` (1, d), `current_state` (N, d), `seed_` - but
here a process is a DESCRIPTOR: it names a device implementation (`device_kind`) and carries its parameters
(`device_params()`).  Inside an environment its numerics run in the fused HIP step kernel (csrc/step_kernel.hpp) and
`update()` raises: the kernel, not the caller, advances it.  ON ITS OWN (not attached to an environment: a midprice path, an
arrival stream - the reference's objects can be driven that way, SP:33-35) a built-in process keeps a host copy of its (N, d)
state, draws from its own numpy Generator exactly like the reference's class does, and has the arithmetic of each call
evaluated on the device in double (`mbt_process_evaluate_host`): there is no CPU implementation of the maths here either, and
a process seeded like the reference's walks the reference's path.
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
        self._host_state = None  # stand-alone use: the (N, d) float64 state between update() calls
        self._host_callback = False  # a NumPy-only subclass inside an environment: ITS update() advances its state, on the host

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
    def _stand_alone(self) -> bool:
        # not handed to a TradingEnvironment: driven by the caller - or handed to one as a host-callback plugin (a NumPy-only
        # subclass): its own methods keep its state on the host, the environment copies it into the state matrix (TE:206-211)
        return self._env is None or self._host_callback

    @property
    def current_state(self) -> np.ndarray:
        """Inside an environment: the process's columns of the device-resident state matrix (host copy).  On its own: the
        (N, d) float64 state its update() calls have produced since reset() (SP:30-31)."""
        if self._stand_alone:
            if self._host_state is None or self._host_state.shape[0] != self.num_trajectories:
                self._host_state = self.initial_vector_state.copy()
            return self._host_state
        if not self._env.has_device_state:
            return self.initial_vector_state
        lo, hi = self._columns
        return self._env.state[:, lo:hi]

    @current_state.setter
    def current_state(self, value):
        if not self._stand_alone:
            raise DeviceResidentError("inside an environment the state matrix lives in HBM: use env.set_state()")
        self._host_state = np.array(value, dtype=np.float64)

    def reset(self):
        """SP:30-31 for stand-alone use; inside an environment TradingEnvironment.reset() re-initialises the device state."""
        self._host_state = None

    def _evaluate(self, op, a, b=None, c=None, d=None):
        """One method call's arithmetic on host arrays, on the device in double (include/mbt_env.h: mbt_process_evaluate_host)."""
        from mbt_gym_amd import _native

        return _native.process_evaluate(op, self.device_params(), a, b, c, d)

    def seed(self, seed: int = None):
        # The reference gives process i its own generator seeded seed+i+1 (TE:345-348).  On the device all
        # processes read disjoint words of ONE Philox stream keyed by the environment seed; the per-process
        # number is kept for API parity only.
        self.rng = np.random.default_rng(seed)
        self.seed_ = seed

    def update(self, arrivals, fills, action, state=None):
        if not self._stand_alone:
            raise DeviceResidentError(
                f"{type(self).__name__}.update runs inside the fused HIP step kernel; call TradingEnvironment.step()."
            )
        return self._update_stand_alone(arrivals, fills, action)

    def _update_stand_alone(self, arrivals, fills, action):
        raise DeviceResidentError(f"{type(self).__name__} has no host-callable update (its device form runs inside an environment only)")
