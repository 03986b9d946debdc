"""Fill-probability models with a device implementation
(reference: mbt_gym/stochastic_processes/fill_probability_models.py).

ExponentialFillFunction (FILL:42-65):  fill_s = U_s < exp(-kappa * depth_s);  max_depth = -ln(0.01) / kappa.
"""
from typing import Optional

import numpy as np

from mbt_gym_amd import _native
from mbt_gym_amd.stochastic_processes.StochasticProcessModel import DeviceResidentError, StochasticProcessModel

_EMPTY = np.array([[]])


class FillProbabilityModel(StochasticProcessModel):
    def get_fills(self, depths: np.ndarray) -> np.ndarray:
        raise DeviceResidentError(
            "fills are drawn inside the fused HIP step kernel; enable env.record_events(True) and read "
            "env.last_fills after a step."
        )

    @property
    def max_depth(self) -> float:
        raise NotImplementedError


class ExponentialFillFunction(FillProbabilityModel):
    device_kind = _native.FILL_EXPONENTIAL

    def __init__(
        self, fill_exponent: float = 1.5, step_size: float = 0.1, num_trajectories: int = 1, seed: Optional[int] = None
    ):
        self.fill_exponent = fill_exponent
        super().__init__(_EMPTY, _EMPTY, step_size, 0.0, _EMPTY, num_trajectories, seed)

    def _get_fill_probabilities(self, depths: np.ndarray) -> np.ndarray:
        """Closed-form probability (host utility for agents and plots; the kernel has its own evaluation)."""
        return np.exp(-self.fill_exponent * np.asarray(depths))

    @property
    def max_depth(self) -> float:
        return -np.log(0.01) / self.fill_exponent  # the depth whose fill probability is 1 %

    def device_params(self):
        return dict(fill_kind=self.device_kind, fill_exponent=self.fill_exponent)
