"""Fill-probability models with a device implementation
(reference: mbt_gym/stochastic_processes/fill_probability_models.py).

ExponentialFillFunction (FILL:42-65):  fill_s = U_s < exp(-kappa * depth_s);  max_depth = -ln(0.01) / kappa.
ExogenousMmFillProbabilityModel (FILL:126-170):  fill_s = U_s < 1 for a quote at or inside the exogenous best depth
    b_s, else U_s < base * exp(-kappa * (depth_s - b_s)); contributes the two columns (b_bid, b_ask) to the state.

TriangularFillFunction / PowerFillFunction (FILL:68-123) reduce over the TRAJECTORY axis (`np.max(depths, 0)`), i.e.
they are not per-trajectory models for num_trajectories > 1; they have no device implementation.
"""
from typing import Optional, Tuple

import numpy as np

from mbt_gym_amd import _native
from mbt_gym_amd.stochastic_processes.StochasticProcessModel import DeviceResidentError, StochasticProcessModel

_EMPTY = np.array([[]])


class FillProbabilityModel(StochasticProcessModel):
    def get_fills(self, depths: np.ndarray) -> np.ndarray:
        """Inside an environment fills are drawn by the fused step kernel.  On its own (FILL:28-34) a built-in model draws its
        (N, 2) uniforms from its generator like the reference's class and has `u < p(depth)` evaluated on the device, in double."""
        if not self._stand_alone:
            raise DeviceResidentError(
                "fills are drawn inside the fused HIP step kernel; enable env.record_events(True) and read "
                "env.last_fills after a step."
            )
        if self.device_kind not in (_native.FILL_EXPONENTIAL, _native.FILL_EXOGENOUS_MM):
            raise DeviceResidentError(f"{type(self).__name__} has no host-callable get_fills (its device form runs inside an environment only)")
        depths = np.asarray(depths, dtype=np.float64)
        assert depths.shape == (self.num_trajectories, 2), (  # FILL:29-32
            "Depths must be a numpy array of shape " + f"({self.num_trajectories},2). Instead it is a numpy array of shape {depths.shape}.")
        unif = self.rng.uniform(size=(self.num_trajectories, 2))  # FILL:33
        return self._evaluate(_native.PROCESS_FILLS, unif, depths) != 0.0

    def _update_stand_alone(self, arrivals, fills, action):
        return self.current_state  # FILL:64-65, FILL:168-170: the fill models' own state never moves

    @property
    def max_depth(self) -> float:
        raise NotImplementedError


class DeviceExpressionFillModel(FillProbabilityModel):
    """The device route for USER-DEFINED fill-probability models (the reference's plugin contract, FILL:9-39).

    The reference asks a subclass for `_get_fill_probabilities(depths)` in NumPy and `max_depth`; there is no CPU path
    here to run NumPy code in the step, so a subclass ALSO states the same function as a C++ device expression:

        class PowerLawFill(DeviceExpressionFillModel):
            device_expression = "1.0 / (1.0 + pow(scale * depth, exponent))"     # in `depth` (double), `side` (0 bid / 1 ask)
            def __init__(self, scale, exponent, **kw):
                self.scale, self.exponent = scale, exponent
                super().__init__(**kw)
            def device_expression_params(self):
                return {"scale": self.scale, "exponent": self.exponent}           # at most 8, bound by name
            def _get_fill_probabilities(self, depths):                            # the reference's method (host utility)
                return 1.0 / (1.0 + (self.scale * depths) ** self.exponent)
            @property
            def max_depth(self):
                return 99.0 ** (1.0 / self.exponent) / self.scale                 # where the probability is 1 %

    The library compiles the step and rollout kernels around the expression at run time (hiprtc, a few seconds, cached
    per process: include/mbt_env.h, mbt_env_create_jit); a fill happens when the lane's uniform u < expression, compared
    in double on the de-normalised depth.  `check_device_expression()` compiles without a GPU."""

    device_kind = _native.FILL_USER
    device_expression: str = None

    def __init__(self, step_size: float = 0.1, num_trajectories: int = 1, seed: Optional[int] = None):
        if not self.device_expression:
            raise TypeError(f"{type(self).__name__} must define `device_expression` (the device form of _get_fill_probabilities)")
        super().__init__(_EMPTY, _EMPTY, step_size, 0.0, _EMPTY, num_trajectories, seed)

    def device_expression_params(self) -> dict:
        return {}

    def device_params(self):
        return dict(fill_kind=self.device_kind)

    def device_code(self):
        return self.device_expression, dict(self.device_expression_params())


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


class ExogenousMmFillProbabilityModel(FillProbabilityModel):
    """Fill probability relative to an exogenous best bid / ask depth (FILL:126-170).

    Reference behaviour kept on purpose: `update` advances the two depth processes but never copies their state into
    this model's `current_state` (FILL:168-170), and `reset` re-tiles `initial_state` (SP:30-31), so the best depths
    the environment observes and prices fills against are the two processes' INITIAL states for the whole episode.
    The device therefore needs only those two numbers; the processes' own dynamics are never evaluated."""

    device_kind = _native.FILL_EXOGENOUS_MM

    def __init__(
        self,
        exogenous_best_depth_processes: Tuple[StochasticProcessModel],
        fill_exponent: float = 1.5,
        base_fill_probability: float = 1.0,
        step_size: float = 0.1,
        num_trajectories: int = 1,
        seed: Optional[int] = None,
    ):
        assert len(exogenous_best_depth_processes) == 2, "exogenous_best_depth_processes must be length 2 (bid and ask)"
        assert all(
            len(process.initial_state) > 0 for process in exogenous_best_depth_processes
        ), "Exogenous best depth processes must have a state of at least size 1."
        self.exogenous_best_depth_processes = exogenous_best_depth_processes
        self.fill_exponent = fill_exponent
        self.base_fill_probability = base_fill_probability
        bid, ask = exogenous_best_depth_processes
        super().__init__(
            np.concatenate([bid.min_value, ask.min_value], axis=1),
            np.concatenate([bid.max_value, ask.max_value], axis=1),
            step_size,
            0.0,
            np.concatenate((bid.initial_state, ask.initial_state), axis=1),
            num_trajectories,
            seed,
        )

    def _get_fill_probabilities(self, depths: np.ndarray) -> np.ndarray:
        """Closed-form probability against the constant best depths (host utility; the kernel has its own evaluation)."""
        depths = np.asarray(depths, dtype=np.float64)
        best = self.initial_state
        return (depths > best) * self.base_fill_probability * np.exp(-self.fill_exponent * (depths - best)) + (depths <= best)

    @property
    def max_depth(self) -> float:
        return -np.log(0.01) / self.fill_exponent + np.max(self.exogenous_best_depth_processes[0].max_value)

    def device_params(self):
        if self.initial_state.shape[1] != 2:  # FILL:159-163 compares (N, 2) depths with this state
            raise DeviceResidentError("each exogenous best-depth process must have a one-dimensional state")
        return dict(
            fill_kind=self.device_kind, fill_exponent=self.fill_exponent, base_fill_probability=self.base_fill_probability,
            exogenous_depth=(float(self.initial_state[0, 0]), float(self.initial_state[0, 1])),
        )
