"""Model dynamics: how arrivals, fills and the agent's action move cash and inventory
(reference: mbt_gym/gym/ModelDynamics.py).

LimitOrderModelDynamics          (MD:87-131)   action = (bid depth, ask depth)
LimitAndMarketOrderModelDynamics (MD:179-240)  action = (bid depth, ask depth, market buy, market sell);
    market orders execute first at midprice -/+ fixed_market_half_spread and are NOT blocked by the inventory
    limit (only clipped afterwards), then limit fills as above.

AtTheTouchModelDynamics          (MD:134-176)  action in {0,1}^2: post one unit at the best bid / ask (midprice -/+
    fixed_market_half_spread); an arrival on a posted side always fills.
TradinghWithSpeedModelDynamics   (MD:243-275)  action = trading speed v: inventory += v dt, cash -= v dt (S + impact(v)),
    dt being the MIDPRICE model's step size (MD:265); needs a price impact model, no order flow.

As with the processes, a dynamics object is a descriptor: it holds the three process descriptors, names the
device implementation and owns nothing numeric on the host.  The state matrix lives in HBM; `.state` fetches
a host copy through the environment.
"""
import abc
from typing import Optional

import numpy as np

from mbt_gym_amd import _native
from mbt_gym_amd.spaces import Box, MultiBinary
from mbt_gym_amd.stochastic_processes.StochasticProcessModel import DeviceResidentError
from mbt_gym_amd.stochastic_processes.arrival_models import ArrivalModel
from mbt_gym_amd.stochastic_processes.fill_probability_models import FillProbabilityModel
from mbt_gym_amd.stochastic_processes.midprice_models import MidpriceModel


class ModelDynamics(metaclass=abc.ABCMeta):
    device_kind: Optional[int] = None
    required_processes = ()

    def __init__(
        self,
        midprice_model: MidpriceModel = None,
        arrival_model: ArrivalModel = None,
        fill_probability_model: FillProbabilityModel = None,
        price_impact_model=None,
        num_trajectories: int = 1,
        seed: int = None,
    ):
        self.midprice_model = midprice_model
        self.arrival_model = arrival_model
        self.fill_probability_model = fill_probability_model
        self.price_impact_model = price_impact_model
        self.num_trajectories = num_trajectories
        self.seed_ = seed
        self.round_initial_inventory = False
        for name in self.get_required_stochastic_processes():
            assert getattr(self, name) is not None, f"This model dynamics cannot have env.{name} to be None."
        self._env = None

    # ---- reference surface ------------------------------------------------------------------------------
    def get_required_stochastic_processes(self):
        return list(self.required_processes)

    @abc.abstractmethod
    def get_action_space(self):
        pass

    @property
    def state(self) -> Optional[np.ndarray]:
        """Host copy of the (N, D) un-normalised state (MD:39: the reference keeps the matrix here)."""
        return None if self._env is None else self._env.state

    @state.setter
    def state(self, value):
        if self._env is None:
            raise DeviceResidentError("the dynamics object is not attached to a TradingEnvironment yet")
        self._env.set_state(value)

    @property
    def fill_multiplier(self) -> np.ndarray:
        """(N, 2) signs [-1, +1]: a bid fill buys, an ask fill sells (MD:71-73; the kernel has them built in)."""
        ones = np.ones((self.num_trajectories, 1))
        return np.append(-ones, ones, axis=1)

    @property
    def midprice(self) -> np.ndarray:
        return self.midprice_model.current_state[:, 0].reshape(-1, 1)

    def update_state(self, arrivals, fills, action):
        raise DeviceResidentError("cash/inventory updates run inside the fused HIP step kernel; call env.step().")

    def get_fills(self, action):
        return None  # a stub upstream too (MD:44-45)

    def get_arrivals_and_fills(self, action):
        raise DeviceResidentError("arrivals and fills are drawn inside the fused HIP step kernel; call env.step().")

    def _get_max_depth(self) -> Optional[float]:
        return None if self.fill_probability_model is None else self.fill_probability_model.max_depth

    def _get_max_speed(self) -> Optional[float]:
        return None if self.price_impact_model is None else self.price_impact_model.max_speed

    # ---- descriptor side --------------------------------------------------------------------------------
    def device_params(self) -> dict:
        return dict(dynamics_kind=self.device_kind)


class LimitOrderModelDynamics(ModelDynamics):
    """The agent posts a bid and an ask at chosen depths every step."""

    device_kind = _native.DYN_LIMIT
    required_processes = ("arrival_model", "fill_probability_model")

    def __init__(
        self,
        midprice_model: MidpriceModel = None,
        arrival_model: ArrivalModel = None,
        fill_probability_model: FillProbabilityModel = None,
        num_trajectories: int = 1,
        seed: int = None,
        max_depth: float = None,
    ):
        super().__init__(
            midprice_model=midprice_model, arrival_model=arrival_model, fill_probability_model=fill_probability_model,
            num_trajectories=num_trajectories, seed=seed,
        )
        self.max_depth = max_depth or self._get_max_depth()
        self.round_initial_inventory = True

    def get_action_space(self):
        assert self.max_depth is not None, "For limit orders max_depth cannot be None."
        return Box(low=np.float32(0.0), high=np.float32(self.max_depth), shape=(2,))


class LimitAndMarketOrderModelDynamics(ModelDynamics):
    """Limit orders as above plus the option to cross the spread with a unit market order on either side."""

    device_kind = _native.DYN_LIMIT_AND_MARKET
    required_processes = ("arrival_model", "fill_probability_model")

    def __init__(
        self,
        midprice_model: MidpriceModel = None,
        arrival_model: ArrivalModel = None,
        fill_probability_model: FillProbabilityModel = None,
        num_trajectories: int = 1,
        seed: int = None,
        max_depth: float = None,
        fixed_market_half_spread: float = 0.5,
    ):
        super().__init__(
            midprice_model=midprice_model, arrival_model=arrival_model, fill_probability_model=fill_probability_model,
            num_trajectories=num_trajectories, seed=seed,
        )
        self.max_depth = max_depth or self._get_max_depth()
        self.fixed_market_half_spread = fixed_market_half_spread
        self.round_initial_inventory = True

    def get_action_space(self):
        assert self.max_depth is not None, "For limit orders max_depth cannot be None."
        return Box(low=np.zeros(4), high=np.array([self.max_depth, self.max_depth, 1, 1], dtype=np.float32))

    def device_params(self):
        return dict(dynamics_kind=self.device_kind, market_half_spread=self.fixed_market_half_spread)


class AtTheTouchModelDynamics(ModelDynamics):
    """The agent decides, per side, whether to post one unit at the touch."""

    device_kind = _native.DYN_AT_THE_TOUCH
    required_processes = ("arrival_model",)

    def __init__(
        self,
        midprice_model: MidpriceModel = None,
        arrival_model: ArrivalModel = None,
        fill_probability_model: FillProbabilityModel = None,
        num_trajectories: int = 1,
        fixed_market_half_spread: float = 0.5,
        seed: int = None,
    ):
        super().__init__(
            midprice_model=midprice_model, arrival_model=arrival_model, fill_probability_model=fill_probability_model,
            num_trajectories=num_trajectories, seed=seed,
        )
        self.round_initial_inventory = True
        self.fixed_market_half_spread = fixed_market_half_spread

    def get_action_space(self):
        return MultiBinary(2)

    def device_params(self):
        return dict(dynamics_kind=self.device_kind, market_half_spread=self.fixed_market_half_spread)


class TradinghWithSpeedModelDynamics(ModelDynamics):  # the class name is spelt this way in the reference (MD:243)
    """The agent chooses a trading speed (positive buys, negative sells); price impact makes it costly."""

    device_kind = _native.DYN_SPEED
    required_processes = ("price_impact_model",)

    def __init__(
        self,
        midprice_model: MidpriceModel = None,
        price_impact_model=None,
        num_trajectories: int = 1,
        seed: int = None,
        max_speed: float = None,
    ):
        super().__init__(midprice_model=midprice_model, price_impact_model=price_impact_model, num_trajectories=num_trajectories, seed=seed)
        self.max_speed = max_speed or self._get_max_speed()
        self.round_initial_inventory = False

    def get_action_space(self):
        return Box(low=np.float32([-self.max_speed]), high=np.float32([self.max_speed]))


TradingWithSpeedModelDynamics = TradinghWithSpeedModelDynamics  # correctly spelt alias
