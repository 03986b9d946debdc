"""Closed-form baseline policies (reference: mbt_gym/agents/BaselineAgents.py).  Policies are CALLERS of the hot
path: they map the observation matrix to an action matrix on the host, once per step."""
import warnings

from copy import deepcopy

import numpy as np

from mbt_gym_amd import _native
from mbt_gym_amd.agents.Agent import Agent
from mbt_gym_amd.gym.index_names import ASSET_PRICE_INDEX, CASH_INDEX, INVENTORY_INDEX, TIME_INDEX


class RandomAgent(Agent):
    """One draw from a private copy of the action space, repeated for every trajectory (AG:15-22)."""

    def __init__(self, env, seed: int = None):
        self.action_space = deepcopy(env.action_space)
        self.action_space.seed(seed)
        self.num_trajectories = env.num_trajectories

    def get_action(self, state: np.ndarray) -> np.ndarray:
        one = np.asarray(self.action_space.sample(), dtype=np.float32).reshape(1, -1)
        return np.repeat(one, self.num_trajectories, axis=0)


class FixedActionAgent(Agent):
    def __init__(self, fixed_action: np.ndarray, env):
        self.fixed_action = np.asarray(fixed_action, dtype=np.float32)
        self.env = env

    def get_action(self, state: np.ndarray) -> np.ndarray:
        return np.repeat(self.fixed_action.reshape(1, -1), self.env.num_trajectories, axis=0)

    def device_policy(self) -> _native.MbtPolicy:
        """The same policy as a descriptor the fused rollout kernel evaluates on the device."""
        pol = _native.MbtPolicy(kind=_native.POLICY_FIXED)
        for j, value in enumerate(self.fixed_action.reshape(-1)):
            pol.params[j] = float(value)
        return pol


class FixedSpreadAgent(Agent):
    """Quotes half_spread -/+ offset on bid/ask (AG:34-42)."""

    def __init__(self, env, half_spread: float = 1.0, offset: float = 0.0):
        self.half_spread, self.offset, self.env = half_spread, offset, env

    def get_action(self, state: np.ndarray) -> np.ndarray:
        quote = np.array([[self.half_spread - self.offset, self.half_spread + self.offset]], dtype=np.float32)
        return np.repeat(quote, self.env.num_trajectories, axis=0)

    def device_policy(self) -> _native.MbtPolicy:
        pol = _native.MbtPolicy(kind=_native.POLICY_FIXED)
        pol.params[0], pol.params[1] = self.half_spread - self.offset, self.half_spread + self.offset
        return pol


class AvellanedaStoikovAgent(Agent):
    """The Avellaneda-Stoikov (2008) quotes: reservation-price shift q gamma sigma^2 (T-t) around a spread of
    gamma sigma^2 (T-t) + (2/gamma) ln(1 + gamma/kappa) (AG:52-83).  Expects un-normalised observations."""

    def __init__(self, risk_aversion: float = 0.1, env=None):
        assert env is not None
        self.risk_aversion = risk_aversion
        self.env = env
        self.terminal_time = env.terminal_time
        self.volatility = env.model_dynamics.midprice_model.volatility
        self.rate_of_arrival = env.model_dynamics.arrival_model.intensity
        self.fill_exponent = env.model_dynamics.fill_probability_model.fill_exponent

    def get_action(self, state: np.ndarray) -> np.ndarray:
        q = state[:, INVENTORY_INDEX].astype(np.float64)
        tau = self.terminal_time - state[:, TIME_INDEX].astype(np.float64)
        g, s2, k = self.risk_aversion, self.volatility**2, self.fill_exponent
        shift = q * g * s2 * tau
        spread = 2 / k if g == 0 else g * s2 * tau + 2 / g * np.log(1 + g / k)
        action = np.stack((shift + spread / 2, -shift + spread / 2), axis=1)
        if action.min() < 0:
            warnings.warn("Avellaneda-Stoikov agent is quoting a negative spread")
        return action.astype(np.float32)

    def device_policy(self) -> _native.MbtPolicy:
        pol = _native.MbtPolicy(kind=_native.POLICY_AVELLANEDA_STOIKOV)
        pol.params[0] = self.risk_aversion
        return pol


class CarteaJaimungalMmAgent(Agent):
    """Optimal market-making quotes of Cartea, Jaimungal & Penalva (2015), section 10.2, for the running-inventory
    criterion: omega(t) = exp(A (T - t)) z with A tridiagonal over the inventory grid -Q..Q, h = ln(omega) / kappa,
    depth_bid(q) = 1/kappa - h(q+1) + h(q), depth_ask(q) = 1/kappa - h(q-1) + h(q) (reference:
    agents/BaselineAgents.py:86-170).  At the inventory limits the blocked side quotes a very large depth.

    The whole (time step, inventory) table is built once - one matrix exponential for the step dt, then n_steps
    mat-vecs backwards from T - instead of one 201x201 expm per get_action call; `device_policy()` hands the same
    table to the fused rollout kernel."""

    large_depth = 10_000

    def __init__(self, env=None, max_inventory: int = None):
        from mbt_gym_amd.gym.ModelDynamics import LimitOrderModelDynamics
        from mbt_gym_amd.rewards.RewardFunctions import CjMmCriterion, PnL

        assert env is not None
        assert isinstance(env.model_dynamics, LimitOrderModelDynamics), "Trader must be type LimitOrderTrader"
        assert isinstance(env.reward_function, (CjMmCriterion, PnL)), "Reward function for CjMmAgent is incorrect."
        self.env = env
        self.kappa = env.model_dynamics.fill_probability_model.fill_exponent
        self.num_trajectories = env.num_trajectories
        self.inventory_neutral = isinstance(env.reward_function, PnL)
        if self.inventory_neutral:
            self.risk_neutral_action = np.full((env.num_trajectories, env.action_space.shape[0]), 1 / self.kappa, dtype=np.float32)
            return
        self.phi = env.reward_function.per_step_inventory_aversion
        self.alpha = env.reward_function.terminal_inventory_aversion
        assert env.reward_function.inventory_exponent == 2.0, "Inventory exponent must be = 2."
        self.terminal_time = env.terminal_time
        self.lambdas = np.asarray(env.model_dynamics.arrival_model.intensity, dtype=np.float64).reshape(-1)
        self.max_inventory = int(env.max_inventory if max_inventory is None else max_inventory)
        self.a_matrix, self.z_vector = self._calculate_a_and_z()
        self._h = None  # (n_steps + 1, 2Q + 1): h at t_k = k dt

    def _calculate_a_and_z(self):
        size = 2 * self.max_inventory + 1
        q = self.max_inventory - np.arange(size)  # row i holds inventory Q - i (descending, as in the reference)
        a = np.diag(-self.phi * self.kappa * q.astype(np.float64) ** 2)
        a += np.diag(np.full(size - 1, self.lambdas[0] * np.exp(-1)), k=1)
        a += np.diag(np.full(size - 1, self.lambdas[1] * np.exp(-1)), k=-1)
        z = np.exp(-self.alpha * self.kappa * q.astype(np.float64) ** 2).reshape(-1, 1)
        return a, z

    def _calculate_omega(self, current_time: float) -> np.ndarray:
        """Equation (10.11) of [CJP15] at an arbitrary time."""
        from scipy.linalg import expm

        return expm(self.a_matrix * (self.terminal_time - current_time)) @ self.z_vector

    def _calculate_ht(self, current_time: float) -> np.ndarray:
        return np.log(self._calculate_omega(current_time)) / self.kappa

    def h_table(self) -> np.ndarray:
        """h(t_k, .) for k = 0..n_steps on the environment's time grid; column j is what the reference reads for inventory
        q = j - Q.  The matrix of `_calculate_a_and_z` is laid out with row i <-> inventory Q - i, but the reference indexes
        the resulting vector with Q + q WITHOUT flipping it (AG:117-136); for symmetric intensities h is even and the two
        conventions coincide, for asymmetric ones they do not - the reference's indexing is reproduced here."""
        if self._h is None:
            from scipy.linalg import expm

            n, dt = self.env.n_steps, self.env.step_size
            step = expm(self.a_matrix * dt)
            omega = np.empty((n + 1, self.a_matrix.shape[0]))
            omega[n] = self.z_vector[:, 0]
            for k in range(n - 1, -1, -1):
                omega[k] = step @ omega[k + 1]
            self._h = np.log(omega) / self.kappa  # NOT flipped: indexed with Q + q like the reference (AG:117-119)
        return self._h

    def depth_table(self) -> np.ndarray:
        """(n_steps + 1, 2Q + 1, 2) optimal (bid, ask) depths over (time step, inventory -Q..Q)."""
        h = self.h_table()
        up = np.concatenate((h[:, 1:], h[:, -1:]), axis=1)    # h(q + 1), clipped at +Q
        down = np.concatenate((h[:, :1], h[:, :-1]), axis=1)  # h(q - 1), clipped at -Q
        bid = 1 / self.kappa - up + h + self.large_depth * (up == h)
        ask = 1 / self.kappa - down + h + self.large_depth * (down == h)
        return np.stack((bid, ask), axis=2)

    def get_action(self, state: np.ndarray) -> np.ndarray:
        if self.inventory_neutral:
            return self.risk_neutral_action
        assert state[0, TIME_INDEX] == state[-1, TIME_INDEX], "CarteaJaimungalMmAgent needs a uniform time stamp."
        k = int(np.clip(np.rint(state[0, TIME_INDEX] / self.env.step_size), 0, self.env.n_steps))
        cols = np.clip(self.max_inventory + state[:, INVENTORY_INDEX], 0, 2 * self.max_inventory).astype(int)
        return self.depth_table()[k, cols].astype(np.float32)

    def calculate_true_value_function(self, state: np.ndarray) -> np.ndarray:
        """h(t, q) + cash + q S: the closed-form value the Monte-Carlo mean of the total reward must match."""
        h_t = self._calculate_ht(float(state[0, TIME_INDEX]))[:, 0]  # indexed with Q + q, un-flipped (AG:160-167)
        cols = np.clip(self.max_inventory + state[:, INVENTORY_INDEX], 0, 2 * self.max_inventory).astype(int)
        return h_t[cols] + state[:, CASH_INDEX] + state[:, INVENTORY_INDEX] * state[:, ASSET_PRICE_INDEX]

    def device_policy(self) -> _native.MbtPolicy:
        if self.inventory_neutral:
            pol = _native.MbtPolicy(kind=_native.POLICY_FIXED)
            pol.params[0] = pol.params[1] = 1 / self.kappa
            return pol
        return _native.table_policy(self.depth_table(), self.max_inventory)


class CarteaJaimungalOeAgent(Agent):
    """Optimal liquidation/acquisition speed of Cartea, Jaimungal & Penalva (2015), p. 147, for trading-with-speed dynamics
    with temporary and permanent impact (reference: agents/BaselineAgents.py:173-210):

        v(t) = -sign(q0) gamma q0 (zeta e^{gamma (T-t)} + e^{-gamma (T-t)}) / (zeta e^{gamma T} - e^{-gamma T}),
        gamma = sqrt(phi / k),  zeta = (alpha - b/2 + sqrt(k phi)) / (alpha - b/2 - sqrt(k phi))

    with k / b the temporary / permanent impact coefficients.  The speed depends on time only, so the whole episode is an
    open-loop schedule: `device_policy()` hands it to the fused rollout kernel as a time table."""

    def __init__(self, phi: float = 2 * 10 ** (-4), alpha: float = 0.0001, env=None):
        from mbt_gym_amd.gym.ModelDynamics import TradinghWithSpeedModelDynamics

        assert env is not None
        assert isinstance(env.model_dynamics, TradinghWithSpeedModelDynamics), "Trader must be type TradinghWithSpeedTrader"
        self.phi, self.alpha, self.env = phi, alpha, env
        self.price_impact_model = env.model_dynamics.price_impact_model
        self.terminal_time = env.terminal_time
        self.temporary_price_impact = self.price_impact_model.temporary_impact_coefficient
        self.permanent_price_impact = self.price_impact_model.permanent_impact_coefficient
        self.num_trajectories = env.num_trajectories

    def speed_at(self, time) -> np.ndarray:
        k, b = self.temporary_price_impact, self.permanent_price_impact
        gamma = np.sqrt(self.phi / k)
        root = np.sqrt(k * self.phi)
        zeta = (self.alpha - 0.5 * b + root) / (self.alpha - 0.5 * b - root)
        q0 = self.env.initial_inventory
        left = self.terminal_time - np.asarray(time, dtype=np.float64)
        magnitude = gamma * q0 * (zeta * np.exp(gamma * left) + np.exp(-gamma * left)) / (
            zeta * np.exp(gamma * self.terminal_time) - np.exp(-gamma * self.terminal_time)
        )
        return -np.sign(q0) * magnitude

    def get_action(self, state: np.ndarray) -> np.ndarray:
        return np.full((self.num_trajectories, 1), self.speed_at(state[0, TIME_INDEX]), dtype=np.float32)

    def schedule(self) -> np.ndarray:
        """(n_steps + 1, 1): the speed quoted at every observation time of the environment's grid."""
        times = np.float32(np.arange(self.env.n_steps + 1) * self.env.step_size)  # the float32 time the observation carries
        return self.speed_at(times).reshape(-1, 1).astype(np.float32)

    def device_policy(self) -> _native.MbtPolicy:
        return _native.schedule_policy(self.schedule())
