"""Plugin subclasses as a user of the REFERENCE writes them: NumPy only, against the reference's plugin contract
(`FillProbabilityModel._get_fill_probabilities` FILL:22-34, `ArrivalModel.get_arrivals` ARR:27-29, `RewardFunction.calculate`
RW:10-13, `MidpriceModel.update` SP:33-35, `PriceImpactModel.get_impact` IMP:25-27), with no device expression and no knowledge of
this package.

ONE source for both sides of the parity tests: `define(...)` is handed the base classes of whichever package the classes are
to live in - the reference's (`tools/refgen/make_golden.py`, build container only: the fixtures `user_fill_and_reward`,
`user_fill_hawkes_market_normalised`, `user_seasonal_arrivals`, `user_cross_hawkes`, `user_cev_midprice`, `user_two_factor_midprice(_normalised)`
are the REAL reference running these classes) or
mbt_gym_amd's (tests/test_gpu_host_callbacks.py: the same classes, unmodified, in `env.step()` through the host-callback
route of include/mbt_env.h).  Nothing here imports either package."""
import types

import numpy as np


def define(FillProbabilityModel, ArrivalModel, RewardFunction, index_names, MidpriceModel=None, PriceImpactModel=None):
    """The classes, bound to the given plugin base classes and state-column indices."""
    MidpriceModel = MidpriceModel or ArrivalModel.__mro__[1]  # (MidpriceModel IS StochasticProcessModel, MID:9)
    PriceImpactModel = PriceImpactModel or MidpriceModel
    CASH_INDEX, INVENTORY_INDEX, TIME_INDEX, ASSET_PRICE_INDEX = (
        index_names.CASH_INDEX, index_names.INVENTORY_INDEX, index_names.TIME_INDEX, index_names.ASSET_PRICE_INDEX)

    class UserPowerLawFill(FillProbabilityModel):
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

    class UserExponentialInventoryCost(RewardFunction):
        """PnL - dt phi (exp(eta |q'|) - 1) - alpha [terminal] q'^2: an inventory cost that grows exponentially."""

        def __init__(self, phi, eta, alpha):
            self.phi, self.eta, self.alpha = phi, eta, alpha

        def calculate(self, current_state, action, next_state, is_terminal_step=False):
            value = lambda s: s[:, CASH_INDEX] + s[:, INVENTORY_INDEX] * s[:, ASSET_PRICE_INDEX]  # noqa: E731
            dt = next_state[:, TIME_INDEX] - current_state[:, TIME_INDEX]
            q = next_state[:, INVENTORY_INDEX]
            return value(next_state) - value(current_state) - dt * self.phi * (np.exp(self.eta * np.abs(q)) - 1.0) - self.alpha * int(is_terminal_step) * q**2

        def reset(self, initial_state):
            pass

    class UserSeasonalArrivals(ArrivalModel):
        """A time-of-day intensity profile.  Stateless; the reference hands the state matrix to update() (TE:206-211), which is
        where a plugin written against its API learns the time."""

        def __init__(self, base, amplitude, period, step_size, num_trajectories, seed=None):
            self.base, self.amplitude, self.period, self.time = np.array(base, dtype=float), amplitude, period, 0.0
            super().__init__(min_value=np.array([[]]), max_value=np.array([[]]), step_size=step_size, terminal_time=0.0,
                             initial_state=np.array([[]]), num_trajectories=num_trajectories, seed=seed)

        def reset(self):
            super().reset()
            self.time = 0.0

        def update(self, arrivals, fills, actions, state=None):
            self.time = state[0, TIME_INDEX]

        def get_arrivals(self):
            unif = self.rng.uniform(size=(self.num_trajectories, 2))
            return unif < self.base * (1.0 + self.amplitude * np.cos(2 * np.pi * self.time / self.period)) * self.step_size

    class UserCrossExcitingHawkes(ArrivalModel):
        """An arrival model WITH STATE (SP:8-53: a subclass carries its own (N, d) current_state): two intensities like the
        reference's Hawkes model (ARR:86-126) in which an arrival on one side also excites the other."""

        def __init__(self, baseline, speed, jump, cross, step_size, terminal_time, num_trajectories, seed=None):
            self.baseline, self.speed, self.jump, self.cross = np.array(baseline, dtype=float).reshape(1, 2), speed, jump, cross
            super().__init__(min_value=np.zeros((1, 2)), max_value=self.baseline * 10, step_size=step_size, terminal_time=terminal_time,
                             initial_state=self.baseline, num_trajectories=num_trajectories, seed=seed)

        def update(self, arrivals, fills, actions, state=None):
            lam = self.current_state
            self.current_state = lam + self.speed * (self.baseline - lam) * self.step_size + self.jump * arrivals + self.cross * arrivals[:, ::-1]

        def get_arrivals(self):
            unif = self.rng.uniform(size=(self.num_trajectories, 2))
            return unif < self.current_state * self.step_size

    class UserStateReadingArrivals(ArrivalModel):
        """An arrival model whose update() READS THE STATE MATRIX it is handed (TE:206-211).  At that point of the reference's step cash,
        inventory and time have advanced (TE:213-216), the midprice - first in the registry - has too, the model's own columns have
        not.  Two intensities that relax to a baseline tilted by the NEW time, lean against the NEW price's distance from a
        reference price (sellers arrive as it rises) and thin out as the agent's NEW inventory grows."""

        def __init__(self, baseline, speed, tilt, sensitivity, crowding, reference_price, step_size, terminal_time, num_trajectories, seed=None):
            self.baseline, self.speed, self.tilt = np.array(baseline, dtype=float).reshape(1, 2), speed, tilt
            self.sensitivity, self.crowding, self.reference_price = sensitivity, crowding, reference_price
            super().__init__(min_value=np.zeros((1, 2)), max_value=self.baseline * 10, step_size=step_size, terminal_time=terminal_time,
                             initial_state=self.baseline, num_trajectories=num_trajectories, seed=seed)

        def update(self, arrivals, fills, actions, state=None):
            lam = self.current_state
            t, price, q = state[0, TIME_INDEX], state[:, ASSET_PRICE_INDEX:ASSET_PRICE_INDEX + 1], state[:, INVENTORY_INDEX:INVENTORY_INDEX + 1]
            target = self.baseline * (1.0 + self.tilt * t) + self.sensitivity * (price - self.reference_price) * np.array([[-1.0, 1.0]])
            self.current_state = np.maximum(lam + self.speed * (target - lam) * self.step_size - self.crowding * np.abs(q) * lam * self.step_size, 0.0)

        def get_arrivals(self):
            unif = self.rng.uniform(size=(self.num_trajectories, 2))
            return unif < self.current_state * self.step_size

    class UserCevMidprice(MidpriceModel):
        """dS = mu S dt + sigma S^gamma sqrt(dt) Z: constant elasticity of variance, written for any number of trajectories (the
        reference's own CEV class broadcasts (N,) noise against an (N, 1) state and cannot be used for N > 1, MID:401-409)."""

        def __init__(self, drift, volatility, gamma, initial_price, lo, hi, terminal_time, step_size, num_trajectories, seed=None):
            self.drift, self.volatility, self.gamma = drift, volatility, gamma
            super().__init__(min_value=np.array([[lo]]), max_value=np.array([[hi]]), step_size=step_size, terminal_time=terminal_time,
                             initial_state=np.array([[initial_price]]), num_trajectories=num_trajectories, seed=seed)

        def update(self, arrivals, fills, actions, state=None):
            s = self.current_state
            z = self.rng.normal(size=(self.num_trajectories, 1))
            self.current_state = s + self.drift * s * self.step_size + self.volatility * s**self.gamma * np.sqrt(self.step_size) * z

    class UserShortTermAlphaMidprice(MidpriceModel):
        """A midprice WITH A SECOND STATE COLUMN: dS = alpha dt + sigma sqrt(dt) Z1, and a short-term alpha that mean-reverts, diffuses
        and jumps on the market's order flow, d alpha = -kappa alpha dt + xi sqrt(dt) Z2 + eps (sell arrivals - buy arrivals)."""

        def __init__(self, volatility, kappa, xi, eps, initial_price, lo, hi, alpha_lo, alpha_hi, terminal_time, step_size, num_trajectories, seed=None):
            self.volatility, self.kappa, self.xi, self.eps = volatility, kappa, xi, eps
            super().__init__(min_value=np.array([[lo, alpha_lo]]), max_value=np.array([[hi, alpha_hi]]), step_size=step_size, terminal_time=terminal_time,
                             initial_state=np.array([[initial_price, 0.0]]), num_trajectories=num_trajectories, seed=seed)

        def update(self, arrivals, fills, actions, state=None):
            s, a = self.current_state[:, 0:1], self.current_state[:, 1:2]
            z = self.rng.normal(size=(self.num_trajectories, 2))
            dt = self.step_size
            s_new = s + a * dt + self.volatility * np.sqrt(dt) * z[:, 0:1]
            a_new = a - self.kappa * a * dt + self.xi * np.sqrt(dt) * z[:, 1:2] + self.eps * (arrivals[:, 1:2] * 1.0 - arrivals[:, 0:1] * 1.0)
            self.current_state = np.append(s_new, a_new, axis=1)

    class UserAdaptiveFill(FillProbabilityModel):
        """A fill model WITH STATE (SP:8-53): p(depth) = exp(-kappa depth) with a decay rate kappa of its own - the agent's fills thin
        the book out behind them (kappa jumps) and it relaxes to its level: kappa <- kappa + speed (level - kappa) dt + jump trades."""

        def __init__(self, level, speed, jump, lo, hi, step_size, num_trajectories, seed=None):
            self.level, self.speed, self.jump = level, speed, jump
            super().__init__(min_value=np.array([[lo]]), max_value=np.array([[hi]]), step_size=step_size, terminal_time=0.0,
                             initial_state=np.array([[level]]), num_trajectories=num_trajectories, seed=seed)

        def _get_fill_probabilities(self, depths):
            return np.exp(-self.current_state * depths)

        @property
        def max_depth(self):
            return -np.log(0.01) / self.level

        def update(self, arrivals, fills, actions, state=None):
            k = self.current_state
            self.current_state = k + self.speed * (self.level - k) * self.step_size + self.jump * np.sum(arrivals * fills, axis=1, keepdims=True)

    class UserSquareRootImpact(PriceImpactModel):
        """The square-root law of market impact on top of a transient component that decays (IMP:9-31 asks for get_impact and
        max_speed): impact = c sign(v) sqrt(|v|) + y;  y <- y - rho y dt + k v dt.  The model OWNS the state column y."""

        def __init__(self, coefficient, resilience, kernel, max_speed, step_size, terminal_time, num_trajectories, seed=None):
            self.coefficient, self.resilience, self.kernel, self._max_speed = coefficient, resilience, kernel, max_speed
            bound = max_speed * terminal_time * kernel
            super().__init__(min_value=np.array([[-bound]]), max_value=np.array([[bound]]), step_size=step_size, terminal_time=terminal_time,
                             initial_state=np.array([[0.0]]), num_trajectories=num_trajectories, seed=seed)

        def get_impact(self, action):
            return self.coefficient * np.sign(action) * np.sqrt(np.abs(action)) + self.current_state

        def update(self, arrivals, fills, actions, state=None):
            y = self.current_state
            self.current_state = y - self.resilience * y * self.step_size + self.kernel * actions * self.step_size

        @property
        def max_speed(self):
            return self._max_speed

    return types.SimpleNamespace(UserStateReadingArrivals=UserStateReadingArrivals, UserSquareRootImpact=UserSquareRootImpact, UserAdaptiveFill=UserAdaptiveFill, UserPowerLawFill=UserPowerLawFill, UserExponentialInventoryCost=UserExponentialInventoryCost,
                                 UserSeasonalArrivals=UserSeasonalArrivals, UserCrossExcitingHawkes=UserCrossExcitingHawkes,
                                 UserCevMidprice=UserCevMidprice, UserShortTermAlphaMidprice=UserShortTermAlphaMidprice)


class Replay:
    """Stands in for a numpy Generator inside one process (what tools/refgen/make_golden.py gives the reference's processes):
    `uniform` / `normal` hand out the pre-drawn, float32-representable arrays of a fixture, step by step."""

    def __init__(self, uniforms=None, normals=None):
        self.uniforms, self.normals, self.ku, self.kn = uniforms, normals, 0, 0

    def uniform(self, size=None):
        out = self.uniforms[self.ku].astype(np.float64)
        assert out.shape == tuple(size)
        self.ku += 1
        return out

    def normal(self, size=None):
        out = self.normals[self.kn].astype(np.float64).reshape(size)
        self.kn += 1
        return out
