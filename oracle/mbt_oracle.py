"""CPU oracle for the TradingEnvironment.step() hot path.  TEST INFRASTRUCTURE - NOT PRODUCT CODE.

This module is a float64 NumPy *restatement* of the reference algorithm (JJJerome/mbt_gym, mounted
read-only at /root/reference in the build container).  It exists to check the HIP path.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; nothing
under `mbt_gym_amd/` does, and the product path never falls back to it.

Parity status: PINNED.
  * bit-for-bit (float64) against the reference itself, run in the build container under injected
    noise, for every fixture in tests/golden/*.npz (generator: tools/refgen/make_golden.py);
  * bit-level known-answer test against the reference's published Avellaneda-Stoikov table
    (notebooks/Test_1_-_replicate_AS_original_results.ipynb:219-231 and :338-350) through
    `NumpyProtocolNoise`, which follows the reference's RNG protocol exactly;
  * the reference's own unit tests (mbt_gym/rewards/tests/testRewardFunctions.py) re-expressed in
    tests/test_oracle_rewards.py.

Citation shorthand (paths relative to /root/reference/mbt_gym):
  TE   gym/TradingEnvironment.py        MD   gym/ModelDynamics.py
  MID  stochastic_processes/midprice_models.py     ARR  stochastic_processes/arrival_models.py
  FILL stochastic_processes/fill_probability_models.py   RW rewards/RewardFunctions.py
  SP   stochastic_processes/StochasticProcessModel.py    AG agents/BaselineAgents.py
  GT   gym/helpers/generate_trajectory.py

State columns (gym/index_names.py:1-4): 0 cash, 1 inventory, 2 time, 3 midprice, then the
arrival model's state (Hawkes: 4 = bid intensity, 5 = ask intensity; TE:311-318).
Side 0 = bid, side 1 = ask (index_names.py:6-7).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple, Union

import numpy as np

CASH, INVENTORY, TIME, PRICE = 0, 1, 2, 3

# ----------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------


@dataclass
class OracleConfig:
    """Flat description of one environment.  Field defaults follow the reference constructors."""

    num_trajectories: int = 1
    n_steps: int = 200  # TE:30
    terminal_time: float = 1.0  # TE:29
    # midprice: "bm" (MID:36-68) or "ou" (MID:114-146)
    midprice: str = "bm"
    drift: float = 0.0
    volatility: float = 2.0
    initial_price: float = 100.0
    ou_level: float = 0.0  # mean_reversion_level MID:117
    ou_speed: float = 1.0  # mean_reversion_speed MID:118
    # arrivals: "poisson" (ARR:32-56) or "hawkes" (ARR:86-126)
    arrival: str = "poisson"
    intensity: Sequence[float] = (140.0, 140.0)  # Poisson rate / Hawkes baseline (bid, ask)
    hawkes_jump: float = 40.0
    hawkes_speed: float = 60.0
    # fills: exponential (FILL:42-65)
    fill_exponent: float = 1.5
    # dynamics: "limit" (MD:87-131) or "limit_and_market" (MD:179-240)
    dynamics: str = "limit"
    market_half_spread: float = 0.5  # MD:189
    max_depth: Optional[float] = None  # MD:103 / FILL:60-62
    # reward: "pnl" (RW:20-36), "running" (RW:116-143), "cjmm" (RW:77-113)
    reward: str = "pnl"
    phi: float = 0.01  # per_step_inventory_aversion
    alpha: float = 0.0  # terminal_inventory_aversion
    inventory_exponent: float = 2.0
    # environment (TE:27-45)
    initial_cash: float = 0.0
    initial_inventory: Union[int, Tuple[int, int]] = 0
    max_inventory: int = 10_000
    max_cash: Optional[float] = None
    max_stock_price: Optional[float] = None
    start_time: float = 0.0
    seed: Optional[int] = None
    normalise_action_space: bool = True
    normalise_observation_space: bool = True
    reward_scaling: Optional[float] = None  # 1/mean-neutral-reward when normalise_rewards (TE:90-94)

    @property
    def step_size(self) -> float:
        return self.terminal_time / self.n_steps  # TE:49

    @property
    def state_dim(self) -> int:
        return 4 + (2 if self.arrival == "hawkes" else 0)  # TE:311-318

    @property
    def action_dim(self) -> int:
        return 4 if self.dynamics == "limit_and_market" else 2  # MD:121, MD:227-230


# ----------------------------------------------------------------------------------------------
# bounds and spaces
# ----------------------------------------------------------------------------------------------


def midprice_bounds(cfg: OracleConfig) -> Tuple[float, float]:
    """min/max of the midprice process: BM uses 4 sigma sqrt(T) (MID:48-49, MID:67-68), OU uses
    4 sigma T (MID:130-131, MID:145-146)."""
    if cfg.midprice == "bm":
        hi = cfg.initial_price + 4 * cfg.volatility * np.sqrt(cfg.terminal_time)
    elif cfg.midprice == "ou":
        hi = cfg.initial_price + 4 * cfg.volatility * cfg.terminal_time
    else:
        raise ValueError(cfg.midprice)
    lo = cfg.initial_price - (hi - cfg.initial_price)
    return float(lo), float(hi)


def resolved_max_depth(cfg: OracleConfig) -> float:
    """-ln(0.01)/kappa unless given (FILL:60-62, MD:103)."""
    return cfg.max_depth or float(-np.log(0.01) / cfg.fill_exponent)


def resolved_max_stock_price(cfg: OracleConfig) -> float:
    return cfg.max_stock_price or midprice_bounds(cfg)[1]  # TE:75


def resolved_max_cash(cfg: OracleConfig) -> float:
    return cfg.max_cash or cfg.n_steps * resolved_max_stock_price(cfg)  # TE:76, TE:229-230


def observation_bounds(cfg: OracleConfig) -> Tuple[np.ndarray, np.ndarray]:
    """float32 Box bounds of the un-normalised observation (TE:232-241)."""
    max_cash = resolved_max_cash(cfg)
    lo = [-max_cash, -cfg.max_inventory, 0.0]
    hi = [max_cash, cfg.max_inventory, cfg.terminal_time]
    mlo, mhi = midprice_bounds(cfg)
    lo.append(mlo)
    hi.append(mhi)
    if cfg.arrival == "hawkes":
        base = np.asarray(cfg.intensity, dtype=np.float64).reshape(-1)
        lo += [0.0, 0.0]  # ARR:100
        hi += list(base * 10)  # ARR:101, ARR:125-126
    return np.float32(np.array(lo)), np.float32(np.array(hi))


def action_bounds(cfg: OracleConfig) -> Tuple[np.ndarray, np.ndarray]:
    """float32 Box bounds of the un-normalised action (MD:118-121, MD:224-231)."""
    d = resolved_max_depth(cfg)
    if cfg.dynamics == "limit":
        return np.zeros(2, np.float32), np.full(2, np.float32(d), np.float32)
    if cfg.dynamics == "limit_and_market":
        return np.zeros(4, np.float32), np.array([d, d, 1, 1], dtype=np.float32)
    raise ValueError(cfg.dynamics)


# ----------------------------------------------------------------------------------------------
# noise sources
# ----------------------------------------------------------------------------------------------


class InjectedNoise:
    """Replays pre-drawn noise: u_arr (K,N,2), u_fill (K,N,2), z (K,N)."""

    def __init__(self, u_arr, u_fill, z):
        self.u_arr, self.u_fill, self.z = (np.asarray(a, dtype=np.float64) for a in (u_arr, u_fill, z))
        self.k = 0

    def draw(self, n: int):
        k = self.k
        self.k += 1
        return self.u_arr[k], self.u_fill[k], self.z[k].reshape(n, 1)


class NumpyProtocolNoise:
    """The reference's RNG protocol: process i (registry order midprice, arrival, fill; TE:303-309)
    owns default_rng(seed + i + 1) (TE:345-348, SP:37-39); per step one uniform (N,2) from the
    arrival generator (ARR:55), one uniform (N,2) from the fill generator (FILL:33), one
    normal (N,1) from the midprice generator (MID:64, MID:143).  reset() does not reseed."""

    def __init__(self, seed: int):
        self.mid = np.random.default_rng(seed + 1)
        self.arr = np.random.default_rng(seed + 2)
        self.fill = np.random.default_rng(seed + 3)

    def draw(self, n: int):
        u_arr = self.arr.uniform(size=(n, 2))
        u_fill = self.fill.uniform(size=(n, 2))
        z = self.mid.normal(size=(n, 1))
        return u_arr, u_fill, z


# ----------------------------------------------------------------------------------------------
# the environment
# ----------------------------------------------------------------------------------------------


class OracleEnv:
    """float64 restatement of TradingEnvironment (TE:24-348) for the starred plugin classes."""

    def __init__(self, cfg: OracleConfig, noise=None):
        self.cfg = cfg
        self.noise = noise
        n = cfg.num_trajectories
        self.dt = cfg.step_size
        self.obs_lo, self.obs_hi = observation_bounds(cfg)
        self.act_lo, self.act_hi = action_bounds(cfg)
        self.max_cash = resolved_max_cash(cfg)
        self.env_rng = np.random.default_rng(cfg.seed)  # TE:72
        self.bid_ask_sign = np.append(-np.ones((n, 1)), np.ones((n, 1)), axis=1)  # MD:71-73
        self.state = self._initial_state()  # TE:74 (consumes one env-rng draw for tuple inventories)
        self.q_init = None
        self.episode_length = None
        self.last_arrivals = None
        self.last_fills = None

    # -- reset ---------------------------------------------------------------------------------
    def _start_time(self) -> float:
        t = self.cfg.start_time
        assert 0.0 <= t < self.cfg.terminal_time  # TE:267
        return np.round(t / self.dt) * self.dt  # TE:266-268

    def _initial_inventories(self) -> np.ndarray:
        q0 = self.cfg.initial_inventory
        n = self.cfg.num_trajectories
        if isinstance(q0, tuple) and len(q0) == 2:
            return self.env_rng.integers(*q0, size=n)  # TE:271-272
        return q0 * np.ones((n,))  # TE:273-274

    def _initial_state(self) -> np.ndarray:
        """TE:131-140 with SP:48-53 for the process columns."""
        cfg = self.cfg
        n = cfg.num_trajectories
        s = np.repeat(np.array([[cfg.initial_cash, 0, 0.0]]), n, axis=0)
        s[:, TIME] = self._start_time() * np.ones((n,))
        s[:, INVENTORY] = self._initial_inventories()
        cols = [np.repeat(np.array([[cfg.initial_price]], dtype=np.float64), n, axis=0)]
        if cfg.arrival == "hawkes":
            cols.append(np.repeat(np.asarray(cfg.intensity, dtype=np.float64).reshape(1, 2), n, axis=0))
        for c in cols:
            s = np.append(s, c, axis=1)
        return s

    def reset(self) -> np.ndarray:
        """TE:96-101; reward reset RW:111-113."""
        self.state = self._initial_state()
        self.q_init = self.state[:, INVENTORY].copy()
        self.episode_length = self.cfg.terminal_time - self.state[:, TIME]
        return self.normalise_observation(self.state.copy())

    # -- normalisation (TE:112-129, TE:180-194) -------------------------------------------------
    def normalise_observation(self, obs: np.ndarray) -> np.ndarray:
        if not self.cfg.normalise_observation_space:
            return obs
        grad = (self.obs_hi - self.obs_lo) / 2  # float32 arithmetic, as in TE:185
        return (obs - self.obs_lo) / grad - 1

    def denormalise_action(self, action: np.ndarray) -> np.ndarray:
        if not self.cfg.normalise_action_space:
            return action
        grad = (self.act_hi - self.act_lo) / 2  # TE:193
        return (action + 1) * grad + self.act_lo  # TE:124

    # -- one step (TE:103-110) ------------------------------------------------------------------
    def step(self, action: np.ndarray):
        cfg = self.cfg
        n = cfg.num_trajectories
        dt = self.dt
        action = self.denormalise_action(np.asarray(action))
        prev = self.state.copy()  # TE:105
        st = self.state
        u_arr, u_fill, z = self.noise.draw(n)

        # arrivals (ARR:54-56 / ARR:121-123) and raw fills (FILL:28-34, FILL:57-58)
        if cfg.arrival == "poisson":
            arrivals = u_arr < np.array(cfg.intensity) * dt
        else:
            arrivals = u_arr < st[:, 4:6] * dt
        depths = action[:, 0:2]  # MD:50-51
        fills = u_fill < np.exp(-cfg.fill_exponent * depths)
        # no bid fill at +max inventory, no ask fill at -max inventory, pre-update q (TE:323-327)
        at_max = st[:, INVENTORY] >= cfg.max_inventory
        at_min = st[:, INVENTORY] <= -cfg.max_inventory
        mask = np.concatenate(((1 - at_max).reshape(-1, 1), (1 - at_min).reshape(-1, 1)), axis=1)
        fills = mask * fills

        # cash / inventory (MD:108-116, MD:208-222); midprice is the OLD one (MD:82-84)
        mid = st[:, PRICE].reshape(-1, 1)
        sgn = self.bid_ask_sign
        if cfg.dynamics == "limit_and_market":
            mo_buy = np.single(action[:, 2] > 0.5)
            mo_sell = np.single(action[:, 3] > 0.5)
            best_bid = (mid - cfg.market_half_spread).reshape(-1)
            best_ask = (mid + cfg.market_half_spread).reshape(-1)
            st[:, CASH] += mo_sell * best_bid - mo_buy * best_ask
            st[:, INVENTORY] += mo_buy - mo_sell
        st[:, INVENTORY] += np.sum(arrivals * fills * -sgn, axis=1)
        st[:, CASH] += np.sum(sgn * arrivals * fills * (mid + depths * sgn), axis=1)
        # clip (TE:283-289) and advance time (TE:216)
        st[:, INVENTORY] = np.clip(st[:, INVENTORY], -cfg.max_inventory, cfg.max_inventory)
        st[:, CASH] = np.clip(st[:, CASH], -self.max_cash, self.max_cash)
        st[:, TIME] += dt

        # processes in registry order midprice, arrival (TE:206-211, TE:303-309)
        s_old = prev[:, PRICE].reshape(-1, 1)
        if cfg.midprice == "bm":  # MID:60-65
            s_new = s_old + cfg.drift * dt * np.ones((n, 1)) + cfg.volatility * math.sqrt(dt) * z
        else:  # MID:140-143 (mean reversion is NOT scaled by dt in the reference)
            s_new = s_old + (
                -cfg.ou_speed * (s_old - cfg.ou_level * np.ones((n, 1))) + cfg.volatility * math.sqrt(dt) * z
            )
        st[:, PRICE] = s_new[:, 0]
        if cfg.arrival == "hawkes":  # ARR:110-119 (jumps on arrivals, not on fills)
            lam = prev[:, 4:6]
            base = np.asarray(cfg.intensity, dtype=np.float64).reshape(1, 2)
            st[:, 4:6] = (
                lam + cfg.hawkes_speed * (np.ones((n, 2)) * base - lam) * dt * np.ones((n, 2)) + cfg.hawkes_jump * arrivals
            )

        done = bool(st[0, TIME] >= cfg.terminal_time - dt / 2)  # TE:218-220
        rewards = self._reward(prev, st, done)
        if cfg.reward_scaling is not None:
            rewards = cfg.reward_scaling * rewards  # TE:128-129
        self.last_arrivals = arrivals
        self.last_fills = fills
        dones = np.full((n,), done, dtype=bool)
        return self.normalise_observation(st.copy()), rewards, dones

    # -- rewards -------------------------------------------------------------------------------
    def _reward(self, cur: np.ndarray, nxt: np.ndarray, done: bool) -> np.ndarray:
        cfg = self.cfg
        pnl = pnl_reward(cur, nxt)
        if cfg.reward == "pnl":
            return pnl
        if cfg.reward == "running":
            return running_inventory_penalty(cur, nxt, done, cfg.phi, cfg.alpha, cfg.inventory_exponent)
        if cfg.reward == "cjmm":
            return cj_mm_criterion(
                cur, nxt, cfg.phi, cfg.alpha, cfg.inventory_exponent, self.q_init, self.episode_length
            )
        raise ValueError(cfg.reward)


def pnl_reward(cur: np.ndarray, nxt: np.ndarray) -> np.ndarray:
    """Mark-to-market change (RW:23-33)."""
    return (nxt[:, CASH] + nxt[:, INVENTORY] * nxt[:, PRICE]) - (cur[:, CASH] + cur[:, INVENTORY] * cur[:, PRICE])


def running_inventory_penalty(cur, nxt, done: bool, phi: float, alpha: float, p: float) -> np.ndarray:
    """RW:128-138."""
    dt = nxt[:, TIME] - cur[:, TIME]
    return pnl_reward(cur, nxt) - dt * phi * nxt[:, INVENTORY] ** p - alpha * int(done) * nxt[:, INVENTORY] ** p


def cj_mm_criterion(cur, nxt, phi: float, alpha: float, p: float, q_init, episode_length) -> np.ndarray:
    """RW:96-109 with the reset() captures of RW:111-113."""
    dt = nxt[:, TIME] - cur[:, TIME]
    return (
        pnl_reward(cur, nxt)
        - dt * phi * nxt[:, INVENTORY] ** p
        - alpha * (nxt[:, INVENTORY] ** p - cur[:, INVENTORY] ** p + dt / episode_length * q_init**p)
    )


# ----------------------------------------------------------------------------------------------
# callers on the path: closed-form policies and the rollout loop
# ----------------------------------------------------------------------------------------------


def avellaneda_stoikov_action(cfg: OracleConfig, gamma: float, state: np.ndarray) -> np.ndarray:
    """Closed-form AS half-spreads (AG:70-83); `state` is the un-normalised observation."""
    q = state[:, INVENTORY]
    t = state[:, TIME]
    adj = q * gamma * cfg.volatility**2 * (cfg.terminal_time - t)
    if gamma == 0:
        spread = 2 / cfg.fill_exponent
    else:
        spread = gamma * cfg.volatility**2 * (cfg.terminal_time - t) + 2 / gamma * np.log(1 + gamma / cfg.fill_exponent)
    bid = (adj + spread / 2).reshape(-1, 1)
    ask = (-adj + spread / 2).reshape(-1, 1)
    return np.append(bid, ask, axis=1)


def fixed_spread_action(n: int, half_spread: float = 1.0, offset: float = 0.0) -> np.ndarray:
    """AG:34-42."""
    return np.repeat(np.array([[half_spread - offset, half_spread + offset]]), n, axis=0)


def rollout(env: OracleEnv, policy):
    """The canonical rollout loop and its output layout (GT:8-38): observations (N, D, n_steps+1),
    actions (N, A, n_steps), rewards (N, 1, n_steps)."""
    cfg = env.cfg
    n = cfg.num_trajectories
    obs_t = np.zeros((n, cfg.state_dim, cfg.n_steps + 1))
    act_t = np.zeros((n, cfg.action_dim, cfg.n_steps))
    rew_t = np.zeros((n, 1, cfg.n_steps))
    obs = env.reset()
    obs_t[:, :, 0] = obs
    k = 0
    while True:
        a = policy(obs)
        obs, r, done = env.step(a)
        act_t[:, :, k] = a
        obs_t[:, :, k + 1] = obs
        rew_t[:, :, k] = r.reshape(-1, 1)
        if done[0]:
            break
        k += 1
    return obs_t, act_t, rew_t


def results_table(obs_t, act_t, rew_t):
    """The published statistics (gym/helpers/plotting.py:96-108): mean spread, mean/std of total
    reward, mean/std of terminal inventory."""
    total = rew_t.sum(axis=-1).reshape(-1)
    q_t = obs_t[:, INVENTORY, -1]
    return (
        2 * np.mean(act_t.mean(axis=(-1, -2))),
        np.mean(total),
        np.std(total),
        np.mean(q_t),
        np.std(q_t),
    )
