"""Seeded random draws from the plugin space (kinds x parameters) for the differential tests: the HIP environment against
the oracle (tests/test_gpu_random_configs.py, GPU) and the oracle against the reference itself where the reference is
present (tests/test_oracle_vs_reference_live.py, build container only)."""
import numpy as np

from oracle.mbt_oracle import OracleConfig, action_bounds


def random_config(rng, n):
    dynamics = rng.choice(["limit", "limit", "limit_and_market", "touch"])
    arrival = rng.choice(["poisson", "poisson_nonlinear", "hawkes"])
    midprice = rng.choice(["bm", "ou", "gbm", "bm_jump", "ou_jump", "constant"])
    reward = rng.choice(["pnl", "running", "cjmm", "exp_utility"])
    n_steps = int(rng.integers(20, 60))
    normalised = bool(rng.integers(0, 2)) and dynamics != "touch"
    s0 = float(rng.choice([10.0, 50.0, 100.0]))
    cfg = OracleConfig(
        num_trajectories=n, n_steps=n_steps, terminal_time=float(rng.choice([0.5, 1.0, 2.0])), midprice=midprice,
        drift=float(rng.uniform(-0.2, 0.2)), volatility=float(rng.uniform(0.05, 0.3) if midprice == "gbm" else rng.uniform(0.5, 3.0)),
        initial_price=s0, ou_level=s0 + float(rng.uniform(-1, 1)), ou_speed=float(rng.uniform(0.0, 0.1)), jump_size=float(rng.uniform(0.0, 0.3)),
        arrival=arrival, intensity=(float(rng.uniform(5, 60)), float(rng.uniform(5, 60))), hawkes_jump=float(rng.uniform(5, 40)),
        hawkes_speed=float(rng.uniform(20, 80)), fill_exponent=float(rng.uniform(0.5, 3.0)), dynamics=dynamics,
        market_half_spread=float(rng.uniform(0.1, 0.6)), reward=reward, risk_aversion=float(rng.uniform(0.001, 0.05)),
        phi=float(rng.uniform(0.0, 0.05)), alpha=float(rng.uniform(0.0, 0.1)), inventory_exponent=2.0,
        initial_inventory=int(rng.integers(-2, 3)), max_inventory=int(rng.choice([2, 5, 50])), seed=int(rng.integers(1, 2**31)),
        normalise_action_space=normalised, normalise_observation_space=normalised,
    )
    # the explicit Euler recursion of the Hawkes intensity (ARR:110-119) is a contraction only for speed * dt < 1; beyond
    # that it oscillates (>= 2: diverges, in the float64 reference too) and float32 state stops tracking float64 state.
    # That is the documented domain of the device (include/mbt_env.h: allow_stiff_hawkes) - mbt_env_create REFUSES a
    # configuration outside it (tests/test_gpu_round2.py) - so the random draw stays inside
    cfg.hawkes_speed = min(cfg.hawkes_speed, 0.9 / cfg.step_size)
    if rng.integers(0, 4) == 0:
        cfg.start_time = cfg.terminal_time * 0.25
    if dynamics != "touch" and rng.integers(0, 5) == 0:
        # ExogenousMmFillProbabilityModel (FILL:126-170): fills are certain inside an exogenous best depth, exponential
        # beyond it; two more state columns (which the reference never advances).  float32-representable best depths, so
        # that a quote can sit exactly on one
        best = np.float32(rng.uniform(0.05, 0.6, size=2)).astype(np.float64)
        cfg.fill, cfg.exo_depth, cfg.base_fill_probability = "exogenous", (float(best[0]), float(best[1])), float(rng.uniform(0.3, 1.0))
        cfg.exo_depth_lo = (float(best[0] - rng.uniform(0.1, 0.5)), float(best[1] - rng.uniform(0.1, 0.5)))
        cfg.exo_depth_hi = (float(best[0] + rng.uniform(0.1, 0.5)), float(best[1] + rng.uniform(0.1, 0.5)))
    return cfg


def random_actions(rng, cfg, steps):
    lo, hi = action_bounds(cfg)
    n, a = cfg.num_trajectories, cfg.action_dim
    if cfg.dynamics == "touch":
        return rng.integers(0, 2, size=(steps, n, 2)).astype(np.float32)
    if cfg.normalise_action_space:
        act = rng.uniform(-1, 1, size=(steps, n, a))
        if a == 4:  # market orders: mostly clearly off / clearly on
            act[:, :, 2:] = rng.choice([-1.0, 1.0, 0.3], p=[0.85, 0.1, 0.05], size=(steps, n, 2))
        return act.astype(np.float32)
    act = rng.uniform(0, 0.8, size=(steps, n, a)) * hi
    if a == 4:
        act[:, :, 2:] = rng.choice([0.0, 1.0, 0.4], p=[0.85, 0.1, 0.05], size=(steps, n, 2))
    return act.astype(np.float32)


def random_speed_config(rng, n):
    impact = rng.choice(["temp_power", "temp_perm", "temp_transient", "transient"])
    n_steps = int(rng.integers(20, 60))
    T = float(rng.choice([0.5, 1.0, 2.0]))
    normalised = bool(rng.integers(0, 2))
    cfg = OracleConfig(
        num_trajectories=n, n_steps=n_steps, terminal_time=T, midprice=rng.choice(["bm", "ou", "gbm", "constant"]),
        drift=float(rng.uniform(-0.1, 0.1)), volatility=float(rng.uniform(0.05, 0.3)), initial_price=float(rng.choice([20.0, 100.0])),
        ou_level=100.0, ou_speed=float(rng.uniform(0.0, 0.05)), arrival="none", dynamics="speed", impact=impact,
        temporary_impact=float(rng.uniform(0.005, 0.05)), impact_exponent=float(rng.choice([1.0, 1.0, 1.5, 0.6])),
        permanent_impact=float(rng.uniform(0.0, 0.03)), transient_impact=float(rng.uniform(0.1, 0.6)), resilience=float(rng.uniform(0.5, 3.0)),
        initial_transient_impact=float(rng.uniform(0.0, 0.2)), kernel_coefficient=float(rng.uniform(0.0, 0.4)), impact_step_size=T / n_steps,
        reward=rng.choice(["pnl", "running", "cjoe"]), phi=float(rng.uniform(0.0, 0.05)), alpha=float(rng.uniform(0.0, 0.2)),
        initial_inventory=int(rng.integers(1, 20)), max_inventory=int(rng.choice([15, 1000])), seed=int(rng.integers(1, 2**31)),
        normalise_action_space=normalised, normalise_observation_space=normalised,
    )
    if rng.integers(0, 3) == 0:  # MD:265 trades the MIDPRICE model's step size, which need not be the environment's
        cfg.midprice_step_size = float(rng.choice([0.5, 2.0])) * cfg.step_size
    return cfg


def random_speed_actions(rng, cfg, steps):
    lo, hi = action_bounds(cfg)
    n = cfg.num_trajectories
    positive = cfg.impact == "temp_power" and cfg.impact_exponent != 1.0  # v ** e with a fractional exponent needs v >= 0
    if cfg.normalise_action_space:
        return rng.uniform(0.0 if positive else -0.4, 0.4, size=(steps, n, 1)).astype(np.float32)
    return (rng.uniform(0.0 if positive else -0.4, 0.4, size=(steps, n, 1)) * hi).astype(np.float32)


NUMPY_ONLY_KINDS = ("reward_speed", "midprice_speed", "impact_speed", "adaptive_fill", "state_reading_arrivals")


def random_numpy_only_config(rng, n, kind):
    """A random market around one of the NumPy-only user classes of tests/numpy_only_plugins.py (tests/env_factory.py builds it from
    the class itself, on either package): a user reward / midprice / price impact model with trading-with-speed dynamics, or the
    fill model that owns a state column with order-book dynamics."""
    if kind == "state_reading_arrivals":
        cfg = random_config(rng, n)
        cfg.dynamics, cfg.fill = str(rng.choice(["limit", "limit_and_market"])), "exponential"
        cfg.arrival, cfg.hawkes_speed = "user_state_reading", min(float(rng.uniform(5.0, 40.0)), 0.9 / cfg.step_size)
        cfg.arrival_tilt, cfg.arrival_sensitivity, cfg.arrival_crowding = float(rng.uniform(-0.5, 1.0)), float(rng.uniform(0.0, 4.0)), float(rng.uniform(0.0, 1.0))
        cfg.arrival_reference_price = cfg.initial_price
        return cfg
    if kind == "adaptive_fill":
        cfg = random_config(rng, n)
        cfg.dynamics = str(rng.choice(["limit", "limit_and_market"]))
        cfg.arrival = str(rng.choice(["poisson", "poisson_nonlinear"]))  # (two more columns of a Hawkes model + this one: beyond the device's rows)
        cfg.fill, cfg.fill_kappa_speed, cfg.fill_kappa_jump, cfg.fill_kappa_lo, cfg.fill_kappa_hi = "user_adaptive", float(rng.uniform(0.0, 8.0)), float(rng.uniform(0.0, 1.0)), 0.1, 20.0
        cfg.normalise_action_space = cfg.normalise_observation_space = bool(rng.integers(0, 2))
        return cfg
    cfg = random_speed_config(rng, n)
    if kind == "reward_speed":
        cfg.reward, cfg.eta = "user_exp_inventory_cost", float(rng.uniform(0.02, 0.25))
    elif kind == "midprice_speed":
        cfg.midprice, cfg.initial_price, cfg.cev_gamma = "user_cev", float(rng.choice([50.0, 100.0])), float(rng.uniform(0.5, 1.0))
        cfg.volatility, cfg.midprice_lo, cfg.midprice_hi = float(rng.uniform(0.05, 0.3)), 0.4 * cfg.initial_price, 1.6 * cfg.initial_price
    else:
        cfg.impact, cfg.initial_transient_impact = "user_sqrt", 0.0
    return cfg


def random_numpy_only_actions(rng, cfg, steps):
    return random_speed_actions(rng, cfg, steps) if cfg.dynamics == "speed" else random_actions(rng, cfg, steps)
