#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (read-only at /root/reference).

Build-container only: /root/reference does not exist on the GPU box, and nothing in tests/,
bench.py or the package reads it at run time.  The reference's `import gym` is satisfied by the
numerics-free stand-in in tools/refgen/gym_standin (gym is not installed and there is no network).
Only DATA is written: inputs (actions, noise, initial inventories) and the reference's outputs.

Noise injection: each reference process only ever calls `rng.uniform(size=...)` /
`rng.normal(size=...)` (arrival_models.py:55,122; fill_probability_models.py:33;
midprice_models.py:64,143), so replacing `process.rng` by a replay object feeds it pre-drawn,
float32-representable noise.  Some draws are placed exactly on / next to the decision thresholds to
pin the strict `<` comparisons in float64.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/refgen/make_golden.py
"""
import contextlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "gym_standin"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

from mbt_gym.gym.ModelDynamics import (  # noqa: E402
    AtTheTouchModelDynamics,
    LimitAndMarketOrderModelDynamics,
    LimitOrderModelDynamics,
    TradinghWithSpeedModelDynamics,
)
from mbt_gym.gym.TradingEnvironment import TradingEnvironment  # noqa: E402
from mbt_gym.rewards.RewardFunctions import (  # noqa: E402
    CjMmCriterion,
    CjOeCriterion,
    ExponentialUtility,
    PnL,
    RunningInventoryPenalty,
)
from mbt_gym.stochastic_processes.arrival_models import (  # noqa: E402
    HawkesArrivalModel,
    PoissonArrivalModel,
    PoissonArrivalNonLinearModel,
)
from mbt_gym.stochastic_processes.fill_probability_models import (  # noqa: E402
    ExogenousMmFillProbabilityModel,
    ExponentialFillFunction,
)
from mbt_gym.stochastic_processes.midprice_models import (  # noqa: E402
    BrownianMotionJumpMidpriceModel,
    BrownianMotionMidpriceModel,
    ConstantMidpriceModel,
    GeometricBrownianMotionMidpriceModel,
    OuJumpMidpriceModel,
    OuMidpriceModel,
)
from mbt_gym.stochastic_processes.price_impact_models import (  # noqa: E402
    TemporaryAndPermanentPriceImpact,
    TemporaryAndTransientPriceImpact,
    TemporaryPowerPriceImpact,
    TransientPriceImpact,
)

OUT = os.path.join(REPO, "tests", "golden")


class Replay:
    """Stands in for a numpy Generator inside one reference process."""

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


def f32_neighbours(x):
    """float32 values just below / nearest / just above a float64 array."""
    c = np.float32(x)
    return np.nextafter(c, np.float32(-np.inf)), c, np.nextafter(c, np.float32(np.inf))


def draw_noise(rng, k, n):
    u_arr = (rng.integers(0, 1 << 24, size=(k, n, 2)) / float(1 << 24)).astype(np.float32)
    u_fill = (rng.integers(0, 1 << 24, size=(k, n, 2)) / float(1 << 24)).astype(np.float32)
    z = rng.normal(size=(k, n)).astype(np.float32)
    return u_arr, u_fill, z


def draw_actions(rng, k, n, a_dim, max_depth, normalised, kind="depths", max_speed=10.0):
    if kind == "touch":  # MultiBinary(2): post at the touch or not
        return rng.integers(0, 2, size=(k, n, 2)).astype(np.float32)
    if kind == "speed":
        return rng.uniform(-0.4 * max_speed, 0.4 * max_speed, size=(k, n, 1)).astype(np.float32)
    if kind == "speed_positive":
        return rng.uniform(0.0, 0.05 * max_speed, size=(k, n, 1)).astype(np.float32)
    if normalised:
        act = rng.uniform(-1, 1, size=(k, n, a_dim))
    else:
        act = rng.uniform(0, 0.8 * max_depth, size=(k, n, a_dim))
        act[rng.uniform(size=act.shape) < 0.02] = 0.0
        act[rng.uniform(size=act.shape) < 0.02] = max_depth
        if a_dim == 4:
            mo = rng.choice([0.0, 1.0, 0.5, 0.50000006, 0.3], p=[0.86, 0.1, 0.015, 0.015, 0.01], size=(k, n, 2))
            act[:, :, 2:4] = mo
    return act.astype(np.float32)


def run_case(name, build_env, k_steps, n, a_dim, seed, cfg, normalised=False, kappa=1.5, poisson_thr=None, action_kind="depths",
             fill_prob=None, action_hook=None, step_size_changes=None, user_normals=False):
    """step_size_changes: {step index: new step size} - `env.step_size = value` (TE:158-167) before that step.
    user_normals: the midprice model draws TWO normals per lane and step in one call, rng.normal(size=(N, 2)): column 0 is the
    fixture's `z`, column 1 is saved as `z_user[..., 0]` (the extra normals of user processes; `z_user[..., 1]` is not consumed)."""
    if ONLY is not None and name not in ONLY:
        return
    rng = np.random.default_rng(1000 + seed)
    with contextlib.redirect_stdout(io.StringIO()):
        env = build_env()
    md = env.model_dynamics
    max_depth = getattr(md, "max_depth", None)
    u_arr, u_fill, z = draw_noise(rng, k_steps, n)
    actions = draw_actions(rng, k_steps, n, a_dim, max_depth, normalised, action_kind, getattr(md, "max_speed", 10.0))
    # threshold-adjacent draws (strict '<' in float64): arrivals on lanes 0..2, fills on lanes 3..5
    if poisson_thr is not None:
        lo, c, hi = f32_neighbours(np.float64(poisson_thr))
        for j, v in enumerate((lo, c, hi)):
            u_arr[::3, j % n, :] = v
    if action_hook is not None:
        action_hook(actions)
    if not normalised and action_kind == "depths":
        depths64 = actions[:, :, 0:2].astype(np.float64)
        p = np.exp(-kappa * depths64) if fill_prob is None else fill_prob(depths64)
        lo, c, hi = f32_neighbours(p)
        for j, v in enumerate((lo, c, hi)):
            u_fill[1::4, (3 + j) % n, :] = v[1::4, (3 + j) % n, :]
    z_user = None
    if user_normals:
        z_user = rng.normal(size=(k_steps, n, 2)).astype(np.float32)
        md.midprice_model.rng = Replay(normals=np.stack([z, z_user[:, :, 0]], axis=-1))
    else:
        md.midprice_model.rng = Replay(normals=z)
    if md.arrival_model is not None:
        md.arrival_model.rng = Replay(uniforms=u_arr)
    if md.fill_probability_model is not None:
        md.fill_probability_model.rng = Replay(uniforms=u_fill)

    rec = {"arr": [], "fill": []}
    orig_af = md.get_arrivals_and_fills
    orig_mask = env._remove_max_inventory_fills

    def spy_af(action):
        a, f = orig_af(action)
        rec["arr"].append(np.zeros((n, 2), np.uint8) if a is None else np.array(a, dtype=np.uint8))
        if f is None:
            rec["fill"].append(np.zeros((n, 2), np.uint8))
        return a, f

    def spy_mask(fills):
        out = orig_mask(fills)
        rec["fill"].append(np.array(out, dtype=np.uint8))
        return out

    md.get_arrivals_and_fills = spy_af
    env._remove_max_inventory_fills = spy_mask

    with contextlib.redirect_stdout(io.StringIO()):
        obs0 = env.reset()
        q0 = env.model_dynamics.state[:, 1].copy()
        t0 = float(env.model_dynamics.state[0, 2])
        obs, rew, done = [], [], []
        for k in range(k_steps):
            if step_size_changes and k in step_size_changes:
                env.step_size = step_size_changes[k]
            o, r, d, _ = env.step(actions[k].astype(np.float64))
            obs.append(np.array(o, dtype=np.float64))
            rew.append(np.array(np.broadcast_to(np.asarray(r, dtype=np.float64), (n,))))  # ExponentialUtility returns the scalar 0
            done.append(bool(d[0]))
    assert done[-1] and not any(done[:-1]), (name, done)
    lo = env.original_observation_space.low if normalised else env.observation_space.low
    hi = env.original_observation_space.high if normalised else env.observation_space.high
    aspace = env.original_action_space if env.normalise_action_space_ else env.action_space
    alo = aspace.low if hasattr(aspace, "low") else np.zeros(2)  # MultiBinary(2) has no bounds
    ahi = aspace.high if hasattr(aspace, "high") else np.ones(2)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        u_arr=u_arr, u_fill=u_fill, z=z, actions=actions, q0=q0, t0=t0,
        obs0=np.array(obs0, dtype=np.float64), obs=np.stack(obs), rewards=np.stack(rew), done=np.array(done),
        arrivals=np.stack(rec["arr"]), fills=np.stack(rec["fill"]),
        obs_lo=lo, obs_hi=hi, act_lo=np.float32(alo), act_hi=np.float32(ahi), max_cash=float(env.max_cash),
        process_indices=np.array([v for v in env.stochastic_process_indices.values()]),
        config_json=json.dumps(dict(cfg, num_trajectories=n)),
        **({"z_user": z_user} if z_user is not None else {}),
        **({"step_size_at": np.array(sorted(step_size_changes)), "step_size_to": np.array([step_size_changes[k] for k in sorted(step_size_changes)])}
           if step_size_changes else {}),
    )
    fills_total = int(np.sum(np.stack(rec["arr"]) * np.stack(rec["fill"])))
    print(f"{name}: N={n} steps={k_steps} D={obs[0].shape[1]} trades={fills_total} "
          f"clip_q={int(np.sum(np.abs(np.stack(obs)[:, :, 1]) >= env.max_inventory)) if not normalised else '-'} "
          f"reward range [{np.min(rew):.4g}, {np.max(rew):.4g}]")


def lo_dynamics(n, dt, T, mid, arr, kappa=1.5, cls=LimitOrderModelDynamics, **kw):
    fill = ExponentialFillFunction(fill_exponent=kappa, step_size=dt, num_trajectories=n)
    return cls(midprice_model=mid, arrival_model=arr, fill_probability_model=fill, num_trajectories=n, **kw)


def main():
    # A. Avellaneda-Stoikov configuration of notebooks/Test_1 (BASELINE config 0), noise injected
    n, ns = 48, 200
    run_case(
        "as_limit_pnl",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=50, initial_inventory=0, max_inventory=ns,
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                BrownianMotionMidpriceModel(initial_price=100, volatility=2.0, step_size=1 / ns, terminal_time=1.0, num_trajectories=n),
                PoissonArrivalModel(intensity=np.array([140, 140]), step_size=1 / ns, num_trajectories=n)),
            normalise_action_space=False, normalise_observation_space=False, num_trajectories=n),
        ns, n, 2, 50,
        dict(n_steps=ns, terminal_time=1.0, midprice='bm', volatility=2.0, initial_price=100.0, arrival='poisson',
             intensity=[140.0, 140.0], fill_exponent=1.5, dynamics='limit', reward='pnl', initial_inventory=0,
             max_inventory=ns, seed=50, normalise_action_space=False, normalise_observation_space=False),
        poisson_thr=140 * (1 / ns))

    # B. CJP-2015 style rewards, tight inventory limit, random integer initial inventories
    n, ns = 40, 100
    for tag, rew, t0 in (("cjp_running", RunningInventoryPenalty(0.01, 0.001), 0.0),
                         ("cjp_cjmm", CjMmCriterion(0.01, 0.001, terminal_time=1.0), 0.25)):
        steps = ns - int(round(t0 * ns))
        run_case(
            tag,
            lambda: TradingEnvironment(
                terminal_time=1.0, n_steps=ns, seed=410, initial_inventory=(-3, 4), max_inventory=3, start_time=t0,
                reward_function=rew,
                model_dynamics=lo_dynamics(
                    n, 1 / ns, 1.0,
                    BrownianMotionMidpriceModel(drift=0.05, initial_price=100, volatility=2.0, step_size=1 / ns, terminal_time=1.0, num_trajectories=n),
                    PoissonArrivalModel(intensity=np.array([140, 90]), step_size=1 / ns, num_trajectories=n)),
                normalise_action_space=False, normalise_observation_space=False, num_trajectories=n),
            steps, n, 2, 410 + len(tag),
            dict(n_steps=ns, terminal_time=1.0, midprice='bm', drift=0.05, volatility=2.0, initial_price=100.0,
                 arrival='poisson', intensity=[140.0, 90.0], fill_exponent=1.5, dynamics='limit',
                 reward='running' if tag == 'cjp_running' else 'cjmm', phi=0.01, alpha=0.001, inventory_exponent=2.0,
                 initial_inventory=[-3, 4], max_inventory=3, start_time=t0, seed=410,
                 normalise_action_space=False, normalise_observation_space=False),
            poisson_thr=90 * (1 / ns))

    # C. Hawkes arrivals + OU midprice (D = 6)
    n, ns = 40, 150
    run_case(
        "hawkes_ou",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=7, initial_inventory=0, max_inventory=50,
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                OuMidpriceModel(mean_reversion_level=100.0, mean_reversion_speed=0.02, volatility=2.0, initial_price=100.0, step_size=1 / ns, terminal_time=1.0, num_trajectories=n),
                HawkesArrivalModel(baseline_arrival_rate=np.array([[10.0, 10.0]]), step_size=1 / ns, jump_size=40.0, mean_reversion_speed=60.0, terminal_time=1.0, num_trajectories=n)),
            normalise_action_space=False, normalise_observation_space=False, num_trajectories=n),
        ns, n, 2, 7,
        dict(n_steps=ns, terminal_time=1.0, midprice='ou', ou_level=100.0, ou_speed=0.02, volatility=2.0,
             initial_price=100.0, arrival='hawkes', intensity=[10.0, 10.0], hawkes_jump=40.0, hawkes_speed=60.0,
             fill_exponent=1.5, dynamics='limit', reward='pnl', initial_inventory=0, max_inventory=50, seed=7,
             normalise_action_space=False, normalise_observation_space=False))

    # D. limit + market orders (A = 4), inventory clip after market orders, terminal penalty
    n, ns = 40, 120
    run_case(
        "limit_and_market",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=11, initial_inventory=10, max_inventory=12,
            reward_function=RunningInventoryPenalty(0.01, 0.5),
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                BrownianMotionMidpriceModel(initial_price=100, volatility=2.0, step_size=1 / ns, terminal_time=1.0, num_trajectories=n),
                PoissonArrivalModel(intensity=np.array([100, 100]), step_size=1 / ns, num_trajectories=n),
                cls=LimitAndMarketOrderModelDynamics, fixed_market_half_spread=0.5),
            normalise_action_space=False, normalise_observation_space=False, num_trajectories=n),
        ns, n, 4, 11,
        dict(n_steps=ns, terminal_time=1.0, midprice='bm', volatility=2.0, initial_price=100.0, arrival='poisson',
             intensity=[100.0, 100.0], fill_exponent=1.5, dynamics='limit_and_market', market_half_spread=0.5,
             reward='running', phi=0.01, alpha=0.5, inventory_exponent=2.0, initial_inventory=10, max_inventory=12,
             seed=11, normalise_action_space=False, normalise_observation_space=False),
        poisson_thr=100 * (1 / ns))

    # E. the reference's DEFAULT environment: normalised observations and actions (TE:42-63)
    n, ns = 24, 200
    run_case(
        "default_normalised",
        lambda: TradingEnvironment(num_trajectories=n, seed=3),
        ns, n, 2, 3,
        dict(n_steps=ns, terminal_time=1.0, midprice='bm', volatility=2.0, initial_price=100.0, arrival='poisson',
             intensity=[100.0, 100.0], fill_exponent=1.5, dynamics='limit', reward='pnl', initial_inventory=0,
             max_inventory=10_000, seed=3, normalise_action_space=True, normalise_observation_space=True),
        normalised=True)

    # F. cash clipping (small max_cash) together with a tight inventory limit
    n, ns = 16, 50
    run_case(
        "clip_cash",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=5, initial_inventory=0, max_inventory=2, max_cash=150.0,
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                BrownianMotionMidpriceModel(initial_price=100, volatility=2.0, step_size=1 / ns, terminal_time=1.0, num_trajectories=n),
                PoissonArrivalModel(intensity=np.array([40, 40]), step_size=1 / ns, num_trajectories=n)),
            normalise_action_space=False, normalise_observation_space=False, num_trajectories=n),
        ns, n, 2, 5,
        dict(n_steps=ns, terminal_time=1.0, midprice='bm', volatility=2.0, initial_price=100.0, arrival='poisson',
             intensity=[40.0, 40.0], fill_exponent=1.5, dynamics='limit', reward='pnl', initial_inventory=0,
             max_inventory=2, max_cash=150.0, seed=5, normalise_action_space=False, normalise_observation_space=False),
        poisson_thr=40 * (1 / ns))

    extra_cases()


def extra_cases():
    """Plugin classes beyond the five BASELINE configurations."""
    common = dict(normalise_action_space=False, normalise_observation_space=False)

    # G. geometric Brownian motion + non-linear Poisson arrivals + posting at the touch
    n, ns = 32, 80
    run_case(
        "gbm_nonlinear_touch",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=21, initial_inventory=0, max_inventory=4, num_trajectories=n,
            model_dynamics=AtTheTouchModelDynamics(
                midprice_model=GeometricBrownianMotionMidpriceModel(drift=0.05, volatility=0.2, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                arrival_model=PoissonArrivalNonLinearModel(intensity=np.array([60.0, 45.0]), step_size=1 / ns, num_trajectories=n),
                num_trajectories=n, fixed_market_half_spread=0.25),
            **common),
        ns, n, 2, 21,
        dict(n_steps=ns, terminal_time=1.0, midprice="gbm", drift=0.05, volatility=0.2, initial_price=100.0, arrival="poisson_nonlinear",
             intensity=[60.0, 45.0], dynamics="touch", market_half_spread=0.25, reward="pnl", initial_inventory=0, max_inventory=4, seed=21,
             **common),
        poisson_thr=1.0 - np.exp(-45.0 / ns), action_kind="touch")

    # H. Brownian motion with jumps on the agent's fills + terminal exponential utility
    n, ns = 32, 60
    run_case(
        "bmjump_exputility",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=22, initial_inventory=1, max_inventory=30, num_trajectories=n,
            reward_function=ExponentialUtility(risk_aversion=0.01),
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                BrownianMotionJumpMidpriceModel(drift=0.1, volatility=2.0, jump_size=0.3, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                PoissonArrivalModel(intensity=np.array([50.0, 50.0]), step_size=1 / ns, num_trajectories=n)),
            **common),
        ns, n, 2, 22,
        dict(n_steps=ns, terminal_time=1.0, midprice="bm_jump", drift=0.1, volatility=2.0, jump_size=0.3, initial_price=100.0, arrival="poisson",
             intensity=[50.0, 50.0], fill_exponent=1.5, dynamics="limit", reward="exp_utility", risk_aversion=0.01, initial_inventory=1,
             max_inventory=30, seed=22, **common),
        poisson_thr=50.0 / ns)

    # I. OU with jumps + Hawkes arrivals + running inventory penalty (D = 6)
    n, ns = 32, 90
    run_case(
        "oujump_hawkes_running",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=23, initial_inventory=0, max_inventory=25, num_trajectories=n,
            reward_function=RunningInventoryPenalty(0.02, 0.1),
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                OuJumpMidpriceModel(mean_reversion_level=100.0, mean_reversion_speed=0.05, volatility=1.5, jump_size=0.2, initial_price=100.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                HawkesArrivalModel(baseline_arrival_rate=np.array([[12.0, 8.0]]), step_size=1 / ns, jump_size=30.0, mean_reversion_speed=50.0, terminal_time=1.0, num_trajectories=n)),
            **common),
        ns, n, 2, 23,
        dict(n_steps=ns, terminal_time=1.0, midprice="ou_jump", ou_level=100.0, ou_speed=0.05, volatility=1.5, jump_size=0.2, initial_price=100.0,
             arrival="hawkes", intensity=[12.0, 8.0], hawkes_jump=30.0, hawkes_speed=50.0, fill_exponent=1.5, dynamics="limit",
             reward="running", phi=0.02, alpha=0.1, initial_inventory=0, max_inventory=25, seed=23, **common))

    # J. constant midprice
    n, ns = 16, 40
    run_case(
        "constant_midprice",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=24, initial_inventory=0, max_inventory=10, num_trajectories=n,
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                ConstantMidpriceModel(initial_price=50, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                PoissonArrivalModel(intensity=np.array([30.0, 30.0]), step_size=1 / ns, num_trajectories=n)),
            **common),
        ns, n, 2, 24,
        dict(n_steps=ns, terminal_time=1.0, midprice="constant", initial_price=50.0, arrival="poisson", intensity=[30.0, 30.0],
             fill_exponent=1.5, dynamics="limit", reward="pnl", initial_inventory=0, max_inventory=10, seed=24, **common),
        poisson_thr=30.0 / ns)

    # K-N. trading-with-speed dynamics with the four price-impact models (optimal execution)
    n, ns = 32, 100
    speed_cases = [
        ("speed_temp_perm_cjoe", lambda: TemporaryAndPermanentPriceImpact(temporary_impact_coefficient=0.02, permanent_impact_coefficient=0.015, n_steps=ns, terminal_time=1.0, num_trajectories=n),
         CjOeCriterion(per_step_inventory_aversion=0.01, terminal_inventory_aversion=0.05, terminal_time=1.0),
         dict(impact="temp_perm", temporary_impact=0.02, permanent_impact=0.015, impact_step_size=1.0 / ns, reward="cjoe", phi=0.01, alpha=0.05), "speed"),
        ("speed_power_running", lambda: TemporaryPowerPriceImpact(temporary_impact_coefficient=0.03, temporary_impact_exponent=1.5, num_trajectories=n),
         RunningInventoryPenalty(0.01, 0.2),
         dict(impact="temp_power", temporary_impact=0.03, impact_exponent=1.5, reward="running", phi=0.01, alpha=0.2), "speed_positive"),
        ("speed_temp_transient_pnl", lambda: TemporaryAndTransientPriceImpact(temporary_impact_coefficient=0.02, transient_impact_coefficient=0.5, resilience_coefficient=2.0, initial_transient_impact=0.1, linear_kernel_coefficient=0.3, n_steps=ns, terminal_time=1.0, num_trajectories=n),
         PnL(),
         dict(impact="temp_transient", temporary_impact=0.02, transient_impact=0.5, resilience=2.0, initial_transient_impact=0.1, kernel_coefficient=0.3, impact_step_size=1.0 / ns, reward="pnl"), "speed"),
        ("speed_transient_pnl", lambda: TransientPriceImpact(transient_impact_coefficient=0.4, resilience_coefficient=1.0, initial_transient_impact=0.05, linear_kernel_coefficient=0.2, n_steps=ns, terminal_time=1.0, num_trajectories=n),
         PnL(),
         dict(impact="transient", transient_impact=0.4, resilience=1.0, initial_transient_impact=0.05, kernel_coefficient=0.2, impact_step_size=1.0 / ns, reward="pnl"), "speed"),
    ]
    for tag, make_impact, reward, cfg_extra, action_kind in speed_cases:
        mid_dt = 0.02 if tag == "speed_power_running" else 1 / ns  # MD:265 uses the MIDPRICE model's step size
        run_case(
            tag,
            lambda: TradingEnvironment(
                terminal_time=1.0, n_steps=ns, seed=31, initial_inventory=10, max_inventory=1000, num_trajectories=n,
                reward_function=reward,
                model_dynamics=TradinghWithSpeedModelDynamics(
                    midprice_model=BrownianMotionMidpriceModel(drift=0.02, volatility=1.0, initial_price=100, terminal_time=1.0, step_size=mid_dt, num_trajectories=n),
                    price_impact_model=make_impact(), num_trajectories=n),
                **common),
            ns, n, 1, 31 + len(tag),
            dict(n_steps=ns, terminal_time=1.0, midprice="bm", drift=0.02, volatility=1.0, initial_price=100.0, arrival="none", dynamics="speed",
                 midprice_step_size=mid_dt, initial_inventory=10, max_inventory=1000, seed=31, **cfg_extra, **common),
            action_kind=action_kind)

    exogenous_fill_cases()


def exogenous_fill_cases():
    """ExogenousMmFillProbabilityModel (FILL:126-170): two more state columns that hold the exogenous best depths."""
    best = np.array([0.25, 0.375])  # float32-representable, so quotes can sit exactly ON the best depth
    kappa, base = 1.5, 0.8

    def make_fill(n, dt):
        bid = OuMidpriceModel(mean_reversion_level=best[0], mean_reversion_speed=0.1, volatility=0.05, initial_price=best[0],
                              terminal_time=1.0, step_size=dt, num_trajectories=n)
        ask = BrownianMotionMidpriceModel(drift=0.0, volatility=0.04, initial_price=best[1], terminal_time=1.0, step_size=dt, num_trajectories=n)
        bounds = (np.concatenate([bid.min_value, ask.min_value], axis=1)[0], np.concatenate([bid.max_value, ask.max_value], axis=1)[0])
        return ExogenousMmFillProbabilityModel((bid, ask), fill_exponent=kappa, base_fill_probability=base, step_size=dt, num_trajectories=n), bounds

    def prob(depths):
        return (depths > best) * base * np.exp(-kappa * (depths - best)) + (depths <= best)

    def quotes_on_the_best_depth(actions):
        for side in range(2):
            lo, c, hi = f32_neighbours(best[side])
            for j, v in enumerate((lo, c, hi)):
                actions[2::5, (6 + j) % actions.shape[1], side] = v

    common = dict(normalise_action_space=False, normalise_observation_space=False)
    exo_cfg = lambda bounds: dict(fill="exogenous", fill_exponent=kappa, base_fill_probability=base, exo_depth=list(best),  # noqa: E731
                                  exo_depth_lo=[float(v) for v in bounds[0]], exo_depth_hi=[float(v) for v in bounds[1]])

    # O. Brownian midprice + Poisson arrivals + exogenous best depths (D = 6)
    n, ns = 32, 80
    _, bounds = make_fill(n, 1 / ns)
    run_case(
        "exo_fill_bm_poisson",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=41, initial_inventory=0, max_inventory=3, num_trajectories=n,
            model_dynamics=LimitOrderModelDynamics(
                midprice_model=BrownianMotionMidpriceModel(volatility=2.0, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                arrival_model=PoissonArrivalModel(intensity=np.array([50.0, 50.0]), step_size=1 / ns, num_trajectories=n),
                fill_probability_model=make_fill(n, 1 / ns)[0], num_trajectories=n),
            **common),
        ns, n, 2, 41,
        dict(n_steps=ns, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0, arrival="poisson", intensity=[50.0, 50.0],
             dynamics="limit", reward="pnl", initial_inventory=0, max_inventory=3, seed=41, **exo_cfg(bounds), **common),
        poisson_thr=50.0 / ns, fill_prob=prob, action_hook=quotes_on_the_best_depth)

    # P. OU midprice + Hawkes arrivals + exogenous best depths + market orders + running penalty (D = 8, A = 4)
    n, ns = 32, 90
    run_case(
        "exo_fill_hawkes_market",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=42, initial_inventory=(-3, 4), max_inventory=6, num_trajectories=n,
            reward_function=CjMmCriterion(0.02, 0.05, terminal_time=1.0),
            model_dynamics=LimitAndMarketOrderModelDynamics(
                midprice_model=OuMidpriceModel(mean_reversion_level=100.0, mean_reversion_speed=0.02, volatility=1.5, initial_price=100.0,
                                               terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                arrival_model=HawkesArrivalModel(baseline_arrival_rate=np.array([[15.0, 10.0]]), step_size=1 / ns, jump_size=30.0,
                                                 mean_reversion_speed=50.0, terminal_time=1.0, num_trajectories=n),
                fill_probability_model=make_fill(n, 1 / ns)[0], num_trajectories=n, fixed_market_half_spread=0.4),
            **common),
        ns, n, 4, 42,
        dict(n_steps=ns, terminal_time=1.0, midprice="ou", ou_level=100.0, ou_speed=0.02, volatility=1.5, initial_price=100.0, arrival="hawkes",
             intensity=[15.0, 10.0], hawkes_jump=30.0, hawkes_speed=50.0, dynamics="limit_and_market", market_half_spread=0.4, reward="cjmm",
             phi=0.02, alpha=0.05, initial_inventory=[-3, 4], max_inventory=6, seed=42, **exo_cfg(bounds), **common),
        fill_prob=prob, action_hook=quotes_on_the_best_depth)

    # Q. the same model behind normalised actions and observations (TE:112-126 with the extra columns' bounds)
    n, ns = 16, 50
    both = dict(normalise_action_space=True, normalise_observation_space=True)
    run_case(
        "exo_fill_normalised",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=43, initial_inventory=0, max_inventory=5, num_trajectories=n,
            reward_function=RunningInventoryPenalty(0.01, 0.1),
            model_dynamics=LimitOrderModelDynamics(
                midprice_model=BrownianMotionMidpriceModel(volatility=2.0, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                arrival_model=PoissonArrivalModel(intensity=np.array([40.0, 60.0]), step_size=1 / ns, num_trajectories=n),
                fill_probability_model=make_fill(n, 1 / ns)[0], num_trajectories=n),
            **both),
        ns, n, 2, 43,
        dict(n_steps=ns, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0, arrival="poisson", intensity=[40.0, 60.0],
             dynamics="limit", reward="running", phi=0.01, alpha=0.1, initial_inventory=0, max_inventory=5, seed=43, **exo_cfg(bounds), **both),
        normalised=True)


def round2_cases():
    """Round 2: the step_size setter (TE:158-167) used in mid-episode, and a USER-DEFINED midprice plugin."""
    from mbt_gym.gym.index_names import ASK_INDEX, BID_INDEX
    from mbt_gym.stochastic_processes.midprice_models import MidpriceModel

    common = dict(normalise_action_space=False, normalise_observation_space=False)

    # R. step size doubled after 30 of 100 steps, halved again 10 steps later: drift + volatility scaling (MID:63-64), Hawkes
    #    thresholds and decay (ARR:115-123), the reward's dt (RW:131), the clock and the done rule (TE:216-220) all follow
    n, ns = 32, 100
    changes = {30: 0.02, 40: 0.005}
    steps = 30 + 10 + int(round((1.0 - 0.3 - 0.2) / 0.005))
    run_case(
        "step_size_change_hawkes",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=51, initial_inventory=0, max_inventory=20, num_trajectories=n,
            reward_function=RunningInventoryPenalty(0.02, 0.1),
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                BrownianMotionMidpriceModel(drift=0.5, volatility=2.0, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                HawkesArrivalModel(baseline_arrival_rate=np.array([[12.0, 9.0]]), step_size=1 / ns, jump_size=25.0, mean_reversion_speed=30.0, terminal_time=1.0, num_trajectories=n)),
            **common),
        steps, n, 2, 51,
        dict(n_steps=ns, terminal_time=1.0, midprice="bm", drift=0.5, volatility=2.0, initial_price=100.0, arrival="hawkes", intensity=[12.0, 9.0],
             hawkes_jump=25.0, hawkes_speed=30.0, fill_exponent=1.5, dynamics="limit", reward="running", phi=0.02, alpha=0.1, initial_inventory=0,
             max_inventory=20, seed=51, **common),
        step_size_changes=changes)

    # S. the same setter under trading-with-speed dynamics: traded volume (MD:265) and the impact model's step (IMP:87-88)
    n, ns = 32, 80
    changes = {20: 0.025}
    steps = 20 + int(round((1.0 - 20 / ns) / 0.025))
    run_case(
        "step_size_change_speed",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=52, initial_inventory=10, max_inventory=1000, num_trajectories=n,
            reward_function=CjOeCriterion(per_step_inventory_aversion=0.01, terminal_inventory_aversion=0.05, terminal_time=1.0),
            model_dynamics=TradinghWithSpeedModelDynamics(
                midprice_model=BrownianMotionMidpriceModel(drift=0.02, volatility=1.0, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                price_impact_model=TemporaryAndPermanentPriceImpact(temporary_impact_coefficient=0.02, permanent_impact_coefficient=0.015, n_steps=ns, terminal_time=1.0, num_trajectories=n),
                num_trajectories=n),
            **common),
        steps, n, 1, 52,
        dict(n_steps=ns, terminal_time=1.0, midprice="bm", drift=0.02, volatility=1.0, initial_price=100.0, arrival="none", dynamics="speed",
             midprice_step_size=1 / ns, impact="temp_perm", temporary_impact=0.02, permanent_impact=0.015, impact_step_size=1.0 / ns, reward="cjoe",
             phi=0.01, alpha=0.05, initial_inventory=10, max_inventory=1000, seed=52, **common),
        action_kind="speed", step_size_changes=changes)

    # T. a user-defined MidpriceModel written against the reference's plugin API (SP:8-53): a drifting mean-reverting price
    #    whose noise has an arithmetic and a geometric part and which jumps on the agent's own fills
    class UserMixedNoiseOuJumpMidprice(MidpriceModel):
        def __init__(self, drift, volatility, scale_constant, scale_proportional, level, speed, jump_size, initial_price, lo, hi,
                     terminal_time, step_size, num_trajectories, seed=None):
            self.drift, self.volatility, self.a, self.b = drift, volatility, scale_constant, scale_proportional
            self.level, self.speed, self.jump_size = level, speed, jump_size
            super().__init__(min_value=np.array([[lo]]), max_value=np.array([[hi]]), step_size=step_size, terminal_time=terminal_time,
                             initial_state=np.array([[initial_price]]), num_trajectories=num_trajectories, seed=seed)

        def update(self, arrivals, fills, actions, state=None):
            ones = np.ones((self.num_trajectories, 1))
            fills_bid = fills[:, BID_INDEX] * arrivals[:, BID_INDEX]
            fills_ask = fills[:, ASK_INDEX] * arrivals[:, ASK_INDEX]
            jump = (self.jump_size * fills_ask - self.jump_size * fills_bid).reshape(-1, 1)
            s = self.current_state
            scale = self.a + self.b * s
            noise = self.volatility * np.sqrt(self.step_size) * self.rng.normal(size=(self.num_trajectories, 1))
            self.current_state = s + scale * (self.drift * self.step_size * ones + noise) - self.speed * (s - self.level * ones) + jump

    n, ns = 32, 90
    sde = dict(drift=0.3, volatility=0.4, mid_coef_add=1.5, mid_coef_mul=0.02, ou_level=101.0, ou_speed=0.03, jump_size=0.125,
               initial_price=100.0, midprice_lo=80.0, midprice_hi=120.0)
    run_case(
        "user_linear_sde_midprice",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=53, initial_inventory=(-2, 3), max_inventory=4, num_trajectories=n,
            reward_function=CjMmCriterion(0.02, 0.05, terminal_time=1.0),
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                UserMixedNoiseOuJumpMidprice(0.3, 0.4, 1.5, 0.02, 101.0, 0.03, 0.125, 100.0, 80.0, 120.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                PoissonArrivalModel(intensity=np.array([70.0, 55.0]), step_size=1 / ns, num_trajectories=n)),
            **common),
        ns, n, 2, 53,
        dict(n_steps=ns, terminal_time=1.0, midprice="linear_sde", arrival="poisson", intensity=[70.0, 55.0], fill_exponent=1.5, dynamics="limit",
             reward="cjmm", phi=0.02, alpha=0.05, initial_inventory=[-2, 3], max_inventory=4, seed=53, **sde, **common),
        poisson_thr=55.0 / ns)


ONLY = None  # when set: the names of the cases run_case still runs (regenerating some fixtures leaves the others' bytes untouched)


def user_plugin_cases(only=None):
    """User-defined FillProbabilityModel and RewardFunction subclasses written against the reference's plugin API."""
    global ONLY
    ONLY = only
    from mbt_gym.gym.index_names import CASH_INDEX, INVENTORY_INDEX, TIME_INDEX, ASSET_PRICE_INDEX
    from mbt_gym.rewards.RewardFunctions import RewardFunction
    from mbt_gym.stochastic_processes.fill_probability_models import FillProbabilityModel

    # The NumPy-only classes live in tests/numpy_only_plugins.py - ONE source, bound here to the REFERENCE's base classes and in
    # tests/test_gpu_host_callbacks.py to mbt_gym_amd's: the same user code on both sides of the parity test.
    import mbt_gym.gym.index_names as reference_index_names
    from mbt_gym.stochastic_processes.arrival_models import ArrivalModel

    sys.path.insert(0, REPO)
    from tests.numpy_only_plugins import define as define_numpy_only_plugins

    from mbt_gym.stochastic_processes.price_impact_models import PriceImpactModel

    user = define_numpy_only_plugins(FillProbabilityModel, ArrivalModel, RewardFunction, reference_index_names, PriceImpactModel=PriceImpactModel)
    UserPowerLawFill, UserExponentialInventoryCost, UserSeasonalArrivals = user.UserPowerLawFill, user.UserExponentialInventoryCost, user.UserSeasonalArrivals

    common = dict(normalise_action_space=False, normalise_observation_space=False)
    scale, power = 1.25, 1.5
    prob = lambda d: 1.0 / (1.0 + (scale * d) ** power)  # noqa: E731

    def dynamics(n, ns, cls=LimitOrderModelDynamics, mid=None, arr=None, **kw):
        return cls(midprice_model=mid or BrownianMotionMidpriceModel(volatility=2.0, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                   arrival_model=arr or PoissonArrivalModel(intensity=np.array([60.0, 45.0]), step_size=1 / ns, num_trajectories=n),
                   fill_probability_model=UserPowerLawFill(scale, power, step_size=1 / ns, num_trajectories=n), num_trajectories=n, **kw)

    # U. user fill + user reward, limit orders, tight inventory limit
    n, ns = 32, 80
    run_case(
        "user_fill_and_reward",
        lambda: TradingEnvironment(terminal_time=1.0, n_steps=ns, seed=61, initial_inventory=(-2, 3), max_inventory=5, num_trajectories=n,
                                   reward_function=UserExponentialInventoryCost(0.05, 0.3, 0.02), model_dynamics=dynamics(n, ns), **common),
        ns, n, 2, 61,
        dict(n_steps=ns, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0, arrival="poisson", intensity=[60.0, 45.0],
             fill="user_power_law", fill_scale=scale, fill_power=power, dynamics="limit", reward="user_exp_inventory_cost", phi=0.05, eta=0.3,
             alpha=0.02, initial_inventory=[-2, 3], max_inventory=5, seed=61, **common),
        poisson_thr=45.0 / ns, fill_prob=prob)

    # V. user fill with a built-in reward, Hawkes arrivals, OU midprice, limit + market orders, normalised spaces
    n, ns = 24, 70
    both = dict(normalise_action_space=True, normalise_observation_space=True)
    run_case(
        "user_fill_hawkes_market_normalised",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=62, initial_inventory=0, max_inventory=8, num_trajectories=n,
            reward_function=RunningInventoryPenalty(0.01, 0.05),
            model_dynamics=dynamics(
                n, ns, cls=LimitAndMarketOrderModelDynamics, fixed_market_half_spread=0.4,
                mid=OuMidpriceModel(mean_reversion_level=100.0, mean_reversion_speed=0.02, volatility=1.5, initial_price=100.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                arr=HawkesArrivalModel(baseline_arrival_rate=np.array([[15.0, 10.0]]), step_size=1 / ns, jump_size=20.0, mean_reversion_speed=30.0, terminal_time=1.0, num_trajectories=n)),
            **both),
        ns, n, 4, 62,
        dict(n_steps=ns, terminal_time=1.0, midprice="ou", ou_level=100.0, ou_speed=0.02, volatility=1.5, initial_price=100.0, arrival="hawkes",
             intensity=[15.0, 10.0], hawkes_jump=20.0, hawkes_speed=30.0, fill="user_power_law", fill_scale=scale, fill_power=power,
             dynamics="limit_and_market", market_half_spread=0.4, reward="running", phi=0.01, alpha=0.05, initial_inventory=0, max_inventory=8,
             seed=62, **both),
        normalised=True)

    # X. a user-defined, stateless ArrivalModel: a time-of-day intensity profile.  The reference hands the state matrix to
    #    update() (TE:206-211), which is where a plugin written against its API learns the time
    n, ns = 32, 100
    run_case(
        "user_seasonal_arrivals",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=64, initial_inventory=0, max_inventory=6, num_trajectories=n,
            reward_function=RunningInventoryPenalty(0.02, 0.05),
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                BrownianMotionMidpriceModel(volatility=1.5, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                UserSeasonalArrivals([40.0, 30.0], 0.8, 0.5, step_size=1 / ns, num_trajectories=n)),
            **common),
        ns, n, 2, 64,
        dict(n_steps=ns, terminal_time=1.0, midprice="bm", volatility=1.5, initial_price=100.0, arrival="user_seasonal", intensity=[40.0, 30.0],
             seasonal_amplitude=0.8, seasonal_period=0.5, fill_exponent=1.5, dynamics="limit", reward="running", phi=0.02, alpha=0.05,
             initial_inventory=0, max_inventory=6, seed=64, **common))

    # Y. a user-defined midprice with a NON-linear increment: constant elasticity of variance, per trajectory (the reference's
    #    own CEV class broadcasts (N,) noise against an (N, 1) state and cannot be used for N > 1, MID:401-409)
    from mbt_gym.stochastic_processes.midprice_models import MidpriceModel

    UserCevMidprice = user.UserCevMidprice  # (tests/numpy_only_plugins.py: the same source the host-callback tests bind to mbt_gym_amd)

    n, ns = 32, 80
    run_case(
        "user_cev_midprice",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=65, initial_inventory=2, max_inventory=8, num_trajectories=n,
            reward_function=RunningInventoryPenalty(0.01, 0.02),
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                UserCevMidprice(0.05, 0.6, 0.75, 50.0, 20.0, 80.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                PoissonArrivalModel(intensity=np.array([50.0, 50.0]), step_size=1 / ns, num_trajectories=n)),
            **common),
        ns, n, 2, 65,
        dict(n_steps=ns, terminal_time=1.0, midprice="user_cev", drift=0.05, volatility=0.6, cev_gamma=0.75, initial_price=50.0, midprice_lo=20.0,
             midprice_hi=80.0, arrival="poisson", intensity=[50.0, 50.0], fill_exponent=1.5, dynamics="limit", reward="running", phi=0.01, alpha=0.02,
             initial_inventory=2, max_inventory=8, seed=65, **common),
        poisson_thr=50.0 / ns)

    # Z1. a user-defined ArrivalModel WITH STATE (SP:8-53: a subclass carries its own (N, d) current_state): two intensities
    #     like the reference's Hawkes model (ARR:86-126) in which an arrival on one side also excites the other
    UserCrossExcitingHawkes = user.UserCrossExcitingHawkes  # (tests/numpy_only_plugins.py: the same source the host-callback tests bind to mbt_gym_amd)

    n, ns = 32, 90
    run_case(
        "user_cross_hawkes",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=66, initial_inventory=(-2, 3), max_inventory=6, num_trajectories=n,
            reward_function=CjMmCriterion(0.02, 0.05, terminal_time=1.0),
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                OuMidpriceModel(mean_reversion_level=100.0, mean_reversion_speed=0.02, volatility=1.5, initial_price=100.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                UserCrossExcitingHawkes([18.0, 12.0], 25.0, 14.0, 6.0, step_size=1 / ns, terminal_time=1.0, num_trajectories=n)),
            **common),
        ns, n, 2, 66,
        dict(n_steps=ns, terminal_time=1.0, midprice="ou", ou_level=100.0, ou_speed=0.02, volatility=1.5, initial_price=100.0, arrival="user_cross_hawkes",
             intensity=[18.0, 12.0], hawkes_speed=25.0, hawkes_jump=14.0, hawkes_cross=6.0, fill_exponent=1.5, dynamics="limit", reward="cjmm", phi=0.02,
             alpha=0.05, initial_inventory=[-2, 3], max_inventory=6, seed=66, **common))

    # Z2. a user-defined TWO-COLUMN MidpriceModel: price + a mean-reverting short-term alpha that order flow pushes (what the
    #     reference's ShortTermOuAlphaMidpriceModel, MID:149-190, describes and cannot run for N > 1); two normals per step
    UserShortTermAlphaMidprice = user.UserShortTermAlphaMidprice  # (tests/numpy_only_plugins.py)

    n, ns = 32, 80
    alpha = dict(volatility=1.2, alpha_kappa=8.0, alpha_xi=3.0, alpha_eps=0.75, initial_price=100.0, midprice_lo=90.0, midprice_hi=110.0, alpha_lo=-10.0, alpha_hi=10.0)
    for tag, norm in (("user_two_factor_midprice", common), ("user_two_factor_midprice_normalised", dict(normalise_action_space=True, normalise_observation_space=True))):
        run_case(
            tag,
            lambda norm=norm: TradingEnvironment(
                terminal_time=1.0, n_steps=ns, seed=67, initial_inventory=1, max_inventory=5, num_trajectories=n,
                reward_function=RunningInventoryPenalty(0.01, 0.05),
                model_dynamics=lo_dynamics(
                    n, 1 / ns, 1.0,
                    UserShortTermAlphaMidprice(1.2, 8.0, 3.0, 0.75, 100.0, 90.0, 110.0, -10.0, 10.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                    PoissonArrivalModel(intensity=np.array([45.0, 60.0]), step_size=1 / ns, num_trajectories=n)),
                **norm),
            ns, n, 2, 67,
            dict(n_steps=ns, terminal_time=1.0, midprice="user_alpha", arrival="poisson", intensity=[45.0, 60.0], fill_exponent=1.5, dynamics="limit",
                 reward="running", phi=0.01, alpha=0.05, initial_inventory=1, max_inventory=5, seed=67, **alpha, **norm),
            poisson_thr=45.0 / ns, user_normals=True, normalised=norm["normalise_observation_space"])

    # W. user reward with the built-in exponential fill, at the touch
    n, ns = 24, 60
    run_case(
        "user_reward_touch",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=63, initial_inventory=1, max_inventory=4, num_trajectories=n,
            reward_function=UserExponentialInventoryCost(0.1, 0.5, 0.05),
            model_dynamics=AtTheTouchModelDynamics(
                midprice_model=BrownianMotionMidpriceModel(drift=0.1, volatility=1.0, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                arrival_model=PoissonArrivalModel(intensity=np.array([40.0, 40.0]), step_size=1 / ns, num_trajectories=n),
                num_trajectories=n, fixed_market_half_spread=0.25),
            **common),
        ns, n, 2, 63,
        dict(n_steps=ns, terminal_time=1.0, midprice="bm", drift=0.1, volatility=1.0, initial_price=100.0, arrival="poisson", intensity=[40.0, 40.0],
             dynamics="touch", market_half_spread=0.25, reward="user_exp_inventory_cost", phi=0.1, eta=0.5, alpha=0.05, initial_inventory=1,
             max_inventory=4, seed=63, **common),
        poisson_thr=40.0 / ns, action_kind="touch")

    # X1 / X2 (round 4). NumPy-only plugins with TRADING-WITH-SPEED dynamics (MD:243-275): a user reward next to a stateful impact
    #     model, with the inventory limit in reach; a user midprice under the Cartea-Jaimungal execution criterion (RW:39-74)
    n, ns = 32, 100
    run_case(
        "user_reward_speed",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=71, initial_inventory=10, max_inventory=10, num_trajectories=n,
            reward_function=UserExponentialInventoryCost(0.02, 0.15, 0.01),
            model_dynamics=TradinghWithSpeedModelDynamics(
                midprice_model=BrownianMotionMidpriceModel(drift=0.02, volatility=1.0, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                price_impact_model=TemporaryAndPermanentPriceImpact(temporary_impact_coefficient=0.02, permanent_impact_coefficient=0.015, n_steps=ns, terminal_time=1.0, num_trajectories=n),
                num_trajectories=n),
            **common),
        ns, n, 1, 71,
        dict(n_steps=ns, terminal_time=1.0, midprice="bm", drift=0.02, volatility=1.0, initial_price=100.0, arrival="none", dynamics="speed",
             midprice_step_size=1 / ns, impact="temp_perm", temporary_impact=0.02, permanent_impact=0.015, impact_step_size=1.0 / ns,
             reward="user_exp_inventory_cost", phi=0.02, eta=0.15, alpha=0.01, initial_inventory=10, max_inventory=10, seed=71, **common),
        action_kind="speed")
    run_case(
        "user_cev_midprice_speed",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=72, initial_inventory=10, max_inventory=1000, num_trajectories=n,
            reward_function=CjOeCriterion(per_step_inventory_aversion=0.01, terminal_inventory_aversion=0.05, terminal_time=1.0),
            model_dynamics=TradinghWithSpeedModelDynamics(
                midprice_model=UserCevMidprice(0.05, 0.6, 0.75, 50.0, 20.0, 80.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                price_impact_model=TemporaryPowerPriceImpact(temporary_impact_coefficient=0.03, temporary_impact_exponent=1.0, num_trajectories=n),
                num_trajectories=n),
            **common),
        ns, n, 1, 72,
        dict(n_steps=ns, terminal_time=1.0, midprice="user_cev", drift=0.05, volatility=0.6, cev_gamma=0.75, initial_price=50.0, midprice_lo=20.0,
             midprice_hi=80.0, arrival="none", dynamics="speed", midprice_step_size=1 / ns, impact="temp_power", temporary_impact=0.03, impact_exponent=1.0,
             reward="cjoe", phi=0.01, alpha=0.05, initial_inventory=10, max_inventory=1000, seed=72, **common),
        action_kind="speed")

    # X3 (round 4). a user-defined PriceImpactModel (IMP:9-31) that owns the impact-state column: the square-root law on top of a
    #     decaying transient component, under the running inventory penalty; the inventory limit in reach
    run_case(
        "user_impact_speed",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=73, initial_inventory=10, max_inventory=10, num_trajectories=n,
            reward_function=RunningInventoryPenalty(0.01, 0.1),
            model_dynamics=TradinghWithSpeedModelDynamics(
                midprice_model=BrownianMotionMidpriceModel(drift=0.02, volatility=1.0, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                price_impact_model=user.UserSquareRootImpact(0.05, 2.0, 0.3, 10.0, step_size=1 / ns, terminal_time=1.0, num_trajectories=n),
                num_trajectories=n),
            **common),
        ns, n, 1, 73,
        dict(n_steps=ns, terminal_time=1.0, midprice="bm", drift=0.02, volatility=1.0, initial_price=100.0, arrival="none", dynamics="speed",
             midprice_step_size=1 / ns, impact="user_sqrt", temporary_impact=0.05, resilience=2.0, kernel_coefficient=0.3, initial_transient_impact=0.0,
             impact_step_size=1.0 / ns, reward="running", phi=0.01, alpha=0.1, initial_inventory=10, max_inventory=10, seed=73, **common),
        action_kind="speed")

    # X4 (round 4). a user-defined FillProbabilityModel WITH STATE (SP:8-53): the decay rate of the fill probability is a column of its own
    n, ns = 32, 80
    run_case(
        "user_adaptive_fill",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=74, initial_inventory=(-2, 3), max_inventory=5, num_trajectories=n,
            reward_function=RunningInventoryPenalty(0.01, 0.05),
            model_dynamics=LimitOrderModelDynamics(
                midprice_model=BrownianMotionMidpriceModel(volatility=1.5, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                arrival_model=PoissonArrivalModel(intensity=np.array([50.0, 40.0]), step_size=1 / ns, num_trajectories=n),
                fill_probability_model=user.UserAdaptiveFill(1.5, 4.0, 0.5, 0.5, 8.0, step_size=1 / ns, num_trajectories=n), num_trajectories=n),
            **common),
        ns, n, 2, 74,
        dict(n_steps=ns, terminal_time=1.0, midprice="bm", volatility=1.5, initial_price=100.0, arrival="poisson", intensity=[50.0, 40.0], fill="user_adaptive",
             fill_exponent=1.5, fill_kappa_speed=4.0, fill_kappa_jump=0.5, fill_kappa_lo=0.5, fill_kappa_hi=8.0, dynamics="limit", reward="running", phi=0.01,
             alpha=0.05, initial_inventory=[-2, 3], max_inventory=5, seed=74, **common),
        poisson_thr=40.0 / ns, fill_prob=lambda d: np.exp(-1.5 * d))

    # X5 (round 4). a user-defined ArrivalModel whose update() READS THE STATE MATRIX (TE:206-211): pins WHAT the reference hands a process at
    #     that point - new cash / inventory / time, the midprice already advanced, the model's own columns not yet
    n, ns = 32, 90
    run_case(
        "user_state_reading_arrivals",
        lambda: TradingEnvironment(
            terminal_time=1.0, n_steps=ns, seed=75, initial_inventory=(-2, 3), max_inventory=6, num_trajectories=n,
            reward_function=RunningInventoryPenalty(0.01, 0.05),
            model_dynamics=lo_dynamics(
                n, 1 / ns, 1.0,
                BrownianMotionMidpriceModel(drift=0.5, volatility=2.5, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                user.UserStateReadingArrivals([40.0, 30.0], 20.0, 0.6, 3.0, 0.8, 100.0, step_size=1 / ns, terminal_time=1.0, num_trajectories=n)),
            **common),
        ns, n, 2, 75,
        dict(n_steps=ns, terminal_time=1.0, midprice="bm", drift=0.5, volatility=2.5, initial_price=100.0, arrival="user_state_reading", intensity=[40.0, 30.0],
             hawkes_speed=20.0, arrival_tilt=0.6, arrival_sensitivity=3.0, arrival_crowding=0.8, arrival_reference_price=100.0, fill_exponent=1.5, dynamics="limit",
             reward="running", phi=0.01, alpha=0.05, initial_inventory=[-2, 3], max_inventory=6, seed=75, **common))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--only-round4":  # NumPy-only plugins of round 4 (leaves the other fixtures' bytes untouched)
        user_plugin_cases(only=("user_reward_speed", "user_cev_midprice_speed", "user_impact_speed", "user_adaptive_fill", "user_state_reading_arrivals"))
    elif len(sys.argv) > 1 and sys.argv[1] == "--only-user-plugins":
        user_plugin_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "--only-exogenous":  # leave the other fixtures' bytes untouched
        exogenous_fill_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "--only-round3":  # the stateful user processes (leaves the other fixtures' bytes untouched)
        user_plugin_cases(only=("user_cross_hawkes", "user_two_factor_midprice", "user_two_factor_midprice_normalised"))
    elif len(sys.argv) > 1 and sys.argv[1] == "--only-round2":
        round2_cases()
    else:
        main()
        round2_cases()
        user_plugin_cases()
