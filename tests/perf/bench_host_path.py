#!/usr/bin/env python3
"""The reference-compatible host API - `env.step(np.ndarray)` - measured piece by piece (PCIe-inclusive; never the bench
line's `value`).  BASELINE.json configs[0] (Avellaneda-Stoikov, N = 1000, 200 steps: the reference's own CPU-runnable
case) and the same model at N = 2^16 and 2^20 lanes:

  env_step_*        env.step() ALONE, K steps with a fixed action, auto-reset at episode ends like a VecEnv consumer:
                      pinned   - the action written into `env.action_buffer` (pinned): three DMA copies + one launch
                      pageable - an ordinary float32 ndarray: one more pass (np.copyto into the pinned action buffer)
                      float64  - the reference's dtype: the same pass also converts
  agent_get_action  the host agent's NumPy arithmetic alone (AvellanedaStoikovAgent.get_action on an (N, 4) observation)
  gym_loop          agent + env.step() as the reference's users write it
  sb3_vec_env       the same steps through StableBaselinesTradingEnvironment.step() (auto-reset, terminal observations)
  pcie              what the box's link does for the step's three copies alone (pinned torch tensors, same sizes): the
                    wire time a step cannot beat
  rollout / oracle  the fused rollout kernel (generate_trajectory) and the NumPy port of the reference, for scale"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from mbt_gym_amd.agents.BaselineAgents import AvellanedaStoikovAgent  # noqa: E402
from mbt_gym_amd.gym.StableBaselinesTradingEnvironment import StableBaselinesTradingEnvironment  # noqa: E402
from mbt_gym_amd.gym.helpers.generate_trajectory import generate_trajectory  # noqa: E402
from oracle.mbt_oracle import NumpyProtocolNoise, OracleConfig, OracleEnv, avellaneda_stoikov_action  # noqa: E402
from tests.env_factory import make_env  # noqa: E402


def episode_gym_loop(env, agent):
    obs = env.reset()
    total = np.zeros(env.num_trajectories, np.float64)
    while True:
        obs, rew, done, _ = env.step(agent.get_action(obs))
        total += rew
        if done[0]:
            return total


def env_only(env, action, steps):
    """`steps` env.step(action) calls, resetting at episode ends; seconds per step."""
    env.reset()
    for _ in range(3):
        obs, rew, done, _ = env.step(action)
    t0 = time.perf_counter()
    for _ in range(steps):
        obs, rew, done, _ = env.step(action)
        if done[0]:
            obs = env.reset()
    return (time.perf_counter() - t0) / steps


def pcie_wire(n, obs_dim, act_dim, reps=20):
    """Seconds for the step's three copies alone (action H2D, observation + reward D2H) between pinned host tensors and HBM."""
    import torch

    dev = torch.device("cuda", 0)
    h_act, h_obs, h_rew = (torch.empty(shape, dtype=torch.float32).pin_memory() for shape in ((n, act_dim), (n, obs_dim), (n,)))
    d_act, d_obs, d_rew = (torch.empty_like(t, device=dev) for t in (h_act, h_obs, h_rew))
    for _ in range(3):
        d_act.copy_(h_act, non_blocking=True), h_obs.copy_(d_obs, non_blocking=True), h_rew.copy_(d_rew, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        d_act.copy_(h_act, non_blocking=True), h_obs.copy_(d_obs, non_blocking=True), h_rew.copy_(d_rew, non_blocking=True)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def host_callback_rows(n=1000, n_steps=200, steps=1500):
    """env.step() at N = 1000 with plugin classes that only have NumPy code (examples/bring_your_own_numpy_plugins.py: a power-law fill
    model, an exponential-inventory-cost reward), i.e. the host-callback route: the user's method runs on the host every step."""
    import warnings

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples"))
    from bring_your_own_numpy_plugins import ExponentialInventoryCost, PowerLawFill  # noqa: E402

    from mbt_gym_amd.gym.ModelDynamics import LimitOrderModelDynamics
    from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
    from mbt_gym_amd.rewards.RewardFunctions import PnL
    from mbt_gym_amd.stochastic_processes.arrival_models import PoissonArrivalModel
    from mbt_gym_amd.stochastic_processes.fill_probability_models import ExponentialFillFunction
    from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel

    def build(user_fill, user_reward, normalised=False):
        dt = 1.0 / n_steps
        fill = PowerLawFill(1.25, 1.5, step_size=dt, num_trajectories=n) if user_fill else ExponentialFillFunction(fill_exponent=1.5, step_size=dt, num_trajectories=n)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return TradingEnvironment(
                terminal_time=1.0, n_steps=n_steps, num_trajectories=n, seed=7, max_inventory=20, reward_function=ExponentialInventoryCost(0.05, 0.3) if user_reward else PnL(),
                model_dynamics=LimitOrderModelDynamics(
                    midprice_model=BrownianMotionMidpriceModel(volatility=2.0, step_size=dt, num_trajectories=n),
                    arrival_model=PoissonArrivalModel(intensity=np.array([140.0, 140.0]), step_size=dt, num_trajectories=n),
                    fill_probability_model=fill, num_trajectories=n),
                normalise_action_space=normalised, normalise_observation_space=normalised)

    rows = {}
    for name, args in (("fill_model_in_numpy", (True, False)), ("fill_model_in_numpy_normalised_spaces", (True, False, True)), ("reward_in_numpy", (False, True)),
                       ("fill_model_and_reward_in_numpy", (True, True))):
        env = build(*args)
        action = np.full((n, 2), -0.7 if len(args) > 2 else 0.6, np.float32)
        t = env_only(env, action, steps)
        depths = np.full((n, 2), 0.6)
        t0 = time.perf_counter()
        for _ in range(2000):
            env.model_dynamics.fill_probability_model._get_fill_probabilities(depths) if args[0] else None
        own = (time.perf_counter() - t0) / 2000
        rows[name] = {"us_per_step": t * 1e6, "env_steps_per_s": n / t, "of_which_the_users_fill_method_us": own * 1e6 if args[0] else 0.0}
        env.close()
    return rows


def main():
    out = {}
    for log2n, reps, steps in ((None, 20, 2000), (16, 5, 600), (20, 2, 120)):
        n = 1000 if log2n is None else 1 << log2n
        cfg = OracleConfig(num_trajectories=n, n_steps=200, terminal_time=1.0, volatility=2.0, initial_price=100.0, intensity=(140.0, 140.0),
                           fill_exponent=1.5, initial_inventory=0, max_inventory=200, seed=50, normalise_action_space=False,
                           normalise_observation_space=False)
        env = make_env(cfg)
        agent = AvellanedaStoikovAgent(risk_aversion=0.1, env=env)
        bytes_per_step = 4 * n * (env.action_dim + env.observation_dim + 1)
        row = {"bytes_over_pcie_per_step": bytes_per_step}

        # env.step() alone
        env.action_buffer[:] = 0.7
        t = env_only(env, env.action_buffer, steps)
        row["env_step_pinned_action"] = {"us_per_step": t * 1e6, "env_steps_per_s": n / t, "GBps": bytes_per_step / t / 1e9}
        pageable = np.full((n, env.action_dim), 0.7, np.float32)
        t = env_only(env, pageable, steps)
        row["env_step_pageable_action"] = {"us_per_step": t * 1e6, "env_steps_per_s": n / t, "GBps": bytes_per_step / t / 1e9}
        t = env_only(env, pageable.astype(np.float64), steps)
        row["env_step_float64_action"] = {"us_per_step": t * 1e6, "env_steps_per_s": n / t, "GBps": bytes_per_step / t / 1e9}
        row["output_buffers_in_pool"] = len(env._host_buffers()["obs"].buffers)
        try:
            wire = pcie_wire(n, env.observation_dim, env.action_dim)
            row["pcie"] = {"us_per_step": wire * 1e6, "GBps": bytes_per_step / wire / 1e9, "env_steps_per_s": n / wire}
        except Exception as exc:  # noqa: BLE001 - torch missing: the split above stands on its own
            row["pcie"] = {"error": str(exc)}

        # the agent alone, then both (the reference's loop)
        obs = env.reset()
        agent.get_action(obs)
        t0 = time.perf_counter()
        for _ in range(max(5, steps // 10)):
            agent.get_action(obs)
        t = (time.perf_counter() - t0) / max(5, steps // 10)
        row["agent_get_action"] = {"us_per_call": t * 1e6}
        episode_gym_loop(env, agent)
        t0 = time.perf_counter()
        for _ in range(reps):
            total = episode_gym_loop(env, agent)
        gym_s = (time.perf_counter() - t0) / reps
        row["gym_loop"] = {"ms_per_episode": gym_s * 1e3, "us_per_step": gym_s / cfg.n_steps * 1e6, "env_steps_per_s": n * cfg.n_steps / gym_s,
                           "mean_episode_return": float(total.mean())}

        # the SB3 VecEnv adapter (SBE:22-37): auto-reset and terminal observations inside step()
        venv = StableBaselinesTradingEnvironment(env)
        venv.reset()
        for _ in range(3):
            venv.step(pageable)
        t0 = time.perf_counter()
        for _ in range(steps):
            obs, rew, done, infos = venv.step(pageable)
        t = (time.perf_counter() - t0) / steps
        row["sb3_vec_env_step"] = {"us_per_step": t * 1e6, "env_steps_per_s": n / t}

        # fused paths and the CPU port, for scale
        generate_trajectory(env, agent)
        t0 = time.perf_counter()
        for _ in range(reps):
            generate_trajectory(env, agent)
        roll_s = (time.perf_counter() - t0) / reps
        held = generate_trajectory(env, agent)  # the caller keeps each episode until the next one has arrived: two recordings alive
        held = generate_trajectory(env, agent)
        t0 = time.perf_counter()
        for _ in range(reps):
            held = generate_trajectory(env, agent)
        roll_held_s = (time.perf_counter() - t0) / reps
        recorded_bytes = sum(a.nbytes for a in held)
        del held
        env.reset()
        env.rollout(agent, record=False)
        t0 = time.perf_counter()
        for _ in range(reps):
            env.reset()
            env.rollout(agent, record=False)
            env.synchronize()
        fused_s = (time.perf_counter() - t0) / reps
        row["generate_trajectory"] = {"ms_per_episode": roll_s * 1e3, "env_steps_per_s": n * cfg.n_steps / roll_s, "recorded_MB": recorded_bytes / 1e6,
                                      "GBps_device_to_host": recorded_bytes / roll_s / 1e9,
                                      "ms_per_episode_result_kept_until_the_next": roll_held_s * 1e3}
        row["rollout_returns_only"] = {"ms_per_episode": fused_s * 1e3, "env_steps_per_s": n * cfg.n_steps / fused_s}
        if n <= 1 << 16:
            o = OracleEnv(cfg, NumpyProtocolNoise(50))
            t0 = time.perf_counter()
            obs_o = o.reset()
            for _ in range(cfg.n_steps):
                obs_o, _, _ = o.step(avellaneda_stoikov_action(cfg, 0.1, obs_o))
            orc_s = time.perf_counter() - t0
            row["oracle"] = {"ms_per_episode": orc_s * 1e3, "env_steps_per_s": n * cfg.n_steps / orc_s}
        out[f"N={n}"] = row
        env.close()
    out["host_callbacks_N=1000"] = host_callback_rows()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
