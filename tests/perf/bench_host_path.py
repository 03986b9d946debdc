#!/usr/bin/env python3
"""BASELINE.json configs[0] (the reference's own CPU-runnable case: Avellaneda-Stoikov, N = 1000, 200 steps) through
the three ways a caller can run an episode, in wall-clock per episode:
  gym loop   - the reference's usage: agent.get_action(obs) on the host + env.step(action), NumPy in / NumPy out
               (one H2D copy, one launch, two D2H copies and a stream sync per step: latency-bound, not bandwidth)
  rollout    - the fused rollout kernel with the agent's closed form on the device (generate_trajectory)
  oracle     - the float64 NumPy port of the reference on one host core (cpu baseline)
Also at N = 2^16 and 2^20 to show where the host path turns bandwidth-bound (PCIe)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from mbt_gym_amd.agents.BaselineAgents import AvellanedaStoikovAgent  # noqa: E402
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


def main():
    out = {}
    for log2n, reps in ((None, 20), (16, 5), (20, 2)):
        n = 1000 if log2n is None else 1 << log2n
        cfg = OracleConfig(num_trajectories=n, n_steps=200, terminal_time=1.0, volatility=2.0, initial_price=100.0, intensity=(140.0, 140.0),
                           fill_exponent=1.5, initial_inventory=0, max_inventory=200, seed=50, normalise_action_space=False,
                           normalise_observation_space=False)
        env = make_env(cfg)
        agent = AvellanedaStoikovAgent(risk_aversion=0.1, env=env)
        episode_gym_loop(env, agent)
        t0 = time.perf_counter()
        for _ in range(reps):
            total = episode_gym_loop(env, agent)
        gym_s = (time.perf_counter() - t0) / reps
        generate_trajectory(env, agent)
        t0 = time.perf_counter()
        for _ in range(reps):
            obs, act, rew = generate_trajectory(env, agent)
        roll_s = (time.perf_counter() - t0) / reps
        env.reset()
        env.rollout(agent, record=False)
        t0 = time.perf_counter()
        for _ in range(reps):
            env.reset()
            env.rollout(agent, record=False)
            env.synchronize()
        fused_s = (time.perf_counter() - t0) / reps
        row = {"gym_loop_ms_per_episode": gym_s * 1e3, "gym_loop_us_per_step": gym_s / cfg.n_steps * 1e6, "gym_loop_env_steps_per_s": n * cfg.n_steps / gym_s,
               "generate_trajectory_ms_per_episode": roll_s * 1e3, "generate_trajectory_env_steps_per_s": n * cfg.n_steps / roll_s,
               "rollout_returns_only_ms_per_episode": fused_s * 1e3, "rollout_returns_only_env_steps_per_s": n * cfg.n_steps / fused_s,
               "mean_episode_return": float(total.mean())}
        if n <= 1 << 16:
            o = OracleEnv(cfg, NumpyProtocolNoise(50))
            t0 = time.perf_counter()
            obs_o = o.reset()
            for _ in range(cfg.n_steps):
                obs_o, _, _ = o.step(avellaneda_stoikov_action(cfg, 0.1, obs_o))
            orc_s = time.perf_counter() - t0
            row.update({"oracle_ms_per_episode": orc_s * 1e3, "oracle_env_steps_per_s": n * cfg.n_steps / orc_s})
        out[f"N={n}"] = row
        env.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
