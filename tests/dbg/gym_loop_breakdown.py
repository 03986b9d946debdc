#!/usr/bin/env python3
"""Where a step of the reference-style loop goes at 2^20 lanes: the host agent's NumPy, env.step(), the caller's own
accumulation - timed piece by piece inside ONE loop (tests/perf/bench_host_path.py times them separately)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from mbt_gym_amd.agents.BaselineAgents import AvellanedaStoikovAgent  # noqa: E402
from oracle.mbt_oracle import OracleConfig  # noqa: E402
from tests.env_factory import make_env  # noqa: E402

n = 1 << 20
cfg = OracleConfig(num_trajectories=n, n_steps=200, terminal_time=1.0, volatility=2.0, initial_price=100.0, intensity=(140.0, 140.0), fill_exponent=1.5,
                   initial_inventory=0, max_inventory=200, seed=50, normalise_action_space=False, normalise_observation_space=False)
env = make_env(cfg)
agent = AvellanedaStoikovAgent(risk_aversion=0.1, env=env)
obs = env.reset()
total = np.zeros(n)
t = {"agent": 0.0, "step": 0.0, "accumulate": 0.0}
for k in range(60):
    a0 = time.perf_counter()
    action = agent.get_action(obs)
    a1 = time.perf_counter()
    obs, rew, done, _ = env.step(action)
    a2 = time.perf_counter()
    total += rew
    a3 = time.perf_counter()
    if k >= 10:
        t["agent"] += a1 - a0
        t["step"] += a2 - a1
        t["accumulate"] += a3 - a2
out = {k: v / 50 * 1e3 for k, v in t.items()}
out["unit"] = "ms per step at 2^20 lanes"
out["action_dtype"] = str(action.dtype)
print(json.dumps(out))
