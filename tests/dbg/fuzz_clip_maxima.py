#!/usr/bin/env python3
"""The largest reward errors the float32 tier shows on lane-steps where the clip of TE:283-289 fires, over the random
configurations of tests/test_gpu_random_configs.py (the same generators and seeds; MBT_FUZZ_SCALE widens the sample) - what the
fuzz tests' clipped-lane bounds are set from (<= 2x these)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from mbt_gym_amd import _native  # noqa: E402
from oracle.mbt_oracle import InjectedNoise, OracleEnv  # noqa: E402
from tests.env_factory import make_env  # noqa: E402
from tests.random_configs import random_actions, random_config, random_speed_actions, random_speed_config  # noqa: E402

scale = int(os.environ.get("MBT_FUZZ_SCALE", "4"))
out = {"order_book": {"cases": 150 * scale, "clipped_lane_steps": 0, "max_err_clipped": 0.0, "max_err_not_clipped": 0.0},
       "speed": {"cases": 60 * scale, "clipped_lane_steps": 0, "max_err_clipped": 0.0, "max_rel_err_clipped": 0.0}}
for case in range(150 * scale):
    rng = np.random.default_rng(7000 + case)
    n = int(rng.choice([7, 192, 600]))
    cfg = random_config(rng, n)
    if cfg.reward == "exp_utility":
        continue
    env = make_env(cfg, noise="philox")
    steps = cfg.n_steps - int(round(cfg.start_time / cfg.step_size))
    actions = random_actions(rng, cfg, steps)
    draws = [_native.rng_fill(cfg.seed, 0, k, n) for k in range(steps)]
    oracle = OracleEnv(cfg, InjectedNoise(*[np.stack(x) for x in zip(*draws)]))
    env.reset(), oracle.reset()
    for k in range(steps):
        _, rew, _, _ = env.step(actions[k])
        _, o_rew, _ = oracle.step(actions[k].astype(np.float64))
        o_rew = np.broadcast_to(np.asarray(o_rew, dtype=np.float64), (n,))
        clipped = oracle.last_clipped
        err = np.abs(rew - o_rew)
        o = out["order_book"]
        o["clipped_lane_steps"] += int(clipped.sum())
        o["max_err_clipped"] = max(o["max_err_clipped"], float(err[clipped].max(initial=0.0)))
        o["max_err_not_clipped"] = max(o["max_err_not_clipped"], float((err[~clipped] / np.maximum(1.0, np.abs(o_rew[~clipped]))).max(initial=0.0)))
    env.close()
for case in range(60 * scale):
    rng = np.random.default_rng(9000 + case)
    n = int(rng.choice([5, 300, 1100]))
    cfg = random_speed_config(rng, n)
    steps = cfg.n_steps
    actions = random_speed_actions(rng, cfg, steps)
    z = np.stack([_native.rng_fill_quad(cfg.seed, 0, k, n) for k in range(steps)])
    env = make_env(cfg, noise="philox")
    oracle = OracleEnv(cfg, InjectedNoise(np.zeros((steps, n, 2)), np.zeros((steps, n, 2)), z))
    env.reset(), oracle.reset()
    for k in range(steps):
        _, rew, _, _ = env.step(actions[k])
        o_obs, o_rew, _ = oracle.step(actions[k].astype(np.float64))
        raw_q = oracle.state[:, 1]
        clipped = oracle.last_clipped | (np.abs(raw_q) >= cfg.max_inventory - 1e-4)
        err = np.abs(rew - o_rew)
        o = out["speed"]
        o["clipped_lane_steps"] += int(clipped.sum())
        o["max_err_clipped"] = max(o["max_err_clipped"], float(err[clipped].max(initial=0.0)))
        o["max_rel_err_clipped"] = max(o["max_rel_err_clipped"], float((err[clipped] / np.maximum(1.0, np.abs(o_rew[clipped]))).max(initial=0.0)))
    env.close()
print(json.dumps(out, indent=1))
