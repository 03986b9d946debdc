#!/usr/bin/env python3
"""Prints the measured worst-case errors of the HIP path against every golden fixture (GPU box only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from tests.env_factory import make_env  # noqa: E402
from tests.golden_io import CASES, load_case  # noqa: E402

print(f"{'fixture':22s} {'lanes x steps':>14s} {'arr/fill mismatches':>20s} {'inventory mism.':>16s} {'max|d reward|':>14s} {'max|d cash|':>12s} {'max|d mid|':>12s} {'max|d obs| (norm)':>18s}")
for name in CASES:
    cfg, g = load_case(name)
    env = make_env(cfg, noise="injected")
    env.record_events(True)
    env.reset()
    bad_ev = bad_q = 0
    e_r = e_c = e_s = e_o = 0.0
    for k in range(g["actions"].shape[0]):
        env.set_noise(g["u_arr"][k], g["u_fill"][k], g["z"][k])
        obs, rew, done, _ = env.step(g["actions"][k])
        bad_ev += int(np.sum(env.last_arrivals != g["arrivals"][k])) + int(np.sum(env.last_fills != g["fills"][k]))
        e_r = max(e_r, float(np.max(np.abs(rew - g["rewards"][k]))))
        if cfg.normalise_observation_space:
            e_o = max(e_o, float(np.max(np.abs(obs - g["obs"][k]))))
            q = np.rint((obs[:, 1].astype(np.float64) + 1) * cfg.max_inventory - cfg.max_inventory)
            bad_q += int(np.sum(q != np.rint((g["obs"][k][:, 1] + 1) * cfg.max_inventory - cfg.max_inventory)))
        else:
            if cfg.dynamics == "speed":  # real-valued inventory: count lanes off by more than float32 rounding
                bad_q += int(np.sum(np.abs(obs[:, 1] - g["obs"][k][:, 1]) > 1e-6 * (1 + np.abs(g["obs"][k][:, 1]))))
            else:
                bad_q += int(np.sum(obs[:, 1] != g["obs"][k][:, 1]))
            e_c = max(e_c, float(np.max(np.abs(obs[:, 0] - g["obs"][k][:, 0]))))
            e_s = max(e_s, float(np.max(np.abs(obs[:, 3] - g["obs"][k][:, 3]))))
    shape = f"{cfg.num_trajectories} x {g['actions'].shape[0]}"
    print(f"{name:22s} {shape:>14s} {bad_ev:20d} {bad_q:16d} {e_r:14.3e} {e_c:12.3e} {e_s:12.3e} {e_o:18.3e}")
    env.close()
