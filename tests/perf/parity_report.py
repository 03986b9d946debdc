#!/usr/bin/env python3
"""Measured worst-case errors of the HIP path against every golden fixture (GPU box only): the float32 tiers and, for the
order-book fixtures, the precise_state tier.  Reward errors are split by whether the clip of TE:283-289 changed cash or
inventory on that lane-step (there the reward contains the state's level, not just its increment)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from oracle.mbt_oracle import InjectedNoise, OracleEnv  # noqa: E402
from tests.env_factory import make_env  # noqa: E402
from tests.golden_io import CASES, load_case, step_size_changes  # noqa: E402


def run(name, **env_kw):
    cfg, g = load_case(name)
    env = make_env(cfg, noise="injected", **env_kw)
    oracle = OracleEnv(cfg, InjectedNoise(g["u_arr"], g["u_fill"], g["z"]))  # only for the lanes the reference would print (clip)
    env.record_events(True)
    env.reset(), oracle.reset()
    changes = step_size_changes(g)
    bad_ev = bad_q = n_clip = 0
    e_r = e_rc = e_rel = e_c = e_s = e_o = 0.0
    for k in range(g["actions"].shape[0]):
        if k in changes:
            env.step_size = changes[k]
            oracle.set_step_size(changes[k])
        env.set_noise(g["u_arr"][k], g["u_fill"][k], g["z"][k])
        obs, rew, done, _ = env.step(g["actions"][k])
        oracle.step(g["actions"][k].astype(np.float64))
        clipped = oracle.last_clipped
        n_clip += int(clipped.sum())
        if cfg.dynamics != "speed":
            bad_ev += int(np.sum(env.last_arrivals != g["arrivals"][k])) + int(np.sum(env.last_fills != g["fills"][k]))
        err = np.abs(rew - g["rewards"][k])
        e_r = max(e_r, float(err[~clipped].max(initial=0.0)))
        e_rc = max(e_rc, float(err[clipped].max(initial=0.0)))
        e_rel = max(e_rel, float((err / np.maximum(1.0, np.abs(g["rewards"][k]))).max()))
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
    env.close()
    shape = f"{cfg.num_trajectories} x {g['actions'].shape[0]}"
    rmax = float(np.abs(g["rewards"]).max())
    print(f"{name:36s} {shape:>10s} {bad_ev:9d} {bad_q:9d} {e_r:12.3e} {e_rc:12.3e} {n_clip:8d} {e_rel:12.3e} {rmax:9.3g} {e_c:11.3e} {e_s:11.3e} {e_o:11.3e}")


HEADER = (f"{'fixture':36s} {'lanes x k':>10s} {'arr/fill':>9s} {'inventory':>9s} {'max|dr|':>12s} {'max|dr| clip':>12s} {'clipped':>8s} "
          f"{'max rel dr':>12s} {'max|r|':>9s} {'max|dcash|':>11s} {'max|dmid|':>11s} {'max|dobs|n':>11s}")
print("== float32 state (default) ==   dr = reward error vs the reference; 'clip' = lane-steps where TE:283-289 changed a value; rel = |dr| / max(1, |r|)")
print(HEADER)
for name in CASES:
    run(name)
print("\n== precise_state=True (cash / midprice as float32 pairs, double arithmetic) ==")
print(HEADER)
for name in CASES:
    if name.startswith(("speed_", "exo_fill", "user_fill", "user_reward", "user_seasonal", "user_cev")) or name.endswith("_speed"):
        continue
    run(name, precise_state=True)
