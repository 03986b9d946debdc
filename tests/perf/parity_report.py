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
from tests.golden_io import KERNEL_NOISE_CASES as CASES, load_case, step_size_changes  # noqa: E402  (the fixture whose midprice draws on the host: tests/test_gpu_host_callbacks.py)


def run(name, **env_kw):
    cfg, g = load_case(name)
    env = make_env(cfg, noise="injected", **env_kw)
    oracle = OracleEnv(cfg, InjectedNoise(g["u_arr"], g["u_fill"], g["z"], g.get("z_user")))  # only for the lanes the reference would print (clip)
    env.record_events(True)
    env.reset(), oracle.reset()
    changes = step_size_changes(g)
    bad_ev = bad_q = n_clip = 0
    e_r = e_rc = e_rel = e_c = e_s = e_o = 0.0
    for k in range(g["actions"].shape[0]):
        if k in changes:
            env.step_size = changes[k]
            oracle.set_step_size(changes[k])
        speed = cfg.dynamics == "speed"
        env.set_noise(None if speed else g["u_arr"][k], None if speed else g["u_fill"][k], g["z"][k], g["z_user"][k] if "z_user" in g else None)
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


def run_precise(name):
    """precise_state=True: the device carries the reference's float64 state exactly and steps it in the reference's order of
    operations.  Counts of entries that are NOT bit-equal, and the largest deviations where any exist."""
    cfg, g = load_case(name)
    env = make_env(cfg, noise="injected", precise_state=True)
    env.record_events(True)
    env.reset()
    changes = step_size_changes(g)
    speed = cfg.dynamics == "speed"
    bad_ev = bad_state = bad_rew = bad_obs = n_state = 0
    e_state = e_rew = 0.0
    for k in range(g["actions"].shape[0]):
        if k in changes:
            env.step_size = changes[k]
        env.set_noise(None if speed else g["u_arr"][k], None if speed else g["u_fill"][k], g["z"][k], g["z_user"][k] if "z_user" in g else None)
        obs, rew, done, _ = env.step(g["actions"][k])
        if not speed:
            bad_ev += int(np.sum(env.last_arrivals != g["arrivals"][k])) + int(np.sum(env.last_fills != g["fills"][k]))
        want = g["obs"][k]
        bad_obs += int(np.sum(obs != want.astype(np.float32)) - np.sum(np.isnan(obs) & np.isnan(want)))
        if not cfg.normalise_observation_space:
            state = env.state64
            n_state += state.size
            bad_state += int(np.sum(state != want))
            e_state = max(e_state, float(np.max(np.abs(state - want) / np.maximum(1.0, np.abs(want)))))
        bad_rew += int(np.sum(rew != g["rewards"][k].astype(np.float32)))
        e_rew = max(e_rew, float(np.max(np.abs(rew.astype(np.float64) - g["rewards"][k]))))
    env.close()
    shape = f"{cfg.num_trajectories} x {g['actions'].shape[0]}"
    rmax = float(np.abs(g["rewards"]).max())
    print(f"{name:36s} {shape:>10s} {bad_ev:9d} {bad_state:12d} {e_state:12.3e} {bad_obs:10d} {bad_rew:12d} {e_rew:12.3e} {rmax:9.3g}")


print("\n== precise_state=True: float64 state held exactly (float32 + int32 remainder), stepped in the reference's operation order ==")
print("   counts are entries that differ in ANY bit from the reference (state64 vs its float64 state, observation vs np.float32 of its observation, reward vs")
print("   np.float32 of its reward); non-zero only where a transcendental or a user's own expression sits in the path (pow in speed_power_running, exp in")
print("   bmjump_exputility / the user rewards, the association of a user's increment expression: user_cev / user_two_factor)")
print(f"{'fixture':36s} {'lanes x k':>10s} {'arr/fill':>9s} {'state64 !=':>12s} {'max rel d':>12s} {'obs32 !=':>10s} {'reward32 !=':>12s} {'max|dr|':>12s} {'max|r|':>9s}")
for name in CASES:
    run_precise(name)
