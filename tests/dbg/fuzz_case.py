#!/usr/bin/env python3
"""One random speed configuration of tests/test_gpu_random_configs.py in detail: the lane with the largest reward error."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mbt_gym_amd import _native
from oracle.mbt_oracle import InjectedNoise, OracleEnv, action_bounds
from tests.env_factory import make_env
from tests.test_gpu_random_configs import _random_speed_config

seed, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed + 9000 + case)
n = int(rng.choice([5, 300, 1100]))
cfg = _random_speed_config(rng, n)
print(cfg)
steps = cfg.n_steps
lo, hi = action_bounds(cfg)
positive = cfg.impact == "temp_power" and cfg.impact_exponent != 1.0
if cfg.normalise_action_space:
    actions = rng.uniform(0.0 if positive else -0.4, 0.4, size=(steps, n, 1)).astype(np.float32)
else:
    actions = (rng.uniform(0.0 if positive else -0.4, 0.4, size=(steps, n, 1)) * hi).astype(np.float32)
z = np.stack([_native.rng_fill_quad(cfg.seed, 0, k, n) for k in range(steps)])
env = make_env(cfg, noise="philox")
oracle = OracleEnv(cfg, InjectedNoise(np.zeros((steps, n, 2)), np.zeros((steps, n, 2)), z))
obs, o_obs = env.reset(), oracle.reset()
grad = (oracle.obs_hi.astype(np.float64) - oracle.obs_lo) / 2
print("obs_lo", oracle.obs_lo, "obs_hi", oracle.obs_hi, "act lo/hi", lo, hi, "max_cash", oracle.max_cash)
raw = lambda x: (np.asarray(x, np.float64) + 1) * grad + oracle.obs_lo if cfg.normalise_observation_space else np.asarray(x, np.float64)
r_prev, o_prev = raw(obs), raw(o_obs)
for k in range(steps):
    obs, rew, dones, _ = env.step(actions[k])
    o_obs, o_rew, o_dones = oracle.step(actions[k].astype(np.float64))
    st = env.get_state() if hasattr(env, "get_state") else None
    r, o = raw(obs), raw(o_obs)
    err = np.abs(rew - o_rew)
    i = int(np.argmax(err - 4e-6 * np.abs(o_rew)))
    print(f"k={k:2d} max|dr|={err.max():.3e} lane {i}: r={rew[i]:.6f} ref={o_rew[i]:.6f} a={actions[k][i,0]:.5f} z={z[k][i]:+.4f} "
          f"q={o[i,1]:.5f} dq={r[i,1]-o[i,1]:+.2e} S={o[i,3]:.4f} dS={r[i,3]-o[i,3]:+.2e} cash={o[i,0]:.3f} dcash={r[i,0]-o[i,0]:+.2e} "
          f"Sprev={o_prev[i,3]:.4f} dSprev={r_prev[i,3]-o_prev[i,3]:+.2e} qprev={o_prev[i,1]:.5f} dqprev={r_prev[i,1]-o_prev[i,1]:+.2e} "
          f"max|dS|={np.abs(r[:,3]-o[:,3]).max():.2e} max|dcash|={np.abs(r[:,0]-o[:,0]).max():.2e} clipped={int(oracle.last_clipped.sum())}")
    r_prev, o_prev = r, o
    if st is not None and k == steps - 1:
        print("device state row", st[i])
env.close()
