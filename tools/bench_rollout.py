#!/usr/bin/env python3
"""Throughput of the fused rollout kernel - an extra, NOT the headline metric of bench.py.
Returns-only rollouts touch HBM once per episode (ALU/RNG-bound: no HBM fraction is quoted); recorded rollouts write
4*(D + A + 1) = 28 B per env-step (AS), time-major."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bench import build_env  # noqa: E402
from mbt_gym_amd import _native  # noqa: E402
from mbt_gym_amd.agents.BaselineAgents import AvellanedaStoikovAgent, FixedSpreadAgent  # noqa: E402


def timed(env, agent, episodes, lib):
    env.reset()
    env.rollout_device(agent)  # warm
    env.synchronize()
    _native.check(lib.mbt_env_timer_begin(env._handle))
    for _ in range(episodes):
        env.reset_device()
        steps, done = env.rollout_device(agent)
        assert done and steps == env.n_steps
    ms = C.c_float(0)
    _native.check(lib.mbt_env_timer_end(env._handle, C.byref(ms)))
    return ms.value / 1e3


def main():
    lib = _native.load_library()
    out = {}
    for log2n in (20, 22):
        n = 1 << log2n
        env = build_env(n, 0, 0)
        for name, agent in (("fixed", FixedSpreadAgent(env, half_spread=0.7)), ("avellaneda_stoikov", AvellanedaStoikovAgent(0.1, env))):
            episodes = 5
            t = timed(env, agent, episodes, lib)
            out[f"returns_only/{name}/2^{log2n}"] = {"env_steps_per_s": n * env.n_steps * episodes / t, "ms_per_episode": t / episodes * 1e3}
        env.close()
    # recorded trajectory: 2^18 lanes x 1000 steps x 28 B = 7.3 GB of HBM writes
    n = 1 << 18
    env = build_env(n, 0, 0)
    import torch

    np_ = env.padded_lanes
    obs = torch.empty((env.n_steps + 1, np_, 4), dtype=torch.float32, device="cuda")
    act = torch.empty((env.n_steps, np_, 2), dtype=torch.float32, device="cuda")
    rew = torch.empty((env.n_steps, np_), dtype=torch.float32, device="cuda")
    agent = AvellanedaStoikovAgent(0.1, env)
    env.reset()
    env.rollout_device(agent, obs_ptr=obs.data_ptr(), act_ptr=act.data_ptr(), rew_ptr=rew.data_ptr())
    env.synchronize()
    _native.check(lib.mbt_env_timer_begin(env._handle))
    for _ in range(3):
        env.reset_device()
        env.rollout_device(agent, obs_ptr=obs.data_ptr(), act_ptr=act.data_ptr(), rew_ptr=rew.data_ptr())
    ms = C.c_float(0)
    _native.check(lib.mbt_env_timer_end(env._handle, C.byref(ms)))
    t = ms.value / 1e3 / 3
    out["recorded/avellaneda_stoikov/2^18"] = {"env_steps_per_s": n * env.n_steps / t, "ms_per_episode": t * 1e3,
                                              "write_GBps": 28.0 * n * env.n_steps / t / 1e9}
    total = rew.sum(dim=0)
    out["recorded/mean_return"] = float(total[:n].mean())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
