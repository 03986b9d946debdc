#!/usr/bin/env python3
"""Three fused rollouts of the benchmark market (MBT_ROLLOUT_POLICY = fixed | as): a target for counter collection.  Returns only at 2^20
lanes by default; MBT_ROLLOUT_RECORD = 18 | 20: RECORDED (28 B written per lane and step into torch tensors) at 2^18 / 2^20 lanes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_env
from mbt_gym_amd.agents.BaselineAgents import AvellanedaStoikovAgent, FixedSpreadAgent
record = int(os.environ.get("MBT_ROLLOUT_RECORD", "0"))
env = build_env(1 << (record or 20), 0, 0)
agent = AvellanedaStoikovAgent(0.1, env) if os.environ.get("MBT_ROLLOUT_POLICY", "fixed") == "as" else FixedSpreadAgent(env, half_spread=0.7)
pointers = {}
if record:
    import torch
    lanes = env.padded_lanes
    obs = torch.empty((env.n_steps + 1, lanes, 4), dtype=torch.float32, device="cuda")
    act = torch.empty((env.n_steps, lanes, 2), dtype=torch.float32, device="cuda")
    rew = torch.empty((env.n_steps, lanes), dtype=torch.float32, device="cuda")
    pointers = dict(obs_ptr=obs.data_ptr(), act_ptr=act.data_ptr(), rew_ptr=rew.data_ptr())
for _ in range(3):
    env.reset_device()
    env.rollout_device(agent, **pointers)
env.synchronize()
env.close()
