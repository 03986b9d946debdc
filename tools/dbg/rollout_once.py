#!/usr/bin/env python3
"""Three fused returns-only rollouts of the benchmark market at 2^20 lanes (MBT_ROLLOUT_POLICY = fixed | as): a target for counter collection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_env
from mbt_gym_amd.agents.BaselineAgents import AvellanedaStoikovAgent, FixedSpreadAgent
env = build_env(1 << 20, 0, 0)
agent = AvellanedaStoikovAgent(0.1, env) if os.environ.get("MBT_ROLLOUT_POLICY", "fixed") == "as" else FixedSpreadAgent(env, half_spread=0.7)
for _ in range(3):
    env.reset_device()
    env.rollout_device(agent)
env.synchronize()
env.close()
