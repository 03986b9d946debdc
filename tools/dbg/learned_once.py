#!/usr/bin/env python3
"""Two fused rollouts of the default (normalised) environment at 2^20 lanes under an in-kernel 4-64-64-2 MLP policy
(MBT_ACT=tanh|relu): a target for counter collection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mbt_gym_amd import _native
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench_policy import random_mlp
env = TradingEnvironment(num_trajectories=1 << 20, seed=50, n_steps=200)
policy = _native.mlp_policy(random_mlp(np.random.default_rng(0), 4, 64, 2), os.environ.get("MBT_ACT", "tanh"))
for _ in range(2):
    env.reset_device()
    env.rollout_device(policy)
env.synchronize()
env.close()
