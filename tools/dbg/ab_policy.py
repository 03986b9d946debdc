#!/usr/bin/env python3
"""A/B of library variants on one box: for each name in MBT_LIB_VARIANTS (comma separated; '' = the real library) a child
process runs the learned-policy rollouts of tools/bench_policy.py and prints us per step; three rounds, interleaved."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import numpy as np
v = os.environ.get("MBT_LIB_VARIANT", "")
if v: os.environ["MBT_LIBRARY_VARIANT"] = v  # (libmbtenv.<v>.so of tools/dbg/build_variant.py; read when the binding is imported)
from mbt_gym_amd import _native
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
from bench_policy import random_mlp, timed_rollouts
n = 1 << 20
env = TradingEnvironment(num_trajectories=n, seed=50, n_steps=200)
rng = np.random.default_rng(0)
layers = random_mlp(rng, 4, 64, 2)
fixed = _native.MbtPolicy(kind=_native.POLICY_FIXED); fixed.params[0] = fixed.params[1] = -0.5
cases = {"fixed": fixed, "linear": _native.linear_policy(rng.normal(0, 0.5, (2, 4)), np.zeros(2)), "relu": _native.mlp_policy(layers, "relu"), "tanh": _native.mlp_policy(layers, "tanh")}
out = {}
for name, pol in cases.items():
    steps, seconds = timed_rollouts(env, pol, 5)
    out[name] = round(seconds / steps * 1e6, 3)
print(json.dumps(out))
''' % (ROOT, ROOT)
variants = os.environ.get("MBT_LIB_VARIANTS", "").split(",")
for rnd in range(3):
    for v in variants:
        env = dict(os.environ, MBT_LIB_VARIANT=v)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"round {rnd} variant {v or '(real)':12s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
