#!/usr/bin/env python3
"""Step time of the speed-dynamics kernels (and the AS kernel for reference) at 2^20 lanes as a function of how many
workgroups a CU may hold (dynamic LDS padding, MBT_STEP_DYNAMIC_LDS): with every workgroup resident at once all loads are
issued together and all stores together; fewer resident workgroups means several rounds whose loads and stores overlap."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import ctypes as C, json, os, sys
sys.path.insert(0, %r)
import numpy as np
from mbt_gym_amd import _native
from oracle.mbt_oracle import OracleConfig
from tests.env_factory import make_env
from tests.perf.bench_configs import BASE, CASES
lib = _native.load_library()
out = {}
for name, (kw, log2n, action, *extra) in CASES.items():
    if not (name.startswith("speed") or name.startswith("cfg1 AS 2^20 (D=4")) or "precise" in name or "^1.5" in name:
        continue
    n = 1 << log2n
    env = make_env(OracleConfig(**{**BASE, **kw, "num_trajectories": n}))
    env.set_action_host(np.tile(np.array([action], np.float32), (n, 1)))
    env.reset()
    env.step_many_device(1500)
    env.synchronize()
    best = 1e9
    for _ in range(3):
        _native.check(lib.mbt_env_timer_begin(env._handle))
        env.step_many_device(2000)
        ms = C.c_float(0)
        _native.check(lib.mbt_env_timer_end(env._handle, C.byref(ms)))
        best = min(best, ms.value * 1e3 / 2000)
    out[name[:40]] = round(best, 3)
    env.close()
print(json.dumps(out))
''' % ROOT
res = {}
for lds in ("default", "0", "20480", "33000", "41000", "54000", "65536"):
    env = dict(os.environ)
    if lds != "default":
        env["MBT_STEP_DYNAMIC_LDS"] = lds
    p = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, env=env, cwd=ROOT)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    res[lds] = json.loads(line[-1]) if line else p.stderr[-400:]
print(json.dumps(res, indent=1))
