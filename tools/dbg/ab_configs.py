#!/usr/bin/env python3
"""A/B of library variants on one box for tests/perf/bench_configs.py: MBT_LIB_VARIANTS=,a,b (comma separated; '' = the real
library), MBT_BENCH_ONLY=substring.  Each variant runs in a child process, three rounds interleaved; prints us per step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, runpy
sys.path.insert(0, %r)
v = os.environ.get("MBT_LIB_VARIANT", "")
if v: os.environ["MBT_LIBRARY_VARIANT"] = v  # (libmbtenv.<v>.so of tools/dbg/build_variant.py; read when the binding is imported)
from mbt_gym_amd import _native
sys.argv = ["bench_configs.py"]
runpy.run_path(os.path.join(%r, "tests", "perf", "bench_configs.py"), run_name="__main__")
''' % (ROOT, ROOT)
for rnd in range(int(os.environ.get("MBT_AB_ROUNDS", "3"))):
    for v in os.environ.get("MBT_LIB_VARIANTS", "").split(","):
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MBT_LIB_VARIANT=v), capture_output=True, text=True)
        try:
            d = json.loads(r.stdout)
            print(f"round {rnd} variant {v or '(real)':10s} " + "  ".join(f"{k[:28]}={x['us_per_step']}" for k, x in d.items()), flush=True)
        except Exception:  # noqa: BLE001
            print(f"round {rnd} variant {v}: {r.stderr.strip()[-400:]}", flush=True)
