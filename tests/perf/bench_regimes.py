#!/usr/bin/env python3
"""The two regimes of the step kernel (csrc/mbt_env.hip: tune_for_size): default-policy loads and full occupancy while a
launch's working set fits the Infinity Cache; non-temporal loads and a capped occupancy beyond it.  Every combination of
the two knobs (MBT_STREAM_LOADS, MBT_STEP_DYNAMIC_LDS - read when an environment is created) at every size, so that the
thresholds in tune_for_size are measured, not guessed.   python tests/perf/bench_regimes.py > profiles/rNN_regimes.json"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from mbt_gym_amd import _native  # noqa: E402
from oracle.mbt_oracle import OracleConfig  # noqa: E402
from tests.env_factory import make_env  # noqa: E402

BASE = dict(n_steps=1000, terminal_time=1.0, volatility=2.0, initial_price=100.0, fill_exponent=1.5, initial_inventory=0,
            max_inventory=1000, seed=50, normalise_action_space=False, normalise_observation_space=False)
MODELS = {
    "AS (44 B)": (dict(midprice="bm", arrival="poisson", intensity=(140.0, 140.0), reward="pnl"), [0.7, 0.7], (20, 21, 22, 23, 24)),
    "limit+market (52 B)": (dict(midprice="bm", arrival="poisson", intensity=(140.0, 140.0), dynamics="limit_and_market", reward="pnl", initial_inventory=10),
                            [0.7, 0.7, 0.0, 1.0], (21, 24)),
    "Hawkes+OU (60 B)": (dict(midprice="ou", ou_level=100.0, ou_speed=0.01, arrival="hawkes", intensity=(10.0, 10.0), reward="pnl"), [0.7, 0.7], (22, 24)),
    "speed+impact state (48 B)": (dict(midprice="bm", arrival="none", dynamics="speed", impact="temp_perm", temporary_impact=0.02, permanent_impact=0.015,
                                       reward="cjoe", phi=0.01, alpha=0.05, initial_inventory=10), [0.5], (20, 24)),
    "speed (40 B)": (dict(midprice="bm", arrival="none", dynamics="speed", impact="temp_power", temporary_impact=0.03, impact_exponent=1.0, reward="pnl",
                          initial_inventory=10), [0.5], (20, 24)),
}
KNOBS = [("default policy", None, None), ("loads default, 8 WG/CU", "0", "0"), ("loads default, 5 WG/CU", "0", "32768"),
         ("loads nt, 8 WG/CU", "1", "0"), ("loads nt, 5 WG/CU", "1", "32768"), ("loads nt, 4 WG/CU", "1", "40960")]


def measure(kw, action, log2n, steps):
    n = 1 << log2n
    env = make_env(OracleConfig(**{**BASE, **kw, "num_trajectories": n}))
    env.set_action_host(np.tile(np.array([action], np.float32), (n, 1)))
    env.reset_device()
    lib = _native.load_library()
    env.step_many_device(max(20, steps // 10))
    env.synchronize()
    _native.check(lib.mbt_env_timer_begin(env._handle))
    env.step_many_device(steps)
    ms = C.c_float(0)
    _native.check(lib.mbt_env_timer_end(env._handle, C.byref(ms)))
    d, a = env.observation_dim, env.action_dim
    env.close()
    us = ms.value * 1e3 / steps
    return us, 4 * (2 * d + a + 1) * n / us * 1e-3


def main():
    out = {}
    for name, (kw, action, sizes) in MODELS.items():
        for log2n in sizes:
            row = {}
            for label, stream, lds in KNOBS:
                if name.startswith("speed") and lds not in (None, "0") and label != "default policy":
                    continue  # (the occupancy knob is an order-book experiment)
                for key, val in (("MBT_STREAM_LOADS", stream), ("MBT_STEP_DYNAMIC_LDS", lds)):
                    os.environ.pop(key, None)
                    if val is not None:
                        os.environ[key] = val
                us, gbps = measure(kw, action, log2n, 2000 if log2n <= 21 else 600 if log2n <= 22 else 300)
                row[label] = {"us_per_step": round(us, 2), "algorithmic_GBps": round(gbps), "frac_of_8TBps": round(gbps / 8000, 3)}
            out[f"{name} 2^{log2n}"] = row
    for key in ("MBT_STREAM_LOADS", "MBT_STEP_DYNAMIC_LDS"):
        os.environ.pop(key, None)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
