#!/usr/bin/env python3
"""Step-kernel timing of every BASELINE.json config at its own size (parity-test cases, not bench lines):
algorithmic GB/s = 4*(D + A + D + 1) bytes x lanes / mean launch-to-launch time (HIP events)."""
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
CASES = {
    "cfg1 AS 2^20 (D=4,A=2,44B)": (dict(midprice="bm", arrival="poisson", intensity=(140.0, 140.0), reward="pnl"), 20, [0.7, 0.7]),
    "cfg2 CJP cjmm 2^20 (44B)": (dict(midprice="bm", arrival="poisson", intensity=(140.0, 140.0), reward="cjmm", phi=0.01, alpha=0.001, max_inventory=100), 20, [0.7, 0.7]),
    "cfg2 CJP running 2^20 (44B)": (dict(midprice="bm", arrival="poisson", intensity=(140.0, 140.0), reward="running", phi=0.01, alpha=0.001, max_inventory=100), 20, [0.7, 0.7]),
    "cfg3 Hawkes+OU 2^22 (D=6, exact intensities: 60B credited, 76B moved)": (dict(midprice="ou", ou_level=100.0, ou_speed=0.01, arrival="hawkes", intensity=(10.0, 10.0), reward="pnl"), 22, [0.7, 0.7]),
    "cfg3 Hawkes+OU 2^22, float32 intensities (60B)": (dict(midprice="ou", ou_level=100.0, ou_speed=0.01, arrival="hawkes", intensity=(10.0, 10.0), reward="pnl"), 22, [0.7, 0.7], dict(hawkes_float32_intensities=True)),
    "cfg4 limit+market 2^21 (A=4,52B)": (dict(midprice="bm", arrival="poisson", intensity=(140.0, 140.0), dynamics="limit_and_market", reward="pnl", initial_inventory=10), 21, [0.7, 0.7, 0.0, 1.0]),
    "speed temp+perm impact, CjOe 2^20 (D=5,A=1,48B)": (dict(midprice="bm", arrival="none", dynamics="speed", impact="temp_perm", temporary_impact=0.02, permanent_impact=0.015, reward="cjoe", phi=0.01, alpha=0.05, initial_inventory=10), 20, [0.5]),
    "speed power impact, PnL 2^20 (D=4,A=1,40B)": (dict(midprice="bm", arrival="none", dynamics="speed", impact="temp_power", temporary_impact=0.03, impact_exponent=1.0, reward="pnl", initial_inventory=10), 20, [0.5]),
    "default normalised 2^20 (60B incl. obs)": (dict(midprice="bm", arrival="poisson", intensity=(100.0, 100.0), reward="pnl", normalise_action_space=True, normalise_observation_space=True, max_inventory=10000), 20, [-0.5, -0.5]),
    "speed power impact ^1.5, running penalty 2^20 (40B, the instantiation with powers: power_f32)": (dict(midprice="bm", arrival="none", dynamics="speed", impact="temp_power", temporary_impact=0.03, impact_exponent=1.5, reward="running", phi=0.01, alpha=0.05, initial_inventory=10), 20, [0.5]),
    # the order-book kernels with the general reward (exponential utility; inventory exponents other than 2 through power_f32): no reference configuration
    "AS + exponential utility 2^20 (44B, the general-reward kernel)": (dict(midprice="bm", arrival="poisson", intensity=(140.0, 140.0), reward="exp_utility", risk_aversion=0.1), 20, [0.7, 0.7]),
    "CJP running, inventory exponent 1.5 2^20 (44B, the general-reward kernel: power_f32)": (dict(midprice="bm", arrival="poisson", intensity=(140.0, 140.0), reward="running", phi=0.01, alpha=0.001, inventory_exponent=1.5, max_inventory=100, initial_inventory=50), 20, [0.7, 0.7]),
    # precise_state: the reference's float64 state (float32 row + int32 remainders: +8 B per remainder column and env-step), float64 arithmetic
    "cfg1 AS 2^20, precise_state (60B)": (dict(midprice="bm", arrival="poisson", intensity=(140.0, 140.0), reward="pnl"), 20, [0.7, 0.7], dict(precise_state=True)),
    "cfg3 Hawkes+OU 2^22, precise_state (92B)": (dict(midprice="ou", ou_level=100.0, ou_speed=0.01, arrival="hawkes", intensity=(10.0, 10.0), reward="pnl"), 22, [0.7, 0.7], dict(precise_state=True)),
    "cfg4 limit+market 2^21, precise_state (68B)": (dict(midprice="bm", arrival="poisson", intensity=(140.0, 140.0), dynamics="limit_and_market", reward="pnl", initial_inventory=10), 21, [0.7, 0.7, 0.0, 1.0], dict(precise_state=True)),
    "speed temp+perm impact, CjOe 2^20, precise_state (80B)": (dict(midprice="bm", arrival="none", dynamics="speed", impact="temp_perm", temporary_impact=0.02, permanent_impact=0.015, reward="cjoe", phi=0.01, alpha=0.05, initial_inventory=10), 20, [0.5], dict(precise_state=True)),
}


def main():
    lib = _native.load_library()
    out = {}
    only = os.environ.get("MBT_BENCH_ONLY", "")  # substring of a case name: one case (for counter collection)
    for name, (kw, log2n, action, *extra) in CASES.items():
        if only and only not in name:
            continue
        n = 1 << log2n
        cfg = OracleConfig(**{**BASE, **kw, "num_trajectories": n})
        env_kw = extra[0] if extra else {}
        env = make_env(cfg, **env_kw)
        env.set_action_host(np.tile(np.array([action], np.float32), (n, 1)))
        env.reset()
        # k launches per call into the library (mbt_env_step_many_device), like bench.py: a Python loop of step_device() calls
        # costs 4-6 us of interpreter + ctypes per launch - as much as the lighter kernels take - and would be what is measured
        env.step_many_device(int(os.environ.get("MBT_BENCH_WARMUP", "1500")), auto_reset=True)  # clocks up, like bench.py's prewarm
        env.synchronize()
        steps = int(os.environ.get("MBT_BENCH_STEPS", "1000"))
        _native.check(lib.mbt_env_timer_begin(env._handle))
        env.step_many_device(steps, auto_reset=True)
        ms = C.c_float(0)
        _native.check(lib.mbt_env_timer_end(env._handle, C.byref(ms)))
        us = ms.value * 1e3 / steps
        d, a = env.observation_dim, env.action_dim
        credited = 4 * (d + a + d + 1) + (4 * d if cfg.normalise_observation_space else 0)
        remainders = (4 if (cfg.dynamics == "speed" or cfg.arrival == "hawkes") else 2) if env_kw.get("precise_state") else (2 if cfg.arrival == "hawkes" and not env_kw.get("hawkes_float32_intensities") else 0)
        bytes_step = credited + 8 * remainders  # what the kernel moves: + 4 B read and 4 B written per int32 remainder column
        out[name] = {"us_per_step": round(us, 2), "env_steps_per_s": n / us * 1e6, "credited_bytes_per_env_step": credited, "moved_bytes_per_env_step": bytes_step,
                     "moved_GBps": bytes_step * n / us * 1e-3, "moved_frac_of_8TBps": bytes_step * n / us * 1e-3 / 8000,
                     "credited_frac_of_8TBps": credited * n / us * 1e-3 / 8000}
        env.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
