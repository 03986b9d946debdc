#!/usr/bin/env python3
"""The NumPy restatement of the reference on ALL host cores of the GPU box: K processes,
each stepping N / K trajectories of the benchmark workload - the sharding the reference's MultiprocessTradingEnv intended
(MultiprocessTradingEnv.py:74-80).  Kept out of bench.py (whose cpu_baseline is the single-process figure, like the
reference itself): this one spawns processes.   python tests/perf/cpu_baseline_multicore.py [K ...]"""
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

N, STEPS = 1 << 20, 40


def _work(args):
    lanes, seed, barrier_time = args
    import numpy as np

    from oracle.mbt_oracle import NumpyProtocolNoise, OracleConfig, OracleEnv

    cfg = OracleConfig(num_trajectories=lanes, n_steps=1000, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0,
                       arrival="poisson", intensity=(140.0, 140.0), fill_exponent=1.5, dynamics="limit", reward="pnl",
                       initial_inventory=0, max_inventory=1000, seed=seed, normalise_action_space=False, normalise_observation_space=False)
    env = OracleEnv(cfg, NumpyProtocolNoise(seed))
    env.reset()
    action = np.tile(np.array([[0.7, 0.7]]), (lanes, 1))
    env.step(action)  # warm
    while time.time() < barrier_time:  # start together
        pass
    t0 = time.time()
    for _ in range(STEPS):
        env.step(action)
    return t0, time.time()


def main():
    ks = [int(a) for a in sys.argv[1:]] or [1, 16, 64, os.cpu_count()]
    out = {}
    ctx = mp.get_context("spawn")
    for k in ks:
        lanes = N // k
        with ctx.Pool(k) as pool:
            start = time.time() + 3.0 + 0.02 * k
            spans = pool.map(_work, [(lanes, 50 + i, start) for i in range(k)])
        wall = max(e for _, e in spans) - min(s for s, _ in spans)
        out[f"{k} processes"] = {"env_steps_per_s": lanes * k * STEPS / wall, "lanes_per_process": lanes, "steps": STEPS, "wall_s": wall}
    print(json.dumps({"host_cores": os.cpu_count(), "workload": "Avellaneda-Stoikov, 2^20 lanes in total, oracle/mbt_oracle.py (NumPy float64)", **out}, indent=1))


if __name__ == "__main__":
    main()
