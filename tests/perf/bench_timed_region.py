#!/usr/bin/env python3
"""Where the fixed cost of bench.py's timed region goes (GPU box only).  The driver runs `--steps 20 --warmup 5`: 20 launches
of ~6.8 us sit between two synchronisation points, so every microsecond of begin / end latency is 0.7 % of `value`.
Prints the median host time of each piece of bench.timed_steps, with the stream idle and with k launches in flight."""
import ctypes as C
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from mbt_gym_amd import _native  # noqa: E402


def median_us(fn, reps=300):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        out.append((time.perf_counter() - t0) * 1e6)
    return statistics.median(out)


def main():
    lib = _native.load_library()
    torch.cuda.set_device(0)
    env = bench.build_env(bench.LANES_PER_GPU, 0, 0)
    env.step_many_device(4000, auto_reset=True)
    env.synchronize()
    ms = C.c_float(0)
    out = {}
    out["idle_torch_cuda_synchronize_us"] = median_us(torch.cuda.synchronize)
    out["idle_env_synchronize_us"] = median_us(env.synchronize)
    out["idle_timer_begin_us"] = median_us(lambda: lib.mbt_env_timer_begin(env._handle))
    env.synchronize()
    out["idle_timer_end_us"] = median_us(lambda: lib.mbt_env_timer_end(env._handle, C.byref(ms)))
    for k in (1, 5, 20, 100):
        def enqueue_only():
            env.step_many_device(k, auto_reset=True)
        def enqueue_and_wait():
            env.step_many_device(k, auto_reset=True)
            env.synchronize()
        def sync_all(barrier=True):
            env.synchronize()
            torch.cuda.synchronize()
        events, walls = [], []
        def bench_region_events():  # bench.py's own timed region
            wall, event_s, _ = bench.timed_steps(env, lib, k, sync_all)
            walls.append(wall * 1e6)
            events.append(event_s * 1e6)
        row = {}
        t = []
        for _ in range(200):  # the host is ahead of the device: time the enqueue alone, then drain untimed
            t0 = time.perf_counter()
            enqueue_only()
            t.append((time.perf_counter() - t0) * 1e6)
            env.synchronize()
        row["enqueue_us"] = statistics.median(t)
        row["enqueue_and_wait_us"] = median_us(enqueue_and_wait, 200)
        median_us(bench_region_events, 200)
        row["bench_region_us"] = statistics.median(walls)
        row["event_us"] = statistics.median(events)
        row["fixed_us"] = row["bench_region_us"] - row["event_us"]
        out[f"k={k}"] = row
    env.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
