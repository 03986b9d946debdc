#!/usr/bin/env python3
"""The zero-copy consumer loop, measured: a policy that lives on the device writes `action_device`, the environment steps on the same
stream, nothing crosses PCIe (the consumer of gym/StableBaselinesTradingEnvironment.py:25-37 at N >= 2^16).

Per lane count, Avellaneda-Stoikov (BASELINE configs[1]'s models), two policies x three ways of issuing the steps:
  constant action   - the action buffer is written once: the loop is the environment alone
  torch linear+tanh - action = tanh(obs @ W + b) written into `action_device` in place (two torch kernels per step)
  eager     one Python call of step_device() (and the torch ops) per step: host-bound below ~2^19 lanes
  one_call  k launches in ONE library call (mbt_env_step_many_device) - constant action only: what the host costs from C
  graph     [policy, step_device_captured] x k captured once with torch.cuda.graph in device-clock mode, replayed

    python tools/bench_device_loop.py [--lanes 1000,65536,1048576] [--seconds 0.4]        -> one JSON object on stdout
`bench.py` imports `device_policy_loop_block` for the N = 1 line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

GRAPH_STEPS = 50  # steps per captured graph


def _env(n, device, normalised):
    from mbt_gym_amd.gym.ModelDynamics import LimitOrderModelDynamics
    from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
    from mbt_gym_amd.stochastic_processes.arrival_models import PoissonArrivalModel
    from mbt_gym_amd.stochastic_processes.fill_probability_models import ExponentialFillFunction
    from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel

    n_steps = int(os.environ.get("MBT_DEVICE_LOOP_N_STEPS", "1000"))  # (a short episode makes the episode-end launches visible: tools/dbg/r06_captured_ab.py)
    dt = 1.0 / n_steps
    dynamics = LimitOrderModelDynamics(
        midprice_model=BrownianMotionMidpriceModel(volatility=2.0, initial_price=100, terminal_time=1.0, step_size=dt, num_trajectories=n),
        arrival_model=PoissonArrivalModel(intensity=np.array([140.0, 140.0]), step_size=dt, num_trajectories=n),
        fill_probability_model=ExponentialFillFunction(fill_exponent=1.5, step_size=dt, num_trajectories=n), num_trajectories=n)
    return TradingEnvironment(terminal_time=1.0, n_steps=n_steps, model_dynamics=dynamics, initial_inventory=0, max_inventory=n_steps, seed=50,
                              num_trajectories=n, normalise_action_space=normalised, normalise_observation_space=normalised, device=device)


def _rate(run_chunk, steps_per_chunk, seconds, sync):
    """us per step of `run_chunk` (which enqueues steps_per_chunk steps), over about `seconds` of wall time after a warm-up."""
    for _ in range(3):
        run_chunk()
    sync()
    chunks, t0 = 0, time.perf_counter()
    while True:
        run_chunk()
        chunks += 1
        if chunks % 4 == 0:  # (bounded queue depth: a consumer synchronises now and then, too)
            sync()
            if time.perf_counter() - t0 > seconds:
                break
    sync()
    return (time.perf_counter() - t0) / (chunks * steps_per_chunk) * 1e6


def measure(n, device=0, seconds=0.4):
    import torch

    out = {"lanes": n}
    stream = torch.cuda.Stream(device)

    def sync():
        torch.cuda.synchronize(device)

    for policy in ("constant_action", "torch_linear_tanh"):
        normalised = policy != "constant_action"  # (a learned policy sees normalised observations, TE:112-118)
        env = _env(n, device, normalised)
        env.set_stream(stream.cuda_stream)
        env.reset_device()
        env.set_action_host(np.tile(np.array([[0.7, 0.7]] if not normalised else [[-0.5, -0.5]], dtype=np.float32), (n, 1)))
        obs_t = torch.as_tensor(env.obs_device, device=f"cuda:{device}")
        action_t = torch.as_tensor(env.action_device, device=f"cuda:{device}")
        weight = (torch.randn(obs_t.shape[1], 2, device=obs_t.device) * 0.05).contiguous()
        bias = torch.tensor([-0.4, -0.4], device=obs_t.device)
        hidden = torch.empty((n, 2), device=obs_t.device)

        def act():
            torch.addmm(bias, obs_t, weight, out=hidden)
            torch.tanh(hidden, out=action_t)

        row = {}
        with torch.cuda.stream(stream):
            # eager: one Python call per step (and per torch op); the host's clock, auto-reset through the library's own loop of one
            def eager_chunk():
                for _ in range(GRAPH_STEPS):
                    if policy != "constant_action":
                        act()
                    env.step_many_device(1, auto_reset=True)

            row["eager_us_per_step"] = _rate(eager_chunk, GRAPH_STEPS, seconds, sync)
            if policy == "constant_action":
                row["one_call_us_per_step"] = _rate(lambda: env.step_many_device(GRAPH_STEPS, auto_reset=True), GRAPH_STEPS, seconds, sync)
            while env.episode_log_pop(wait=True) is not None:
                pass
            # graph: the clock on the device, [policy, step] x GRAPH_STEPS captured once
            env.device_clock_begin(auto_reset=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for _ in range(GRAPH_STEPS):
                    if policy != "constant_action":
                        act()
                    env.step_device_captured()
            row["graph_us_per_step"] = _rate(graph.replay, GRAPH_STEPS, seconds, sync)
            # the captured step issued OUTSIDE a capture: one library call = the align kernel + the step (two launches) - the mode's slow
            # way in, for debugging and for the tests' step-by-step comparisons; an eager loop belongs to step_device()
            def captured_eager_chunk():
                for _ in range(GRAPH_STEPS):
                    if policy != "constant_action":
                        act()
                    env.step_device_captured()

            row["captured_step_outside_a_graph_us_per_step"] = _rate(captured_eager_chunk, GRAPH_STEPS, seconds, sync)
            now = env.device_clock_read()
            row["steps_taken_in_device_clock_mode"], row["episodes_ended_there"] = now["steps"], now["episodes"]
            env.device_clock_end()
            del graph
        for key in ("eager", "one_call", "graph"):
            if key + "_us_per_step" in row:
                row[key + "_env_steps_per_s"] = n / row[key + "_us_per_step"] * 1e6
        row["graph_over_eager"] = row["eager_us_per_step"] / row["graph_us_per_step"]
        out[policy] = row
        env.close()
    return out


def device_policy_loop_block(device=0, lanes=(1000, 1 << 16, 1 << 20), seconds=0.4):
    rows = []
    for n in lanes:
        try:
            rows.append(measure(n, device, seconds))
        except Exception as exc:  # noqa: BLE001 - a block of the line, never the reason the line is missing
            rows.append({"lanes": n, "error": f"{type(exc).__name__}: {exc}"})
    return {"what": "zero-copy consumer loop (SURVEY 8f-4): policy on the device -> action_device -> step on the same stream; Avellaneda-Stoikov, "
                    f"n_steps 1000, auto-reset; `graph` = torch.cuda.graph of [policy, mbt_env_step_device_captured] x {GRAPH_STEPS} in device-clock mode, "
                    "`eager` = one Python call per step (host clock), `one_call` = k launches in one library call",
            "graph_steps": GRAPH_STEPS, "rows": rows}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", default="1000,65536,1048576")
    ap.add_argument("--seconds", type=float, default=0.4)
    args = ap.parse_args()
    print(json.dumps(device_policy_loop_block(0, tuple(int(x) for x in args.lanes.split(",")), args.seconds), indent=1))
