#!/usr/bin/env python3
"""An on-GPU policy consuming the environment without a host round trip - eagerly, and as a HIP graph.

obs (library buffer, zero-copy torch view) -> a small MLP in PyTorch-ROCm -> actions written in place into the library's
action buffer -> the step: everything on one HIP stream, nothing crosses PCIe.  PyTorch is the consumer here, not the product:
the environment side is the fused HIP step kernel.

Two ways of issuing the same loop:
  eager   env.step_device() once per step from Python - the host pays for every launch (4-5 us each: the MLP's ~7 kernels and the step)
  graph   env.device_clock_begin(); torch.cuda.graph captures K x [policy forward, env.step_device_captured()] ONCE; graph.replay() runs
          K steps per call - the environment's clock (time, episode end, Philox step) lives on the device in this mode, so every replay
          continues the episode, episode ends reset the lanes inside the launch (SB3's VecEnv contract), and the results are the eager
          loop's to the bit.  Below ~2^19 lanes, where the eager loop is bound by launches, this is 2-4 x faster.

    python examples/torch_policy_loop.py [log2_lanes] [steps]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mbt_gym_amd.gym.ModelDynamics import LimitOrderModelDynamics  # noqa: E402
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment  # noqa: E402
from mbt_gym_amd.stochastic_processes.arrival_models import PoissonArrivalModel  # noqa: E402
from mbt_gym_amd.stochastic_processes.fill_probability_models import ExponentialFillFunction  # noqa: E402
from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel  # noqa: E402
import numpy as np  # noqa: E402

GRAPH_STEPS = 25


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    steps -= steps % GRAPH_STEPS
    n, n_steps = 1 << log2n, 1000
    dt = 1.0 / n_steps
    dynamics = LimitOrderModelDynamics(
        midprice_model=BrownianMotionMidpriceModel(volatility=2.0, step_size=dt, num_trajectories=n),
        arrival_model=PoissonArrivalModel(intensity=np.array([140.0, 140.0]), step_size=dt, num_trajectories=n),
        fill_probability_model=ExponentialFillFunction(fill_exponent=1.5, step_size=dt, num_trajectories=n),
        num_trajectories=n)
    env = TradingEnvironment(terminal_time=1.0, n_steps=n_steps, model_dynamics=dynamics, num_trajectories=n, seed=1,
                             max_inventory=1000)  # normalised observations and actions: what a learning agent consumes
    stream = torch.cuda.Stream()
    env.set_stream(stream.cuda_stream)
    env.reset_device()
    policy = torch.nn.Sequential(torch.nn.Linear(env.observation_dim, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                                 torch.nn.Linear(64, env.action_dim), torch.nn.Tanh()).cuda().half()
    # normalised observations live in ONE buffer, and so does the action: views made once stay valid (for raw observations ask
    # env.obs_device_aliases_next - small batches alternate between two state buffers outside device-clock mode)
    obs = torch.as_tensor(env.obs_device, device="cuda")
    action = torch.as_tensor(env.action_device, device="cuda")  # (N, 2) float32, the buffer the step kernel reads

    def act():
        with torch.no_grad():
            action.copy_(policy(obs.half()))

    out = {"lanes": n, "steps": steps}
    with torch.cuda.stream(stream):
        # ---- eager: one Python call per step
        def eager_step():
            act()
            if env.step_device():  # True when the episode ended
                env.reset_device()

        for _ in range(20):
            eager_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eager_step()
        torch.cuda.synchronize()
        out["eager_us_per_step"] = (time.perf_counter() - t0) / steps * 1e6
        # ---- graph: the clock on the device, K x [policy, step] captured once
        env.device_clock_begin(auto_reset=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for _ in range(GRAPH_STEPS):
                act()
                env.step_device_captured()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps // GRAPH_STEPS):
            graph.replay()
        torch.cuda.synchronize()
        out["graph_us_per_step"] = (time.perf_counter() - t0) / steps * 1e6
        clock = env.device_clock_read()
        out["device_clock"] = {k: clock[k] for k in ("time", "episode_step", "steps", "episodes")}
        env.device_clock_end()  # the host's clock takes over again (the graph must not be replayed from here on)
        del graph
    out["graph_speed_up"] = out["eager_us_per_step"] / out["graph_us_per_step"]
    out["env_steps_per_s_with_mlp_policy"] = {"eager": n / out["eager_us_per_step"] * 1e6, "graph": n / out["graph_us_per_step"] * 1e6}
    out["mean_last_reward"] = float(torch.as_tensor(env.reward_device, device="cuda").mean())
    while env.episode_log_pop() is not None:  # [sum R, sum R^2, lanes] of the episodes that ended inside the graph
        out["episodes_logged"] = out.get("episodes_logged", 0) + 1
    print(json.dumps(out))
    env.close()


if __name__ == "__main__":
    main()
