#!/usr/bin/env python3
"""Throughput of learned policies evaluated inside the fused rollout (csrc/policy_mlp.hpp) against the closed-form rollout
and against the same network driven from PyTorch-ROCm (examples/torch_policy_loop.py's structure): env-steps/s at 2^20 lanes.
python tools/bench_policy.py > profiles/rNN_policy_rollout.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from mbt_gym_amd import _native  # noqa: E402
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment  # noqa: E402


def random_mlp(rng, d, hidden, a):
    def layer(out, inp):
        return (rng.normal(0, 1.0 / np.sqrt(inp), size=(out, inp)).astype(np.float32), rng.normal(0, 0.1, size=out).astype(np.float32))
    return [layer(hidden, d), layer(hidden, hidden), layer(a, hidden)]


def timed_rollouts(env, policy, episodes):
    env.reset_device()
    env.rollout_device(policy)  # warm
    env.synchronize()
    t0 = time.perf_counter()
    steps = 0
    for _ in range(episodes):
        env.reset_device()
        steps += env.rollout_device(policy)[0]
    env.synchronize()
    return steps / episodes, (time.perf_counter() - t0) / episodes


def main():
    n = 1 << int(os.environ.get("MBT_LOG2_LANES", "20"))
    env = TradingEnvironment(num_trajectories=n, seed=50, n_steps=200)  # the reference's default environment (normalised)
    rng = np.random.default_rng(0)
    layers = random_mlp(rng, 4, 64, 2)
    out = {"lanes": n, "n_steps": 200, "environment": "TradingEnvironment() defaults (normalised observations and actions)"}
    fixed = _native.MbtPolicy(kind=_native.POLICY_FIXED)
    fixed.params[0] = fixed.params[1] = -0.5
    cases = {"fixed action (closed form)": fixed, "linear policy (VALU)": _native.linear_policy(rng.normal(0, 0.5, (2, 4)), np.zeros(2)),
             "MLP 4-64-64-2 relu (MFMA fp16)": _native.mlp_policy(layers, "relu"), "MLP 4-64-64-2 tanh (MFMA fp16)": _native.mlp_policy(layers, "tanh")}
    for name, pol in cases.items():
        steps, seconds = timed_rollouts(env, pol, 5)
        flops = 2 * (5 * 64 + 65 * 64 + 65 * 2) * n * steps if "MLP" in name else 0
        out[name] = {"ms_per_episode": seconds * 1e3, "us_per_env_step_batch": seconds / steps * 1e6, "env_steps_per_s": n * steps / seconds,
                     "policy_TFLOPs": flops / seconds / 1e12}
    # the same MLP as a step loop: policy kernel + step kernel per step (what a consumer that needs every observation does)
    pol = cases["MLP 4-64-64-2 tanh (MFMA fp16)"]
    env.reset_device()
    env.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        env.policy_device(pol)
        env.step_device()
    env.synchronize()
    dt = time.perf_counter() - t0
    out["MLP tanh as policy kernel + step kernel per step"] = {"us_per_env_step_batch": dt / 200 * 1e6, "env_steps_per_s": n * 200 / dt}
    try:  # the same network in eager PyTorch-ROCm fp16 on zero-copy views (examples/torch_policy_loop.py)
        import torch

        dev = torch.device("cuda", 0)
        env.set_stream(torch.cuda.current_stream().cuda_stream)
        obs_view = lambda: torch.as_tensor(env.obs_device, device=dev)  # noqa: E731
        act = torch.as_tensor(env.action_device, device=dev)
        ws = [(torch.tensor(w, device=dev, dtype=torch.float16), torch.tensor(b, device=dev, dtype=torch.float16)) for w, b in layers]
        env.reset_device()

        def torch_step():
            x = obs_view().to(torch.float16)
            x = torch.tanh(torch.addmm(ws[0][1], x, ws[0][0].t()))
            x = torch.tanh(torch.addmm(ws[1][1], x, ws[1][0].t()))
            act.copy_(torch.clamp(torch.addmm(ws[2][1], x, ws[2][0].t()), -1, 1))
            env.step_device()

        for _ in range(10):
            torch_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            torch_step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["same MLP in eager PyTorch-ROCm fp16 + step kernel per step"] = {"us_per_env_step_batch": dt / 100 * 1e6, "env_steps_per_s": n * 100 / dt}
    except Exception as exc:  # noqa: BLE001
        out["torch comparison"] = f"unavailable: {exc}"
    env.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
