#!/usr/bin/env python3
"""Benchmark of the hot path: env-steps/s of TradingEnvironment.step() on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Avellaneda-Stoikov market making - Brownian midprice (sigma=2, S0=100),
Poisson arrivals (140, 140), exponential fills (kappa=1.5), limit-order dynamics, PnL reward, un-normalised,
T=1, n_steps=1000, fp32 - with num_trajectories = 2^20 PER GPU (the trajectory axis is sharded: weak scaling),
constant quote (0.7, 0.7) resident in HBM, in-kernel Philox noise keyed on the GLOBAL lane id.
One "step" = one env.step() = one launch of the fused kernel over every lane of the rank; the episode restarts
(reset kernel) whenever it ends, inside the timed region, exactly like a VecEnv consumer would.
No data-path collective; the only RCCL traffic is the 3-double all-reduce of the episode-return sums.

Output: ONE JSON line on rank 0 (see README / the driver contract), including
  roofline      - algorithmic bytes per launch / average launch duration from HIP events on the kernel's stream
  cpu_baseline  - the NumPy restatement of the reference (oracle/, bit-matched to it) timed on this host
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

LANES_PER_GPU = 1 << 20
N_STEPS = 1000
BYTES_PER_ENV_STEP = 4 * (4 + 2 + 4 + 1)  # state read + action read + next-state write + reward write (D=4, A=2)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SEED = 50
QUOTE = (0.7, 0.7)


def build_env(n, rank, device):
    """The environment through the public plugin API, one shard of the trajectory axis per rank."""
    from mbt_gym_amd.gym.ModelDynamics import LimitOrderModelDynamics
    from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
    from mbt_gym_amd.stochastic_processes.arrival_models import PoissonArrivalModel
    from mbt_gym_amd.stochastic_processes.fill_probability_models import ExponentialFillFunction
    from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel

    dt = 1.0 / N_STEPS
    dynamics = LimitOrderModelDynamics(
        midprice_model=BrownianMotionMidpriceModel(volatility=2.0, initial_price=100, terminal_time=1.0, step_size=dt, num_trajectories=n),
        arrival_model=PoissonArrivalModel(intensity=np.array([140.0, 140.0]), step_size=dt, num_trajectories=n),
        fill_probability_model=ExponentialFillFunction(fill_exponent=1.5, step_size=dt, num_trajectories=n),
        num_trajectories=n,
    )
    return TradingEnvironment(
        terminal_time=1.0, n_steps=N_STEPS, model_dynamics=dynamics, initial_inventory=0, max_inventory=N_STEPS,
        seed=SEED, num_trajectories=n, normalise_action_space=False, normalise_observation_space=False,
        device=device, trajectory_offset=rank * n,
    )


def run_steps(env, k, device, returns):
    """k env.step() launches.  When an episode ends: reduce its [sum R, sum R^2, count] on the device, all-reduce the
    three doubles across ranks (the only collective on the path, RCCL) and restart the episode (reset kernel), like a
    VecEnv consumer would.  The boundary is pipelined - the device reduction is read back, and the collective waited
    for, one episode later (and everything outstanding before this function returns) - so that 24 bytes of statistics
    do not drain a stream that has a thousand launches in flight."""
    from mbt_gym_amd.distributed import PendingReturnSums

    reduction_in_flight, collective = False, None
    for _ in range(k):
        if env.step_device():
            if collective is not None:
                returns.append(collective.result())
                collective = None
            if reduction_in_flight:
                collective = PendingReturnSums(env.episode_return_sums_end(), device=device)
            env.episode_return_sums_begin()
            reduction_in_flight = True
            env.reset_device()
    if collective is not None:
        returns.append(collective.result())
    if reduction_in_flight:
        returns.append(PendingReturnSums(env.episode_return_sums_end(), device=device).result())


def pmc_traffic(n):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    profiles/r01_pmc_summary.json, produced by tools/pmc_summary.py for this workload at 2^20 lanes); None for
    other sizes.  Counters cannot be read from inside the benchmark process, so this is the profiled value."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if n != LANES_PER_GPU or not os.path.exists(path):
        return None
    for name, row in json.load(open(path)).items():
        if "step_kernel" in name:
            return row["hbm_bytes_per_launch"]
    return None


def cpu_baseline(budget_s=12.0):
    """The reference's algorithm (oracle = NumPy restatement, bit-matched to the reference) on this host, one
    process / one core like the reference, same model and N = 2^20 lanes, numpy PCG64 noise as the reference."""
    from oracle.mbt_oracle import NumpyProtocolNoise, OracleConfig, OracleEnv  # the ONLY use of oracle/ in this file

    n = LANES_PER_GPU
    cfg = OracleConfig(
        num_trajectories=n, n_steps=N_STEPS, terminal_time=1.0, midprice="bm", drift=0.0, volatility=2.0,
        initial_price=100.0, arrival="poisson", intensity=(140.0, 140.0), fill_exponent=1.5, dynamics="limit",
        reward="pnl", initial_inventory=0, max_inventory=N_STEPS, seed=SEED,
        normalise_action_space=False, normalise_observation_space=False,
    )
    env = OracleEnv(cfg, NumpyProtocolNoise(SEED))
    env.reset()
    action = np.tile(np.array([QUOTE], dtype=np.float64), (n, 1))
    env.step(action)  # warm
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and steps < N_STEPS - 2:
        env.step(action)
        steps += 1
    dt = time.perf_counter() - t0
    return {
        "value": n * steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
        "sample": f"{steps} steps x {n} lanes of the same workload, oracle/mbt_oracle.py (NumPy float64, PCG64 noise), "
                  f"{os.cpu_count()} host cores present, 1 used",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--lanes", type=int, default=LANES_PER_GPU, help="trajectories per GPU (default 2^20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for testing)")
    ap.add_argument("--single-device", action="store_true", help="testing: every rank uses GPU 0 (needs --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # one process per GPU; a launcher that narrows each rank's visible devices leaves fewer ordinals than ranks
    gpu = 0 if args.single_device else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(gpu)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", gpu))
        else:
            dist.init_process_group(backend=args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    n = args.lanes
    from mbt_gym_amd.distributed import shard_bounds
    offset, count = shard_bounds(n * world, rank, world)
    assert count == n and offset == rank * n
    env = build_env(n, rank, gpu)
    env.set_action_host(np.tile(np.array([QUOTE], dtype=np.float32), (n, 1)))
    env.reset()

    def barrier():
        env.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    from mbt_gym_amd.distributed import allreduce_return_sums, return_statistics

    device = torch.device("cuda", gpu) if args.backend == "nccl" else torch.device("cpu")
    episode_returns = []
    run_steps(env, args.warmup, device, episode_returns)
    barrier()
    from mbt_gym_amd import _native
    import ctypes as C

    lib = _native.load_library()
    _native.check(lib.mbt_env_timer_begin(env._handle))
    t0 = time.perf_counter()
    episode_returns.clear()
    run_steps(env, args.steps, device, episode_returns)
    episodes = len(episode_returns)
    ms = C.c_float(0)
    _native.check(lib.mbt_env_timer_end(env._handle, C.byref(ms)))  # HIP events on the kernel's stream
    env.synchronize()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([wall, ms.value / 1e3], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, event_s = float(t[0]), float(t[1])
    else:
        event_s = ms.value / 1e3

    # mean return of the last finished episode over ALL shards (or of the partial episode if none finished)
    sums = episode_returns[-1] if episode_returns else allreduce_return_sums(env.episode_return_sums(), device=device)

    if rank == 0:
        total_lanes = n * world
        value = total_lanes * args.steps / wall
        launch_us = event_s / args.steps * 1e6  # includes the reset launches of finished episodes (1 per 1000)
        achieved = BYTES_PER_ENV_STEP * n / (event_s / args.steps) / 1e9
        out = {
            "metric": "env-steps/s (num_trajectories x steps)", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "Avellaneda-Stoikov (Brownian midprice, Poisson arrivals, exponential fills, PnL), "
                            "fused step kernel, BASELINE.json configs[1]",
                "num_trajectories_per_gpu": n, "num_trajectories_total": total_lanes, "n_steps": N_STEPS,
                "action": "constant quote (0.7, 0.7) resident in HBM", "noise": "in-kernel Philox4x32-10",
                "parallelism": f"trajectory axis sharded over {world} GPU(s), no data-path collective",
                "episodes_finished_in_timed_region": episodes,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": pmc_traffic(n), "bytes_per_env_step": BYTES_PER_ENV_STEP, "avg_launch_us": launch_us,
                "note": "algorithmic bytes (44 B/env-step x lanes per launch) / mean launch-to-launch time from HIP "
                        "events on the kernel's stream; at 2^20 lanes the 44 MB working set is Infinity-Cache resident",
            },
            "mean_episode_return": return_statistics(sums)[0],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
