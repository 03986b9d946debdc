#!/usr/bin/env python3
"""Benchmark of the hot path: env-steps/s of TradingEnvironment.step() on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: spawns its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Avellaneda-Stoikov market making - Brownian midprice (sigma=2, S0=100),
Poisson arrivals (140, 140), exponential fills (kappa=1.5), limit-order dynamics, PnL reward, un-normalised,
T=1, n_steps=1000, fp32 - with num_trajectories = 2^20 PER GPU (the trajectory axis is sharded: weak scaling),
constant quote (0.7, 0.7) resident in HBM, in-kernel Philox noise keyed on the GLOBAL lane id.
One "step" = one env.step() = one launch of the fused kernel over every lane of the rank; the episode restarts
(reset kernel) whenever it ends, inside the timed region, exactly like a VecEnv consumer would.
No data-path collective; the only RCCL traffic is the 3-double all-reduce of the episode-return sums, enqueued on the
environment's stream through the C ABI (mbt_env_set_communicator), one per finished episode.  For N > 1 the line says how
many ranks RCCL itself reports (`rccl_ranks_seen`, ncclCommCount), checks one all-reduce against its known answer and
reports the collective's own latency (`collective`), measured after the timed region: at the driver's --steps 20 no
1000-step episode ends inside the timed region, and forcing one in would charge 20 steps with a cost paid once per 1000.

Timed region: barrier; t0; ONE call into the library that enqueues the K launches (mbt_env_step_many_device); wait for the
stream; torch.cuda.synchronize(); t1.  Nothing else is inside it.  Before the W warm-up steps the GPU's clocks are
brought up with untimed steps (`prewarm_steps` in the output): the driver's W = 5 is 35 microseconds of work.

Output: ONE JSON line on rank 0 (see README / the driver contract), including
  roofline      - algorithmic bytes per launch / average launch duration from HIP events on the kernel's stream, for the
                  timed region, plus (N = 1) the same kernel measured at 2^24 lanes, where the working set (738 MB) no
                  longer fits the 256 MB Infinity Cache: the HBM-resident regime
  cpu_baseline  - the NumPy restatement of the reference (oracle/, bit-matched to it) timed on this host: one core (like
                  the reference) and K processes x N/K lanes (the layout MultiprocessTradingEnv intended)
"""
import argparse
import glob
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# (ranks of one node share device memory handles through dmabuf only on this driver: RCCL's set-up fails with "hipIpcGetMemHandle: invalid
# argument" otherwise.  Exported on the boxes this runs on; kept here for a launcher that starts the ranks from a cleaner environment.)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # (this stack's default; with kernel arguments in host memory every launch waits 1.2-3 us longer: profiles/r06_kernarg_layout.txt)

import numpy as np  # noqa: E402

LANES_PER_GPU = 1 << 20
HBM_RESIDENT_LANES = 1 << 24
N_STEPS = 1000
BYTES_PER_ENV_STEP = 4 * (4 + 2 + 4 + 1)  # state read + action read + next-state write + reward write (D=4, A=2)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SEED = 50
QUOTE = (0.7, 0.7)
PREWARM_STEPS_AT_2_20 = 8192  # untimed, ~55 ms of launches: brings the clocks up before the warm-up the caller asked for


class phase:
    """A named range around a phase of the benchmark - roctx, through torch.cuda.nvtx (rocTX on ROCm builds): `rocprofv3 --marker-trace -- python
    bench.py` shows which launches belong to the timed region, the warm-up, each block of the line.  Costs nothing when no tracer listens; never
    inside the timed bracket itself (the range is pushed before the barrier and popped after the clock stopped)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        try:
            import torch

            torch.cuda.nvtx.range_push(self.name)
            self.pushed = True
        except Exception:  # noqa: BLE001 - a build without the marker library: ranges are an aid, not a requirement
            self.pushed = False
        return self

    def __exit__(self, *exc):
        if self.pushed:
            import torch

            torch.cuda.nvtx.range_pop()
        return False


def default_prewarm_steps(lanes):
    """The same ~55 ms at any size - and a function of the arguments alone: every rank takes EXACTLY the same number of
    steps, so every rank finishes the same number of episodes and enqueues the same number of return all-reduces (a
    time-based warm-up would let the ranks drift apart by an episode, and the odd collective out would never complete)."""
    return int(min(PREWARM_STEPS_AT_2_20, max(64, PREWARM_STEPS_AT_2_20 * LANES_PER_GPU // max(1, lanes))))


# BASELINE.json configs[1..4] as the public plugin API builds them (sizes, parameters, credited bytes).
# `kernel`: what decides the step kernel the library picks (mbt_env.hip: pick_kernel) - arrival layout (0 Poisson, 1 Hawkes), dynamics
# (0 limit, 1 limit + market), plain Brownian midprice, reward weight (0 PnL, 1 quadratic penalties) - from which kernel_shape() forms the
# kernel's list of named tags and kernel_name() the name a rocprofv3 kernel-stats row carries.
WORKLOADS = {
    "cfg1": dict(label="cfg1 Avellaneda-Stoikov (BM, Poisson, exponential fills, PnL)", lanes=1 << 20, dim=4, act=2, kernel=(0, 0, True, 0)),
    "cfg2_cjmm": dict(label="cfg2 Cartea-Jaimungal-Penalva, CjMmCriterion(0.01, 0.001)", lanes=1 << 20, dim=4, act=2, kernel=(0, 0, True, 1)),
    "cfg2_running": dict(label="cfg2 Cartea-Jaimungal-Penalva, RunningInventoryPenalty(0.01, 0.001)", lanes=1 << 20, dim=4, act=2, kernel=(0, 0, True, 1)),
    "cfg3": dict(label="cfg3 Hawkes arrivals + OU midprice, PnL", lanes=1 << 22, dim=6, act=2, kernel=(1, 0, False, 0)),
    "cfg4": dict(label="cfg4 limit + market orders, Bernoulli(0.01) market-order flags, PnL", lanes=1 << 21, dim=4, act=4, kernel=(0, 1, True, 0)),
}


def credited_bytes(key):
    """ALGORITHMIC bytes per env-step, SURVEY.md section 8d: state read + action read + next-state write + reward write in
    float32 - 44 / 60 / 52 B.  The int32 remainders a tier also moves (below) are overhead, not credit."""
    w = WORKLOADS[key]
    return 4 * (w["dim"] + w["act"] + w["dim"] + 1)


def remainder_columns(key, precise, lam32=False):
    """int32 remainder columns per lane (include/mbt_env.h): precise_state holds [cash, midprice (, bid / ask intensity)] exactly, the
    float32 tier of a Hawkes model its two intensities - unless `hawkes_float32_intensities` (lam32) opts out."""
    hawkes = WORKLOADS[key]["kernel"][0] == 1
    if precise:
        return 4 if hawkes else 2
    return 2 if hawkes and not lam32 else 0


def moved_bytes(key, precise, lam32=False):
    """Bytes per env-step the kernel actually moves: the credited ones + 8 per remainder column (4 read, 4 written) - 60 / 92 / 68 B in
    the precise_state tier, 76 B for Hawkes rows with exact intensities.  tests/test_bench_bytes.py ties this to the PMC summaries."""
    return credited_bytes(key) + 8 * remainder_columns(key, precise, lam32)


def tier_label(key, precise, lam32=False):
    if precise:
        return "precise_state (the reference's float64 arithmetic)"
    if WORKLOADS[key]["kernel"][0] == 1:
        return "float32, float32 intensities (hawkes_float32_intensities: 60 B rows, arrivals not exact)" if lam32 else "float32, exact Hawkes intensities (default: arrivals = the reference's)"
    return "float32"


def kernel_shape(key, precise, lam32=False):
    """The named tags of the step kernel the library picks for a workload (csrc/step_kernel.hpp: Variant<tags...>; the mapping from a
    configuration is csrc/kernel_table.hpp: OrderBookShape), in the table's order - e.g. ["brownian", "pnl"] for BASELINE configs[1]."""
    arr, dyn, bm, rew = WORKLOADS[key]["kernel"]
    tags = []
    if arr == 1:
        tags.append("hawkes" if (precise or lam32) else "hawkes_exact")  # (precise_state holds EVERY column exactly: the plain Hawkes layout)
    if dyn == 1:
        tags.append("limit_and_market")
    if bm:
        tags.append("brownian")
    tags.append({0: "pnl", 1: "quadratic"}[rew])
    if precise:
        tags.append("precise")
    return tags


def kernel_name(key, precise, lanes, lam32=False):
    """The name a rocprofv3 kernel-stats row carries for that kernel: step_kernel<Variant<tags...>, STREAM, MIRROR>."""
    # non-temporal loads once the lines a launch touches no longer fit the Infinity Cache: 16-byte rows beyond 300 MB DISTINCT (the state
    # is updated in place), other row widths beyond 640 MB moved (mbt_env.hip: tune_for_size)
    w = WORKLOADS[key]
    distinct = lanes * 4 * (w["dim"] + w["act"] + 1 + remainder_columns(key, precise, lam32))
    stream = distinct > (300 << 20) if w["dim"] == 4 else lanes * moved_bytes(key, precise, lam32) > (640 << 20)
    shape = ", ".join("mbt::shape::" + tag for tag in kernel_shape(key, precise, lam32))
    # (the mirror instantiation - the last argument - serves small batches over the host API only)
    return f"mbt::step_kernel<mbt::Variant<{shape}>, {'true' if stream else 'false'}, false>"


def build_env(n, offset, device, workload="cfg1", precise=False, lam32=False):
    """One shard of the trajectory axis of a BASELINE workload, through the public plugin API."""
    from mbt_gym_amd.gym.ModelDynamics import LimitAndMarketOrderModelDynamics, LimitOrderModelDynamics
    from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
    from mbt_gym_amd.rewards.RewardFunctions import CjMmCriterion, PnL, RunningInventoryPenalty
    from mbt_gym_amd.stochastic_processes.arrival_models import HawkesArrivalModel, PoissonArrivalModel
    from mbt_gym_amd.stochastic_processes.fill_probability_models import ExponentialFillFunction
    from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel, OuMidpriceModel

    dt = 1.0 / N_STEPS
    if workload == "cfg3":
        midprice = OuMidpriceModel(mean_reversion_level=100.0, mean_reversion_speed=0.01, volatility=2.0, initial_price=100.0, terminal_time=1.0,
                                   step_size=dt, num_trajectories=n)
        arrivals = HawkesArrivalModel(baseline_arrival_rate=np.array([[10.0, 10.0]]), step_size=dt, jump_size=40.0, mean_reversion_speed=60.0,
                                      terminal_time=1.0, num_trajectories=n)
    else:
        midprice = BrownianMotionMidpriceModel(volatility=2.0, initial_price=100, terminal_time=1.0, step_size=dt, num_trajectories=n)
        arrivals = PoissonArrivalModel(intensity=np.array([140.0, 140.0]), step_size=dt, num_trajectories=n)
    fills = ExponentialFillFunction(fill_exponent=1.5, step_size=dt, num_trajectories=n)
    dynamics_class = LimitAndMarketOrderModelDynamics if workload == "cfg4" else LimitOrderModelDynamics
    dynamics = dynamics_class(midprice_model=midprice, arrival_model=arrivals, fill_probability_model=fills, num_trajectories=n)
    reward = {"cfg2_cjmm": lambda: CjMmCriterion(0.01, 0.001, terminal_time=1.0), "cfg2_running": lambda: RunningInventoryPenalty(0.01, 0.001)}.get(workload, PnL)()
    env = TradingEnvironment(
        terminal_time=1.0, n_steps=N_STEPS, model_dynamics=dynamics, reward_function=reward, initial_inventory=10 if workload == "cfg4" else 0,
        max_inventory=100 if workload.startswith("cfg2") else N_STEPS, seed=SEED + (360 if workload.startswith("cfg2") else 0), num_trajectories=n,
        normalise_action_space=False, normalise_observation_space=False, device=device, trajectory_offset=offset, precise_state=precise,
        hawkes_float32_intensities=lam32,
    )
    if workload == "cfg4":
        # (bid depth, ask depth, market buy, market sell): the flags ~ Bernoulli(0.01), drawn ON THE DEVICE straight into the
        # library's action buffer (zero copy) - keyed on the global lane id, so the action of a lane does not depend on the sharding
        import torch

        view = torch.as_tensor(env.action_device, device=f"cuda:{device}")
        lane = torch.arange(offset, offset + n, device=f"cuda:{device}", dtype=torch.int64)

        def uniform(column):  # a counter-based draw per (global lane, column): an integer hash (wrapping int64 arithmetic) -> [0, 1)
            mask = (1 << 62) - 1
            x = (lane * 2 + column + 0x1E3779B97F4A7C15) & mask
            x = ((x ^ (x >> 30)) * 0x1CE4E5B9 + 0x133111EB) & mask
            x = ((x ^ (x >> 27)) * 0x2545F491) & mask
            x = x ^ (x >> 31)
            return (x & ((1 << 24) - 1)).to(torch.float32) / float(1 << 24)

        view[:, 0] = QUOTE[0]
        view[:, 1] = QUOTE[1]
        view[:, 2] = (uniform(0) < 0.01).to(torch.float32)
        view[:, 3] = (uniform(1) < 0.01).to(torch.float32)
        torch.cuda.synchronize(device)
    else:
        env.set_action_host(np.tile(np.array([QUOTE], dtype=np.float32), (n, 1)))
    if TRACED_LAUNCH_GATE:
        env.set_launch_gate(TRACED_LAUNCH_GATE)
    env.reset_device()
    return env


# Under rocprofv3 the host needs ~11 us per traced launch - more than the 7 us kernels here take - so the queue runs dry and
# a kernel that starts on an idle chip runs 0.5-1.8 us LONGER than in the untraced run the profile is meant to describe
# (profiles/r03_bench_kernel_trace_hist*.txt).  When a tracer is attached the launches are therefore enqueued in bursts behind
# a gate kernel (include/mbt_env.h: mbt_env_set_launch_gate) and run back to back, as they do untraced; the line says so.
TRACED = "ROCP_TOOL_LIBRARIES" in os.environ  # what rocprofv3 sets for the process it launches
# (bursts of 256: under the tracer the host blocks in a launch once ~550-1000 dispatches are outstanding - a burst of 1024 never
# reached its gate_open and every gate ran into its time-out, profiles/r04_experiments.txt)
TRACED_LAUNCH_GATE = int(os.environ.get("MBT_BENCH_GATE", "256" if TRACED else "0") or 0)  # MBT_BENCH_GATE=0: never


def timed_steps(env, lib, k, sync_all):
    """Exactly k env.step() launches between two synchronisation points.  Returns (wall seconds, HIP-event seconds)."""
    import ctypes as C

    from mbt_gym_amd import _native

    ms = C.c_float(0)
    sync_all()
    _native.check(lib.mbt_env_timer_begin(env._handle))  # HIP event on the kernel's stream
    t0 = time.perf_counter()
    steps, episodes = env.step_many_device(k, auto_reset=True)
    _native.check(lib.mbt_env_timer_stop(env._handle))  # records the closing event, no wait
    sync_all(barrier=False)
    wall = time.perf_counter() - t0
    _native.check(lib.mbt_env_timer_elapsed(env._handle, C.byref(ms)))  # device time between the two events (outside the wall-clock bracket)
    assert steps == k
    return wall, ms.value / 1e3, episodes


def pmc_traffic(n):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, the
    newest profiles/r*_pmc_summary.json, produced by tools/pmc_summary.py for this workload at 2^20 lanes); None for
    other sizes.  Counters cannot be read from inside the benchmark process, so this is the profiled value."""
    paths = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")) if re.fullmatch(r"r\d+_pmc_summary\.json", os.path.basename(p)))
    if n != LANES_PER_GPU or not paths:
        return None
    for name, row in json.load(open(paths[-1])).items():
        if "step_kernel" in name:
            return row["hbm_bytes_per_launch"]
    return None


def rocprof_reference():
    """{kernel name: average duration (ns)} from the newest committed `rocprofv3 --kernel-trace --stats` summary of THIS command
    (profiles/rNN_bench_kernel_stats.csv, tools/refresh_profiles.sh) and the file's name - so that the line can say, per
    kernel, what the tracked profile gives next to what the HIP events of this run give, and take the LOWER fraction."""
    import csv

    paths = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")) if re.fullmatch(r"r\d+_bench_kernel_stats\.csv", os.path.basename(p)))
    if not paths:
        return {}, None
    rows = {}
    with open(paths[-1], newline="") as f:
        for row in csv.DictReader(f):
            rows[row["Name"]] = (float(row["AverageNs"]), int(row["Calls"]))
    return rows, os.path.relpath(paths[-1], ROOT)


def roofline_row(key, precise, lanes, launch_s, reference, lam32=False):
    """One kernel against the HBM roofline in CREDITED bytes: `frac_events` from this run's HIP events, `frac_rocprof` from the
    committed rocprofv3 summary (None when it has no row for this kernel), `frac` the lower of the two."""
    name = kernel_name(key, precise, lanes, lam32)
    credited = credited_bytes(key) * lanes
    frac_events = credited / launch_s / 1e9 / HBM_PEAK_GBPS
    hit = next((v for k, v in reference.items() if name in k), None)
    frac_rocprof = None if hit is None else credited / (hit[0] * 1e-9) / 1e9 / HBM_PEAK_GBPS
    frac = frac_events if frac_rocprof is None else min(frac_events, frac_rocprof)
    moved = moved_bytes(key, precise, lam32)
    return {"config": WORKLOADS[key]["label"], "tier": tier_label(key, precise, lam32),
            "lanes": lanes, "credited_bytes_per_env_step": credited_bytes(key), "moved_bytes_per_env_step": moved,
            "avg_launch_us": launch_s * 1e6, "avg_launch_us_rocprof": None if hit is None else hit[0] / 1e3,
            "frac_events": frac_events, "frac_rocprof": frac_rocprof, "frac": frac, "achieved": frac * HBM_PEAK_GBPS,
            "moved_GBps": moved * lanes / launch_s / 1e9,  # what the memory system actually carried (not a roofline credit)
            "env_steps_per_s_kernel": lanes / launch_s, "kernel": name}


def configs_block(lib, device, reference, steps_budget_s=0.1):
    """Every other BASELINE configuration's step kernel, and the contract tier (`precise_state`) of all four, measured on THIS box
    exactly like the headline: K launches in one library call, HIP events on the kernel's stream, about 0.05 s of warm-up and 0.1 s of timed launches each
    (parity-test cases, not bench lines - they never enter `value`).  cfg3 (Hawkes) has three rows: the default tier (exact
    intensities, 76 B moved), the float32-intensity opt-out (60 B, what SURVEY section 8d prices) and precise_state (92 B)."""
    import torch

    rows = []
    cases = [("cfg2_cjmm", False, False), ("cfg2_running", False, False), ("cfg3", False, False), ("cfg3", False, True), ("cfg4", False, False),
             ("cfg1", True, False), ("cfg2_cjmm", True, False), ("cfg3", True, False), ("cfg4", True, False)]
    for key, precise, lam32 in cases:
        lanes = WORKLOADS[key]["lanes"]
        try:
            env = build_env(lanes, 0, device, workload=key, precise=precise, lam32=lam32)

            def sync_all(barrier=True, env=env):
                env.synchronize()
                torch.cuda.synchronize()

            rough_us = moved_bytes(key, precise, lam32) * lanes / 5.5e6  # at ~5.5 TB/s: only sizes the launch counts
            steps = int(max(200, min(8000, steps_budget_s * 1e6 / rough_us)))
            env.step_many_device(int(max(200, 0.05e6 / rough_us)), auto_reset=True)  # ~50 ms of launches: clocks up, like the headline's prewarm
            _, event_s, _ = timed_steps(env, lib, steps, sync_all)
            env.close()
            row = roofline_row(key, precise, lanes, event_s / steps, reference, lam32)
            row["steps_timed"] = steps
            rows.append(row)
        except Exception as exc:  # noqa: BLE001 - one configuration failing must not take the line down
            rows.append({"config": WORKLOADS[key]["label"], "tier": tier_label(key, precise, lam32), "error": str(exc)})
    return rows


def _cpu_worker(args):
    """One process of the CPU baseline: `steps` steps of the benchmark workload on `lanes` lanes of the oracle."""
    lanes, seed, steps, start_at = args
    from oracle.mbt_oracle import NumpyProtocolNoise, OracleConfig, OracleEnv  # cpu_baseline leg: the oracle is the thing timed here

    cfg = OracleConfig(
        num_trajectories=lanes, n_steps=N_STEPS, terminal_time=1.0, midprice="bm", drift=0.0, volatility=2.0,
        initial_price=100.0, arrival="poisson", intensity=(140.0, 140.0), fill_exponent=1.5, dynamics="limit",
        reward="pnl", initial_inventory=0, max_inventory=N_STEPS, seed=seed,
        normalise_action_space=False, normalise_observation_space=False,
    )
    env = OracleEnv(cfg, NumpyProtocolNoise(seed))
    env.reset()
    action = np.tile(np.array([QUOTE], dtype=np.float64), (lanes, 1))
    env.step(action)  # warm
    if start_at is not None:
        while time.time() < start_at:  # the processes start together
            pass
    t0 = time.time()
    done = 0
    for _ in range(steps):
        env.step(action)
        done += 1
    return t0, time.time(), done


def cpu_baseline(single_budget_s=10.0, multi_steps=24, process_counts=(16, 32, 64, 128, 256)):
    """The reference's algorithm (oracle = NumPy restatement, bit-matched to the reference, float64, PCG64 noise like the
    reference) on this host, same model and N = 2^20 lanes: (i) one process / one core, like the reference itself;
    (ii) K processes x N/K lanes - the sharding the reference's MultiprocessTradingEnv intended
    (gym/MultiprocessTradingEnv.py:74-80) - for every K in `process_counts` the host has cores for; (iii) BASELINE.json
    configs[0], the reference's own CPU-runnable case (N = 1000, 200 steps), one core.  A bounded sample: about 10 s + ~5 s per K."""
    import multiprocessing as mp

    n = LANES_PER_GPU
    # (i) one core
    t0, steps = time.perf_counter(), 0
    from oracle.mbt_oracle import NumpyProtocolNoise, OracleConfig, OracleEnv

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
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < single_budget_s and steps < N_STEPS - 2:
        env.step(action)
        steps += 1
    single = n * steps / (time.perf_counter() - t0)
    del env
    # (ii) K processes
    cores = os.cpu_count() or 1
    multi = {}
    ctx = mp.get_context("spawn")
    for k in process_counts:
        if k > cores:
            continue
        lanes = n // k
        try:
            with ctx.Pool(k) as pool:
                start = time.time() + 2.0 + 0.02 * k
                spans = pool.map(_cpu_worker, [(lanes, SEED + i, multi_steps, start) for i in range(k)])
            span = max(e for _, e, _ in spans) - min(s for s, _, _ in spans)
            multi[k] = lanes * k * multi_steps / span
        except Exception as exc:  # noqa: BLE001 - a host that cannot spawn still reports the one-core figure
            multi[k] = f"failed: {exc}"
    best_k = max((k for k, v in multi.items() if isinstance(v, float)), key=lambda k: multi[k], default=None)
    # (iii) configs[0]: N = 1000, n_steps = 200, the AS agent's closed form on the host like the reference's notebook (NB1:68-99)
    from oracle.mbt_oracle import avellaneda_stoikov_action

    cfg0 = OracleConfig(num_trajectories=1000, n_steps=200, terminal_time=1.0, midprice="bm", drift=0.0, volatility=2.0, initial_price=100.0,
                        arrival="poisson", intensity=(140.0, 140.0), fill_exponent=1.5, dynamics="limit", reward="pnl", initial_inventory=0,
                        max_inventory=200, seed=SEED, normalise_action_space=False, normalise_observation_space=False)
    env0 = OracleEnv(cfg0, NumpyProtocolNoise(SEED))
    t0, episodes0 = time.perf_counter(), 0
    while time.perf_counter() - t0 < 1.0:
        obs0 = env0.reset()
        for _ in range(cfg0.n_steps):
            obs0, _, _ = env0.step(avellaneda_stoikov_action(cfg0, 0.1, obs0))
        episodes0 += 1
    cfg0_rate = 1000 * 200 * episodes0 / (time.perf_counter() - t0)
    out = {
        "value": single, "unit": "env-steps/s", "cores": 1, "kind": "port",
        "configs0": {"value": cfg0_rate, "unit": "env-steps/s", "cores": 1,
                     "sample": f"{episodes0} episodes of BASELINE.json configs[0] (N = 1000, n_steps = 200, AS agent on the host), the same oracle"},
        "sample": f"{steps} steps x {n} lanes of the same workload, oracle/mbt_oracle.py (NumPy float64, PCG64 noise), "
                  f"{cores} host cores present, 1 used",
    }
    if best_k is not None:
        out["multicore"] = {
            "value": multi[best_k], "unit": "env-steps/s", "processes": best_k, "cores": best_k, "host_cores_present": cores,
            "sample": f"{multi_steps} steps x {best_k} processes x {n // best_k} lanes (2^20 in total), the same oracle; "
                      f"tried K = {list(multi)}: " + ", ".join(f"{k}: {v:.3g}" if isinstance(v, float) else f"{k}: {v}" for k, v in multi.items()),
        }
    return out


def hbm_resident_measurement(lib, device, reference, steps=600, warmup=100, lanes=None):
    """The same kernel on 2^24 lanes: 738 MB of state + actions + rewards per launch, beyond the 256 MB Infinity Cache,
    so every byte comes from / goes to HBM.  Reported beside the headline (cache-resident) figure, never instead of it."""
    import torch

    lanes = HBM_RESIDENT_LANES if lanes is None else lanes
    env = build_env(lanes, 0, device)

    def sync_all(barrier=True):
        env.synchronize()
        torch.cuda.synchronize()

    try:
        env.step_many_device(warmup, auto_reset=True)
        wall, event_s, _ = timed_steps(env, lib, steps, sync_all)
    finally:
        env.close()
    launch_s = event_s / steps
    row = roofline_row("cfg1", False, lanes, launch_s, reference)
    return {
        "lanes": lanes, "steps": steps, "bytes_per_launch": BYTES_PER_ENV_STEP * lanes,
        "avg_launch_us": launch_s * 1e6, "avg_launch_us_rocprof": row["avg_launch_us_rocprof"], "achieved": row["achieved"], "frac": row["frac"],
        "frac_events": row["frac_events"], "frac_rocprof": row["frac_rocprof"], "kernel": row["kernel"],
        "env_steps_per_s": lanes * steps / wall,
        "note": "working set 738 MB per launch > 256 MB Infinity Cache: HBM-resident; the achievable copy rate of the chip "
                "is ~6.3 TB/s (0.79 of the spec peak)",
    }


def rollout_counters():
    """Shader-side counters of the returns-only fused rollout from the newest committed rocprofv3 --pmc summary
    (profiles/rNN_pmc_rollout.json, tools/pmc_rollout_summary.py: SQ_INSTS_VALU, SQ_WAVES, GRBM_GUI_ACTIVE ... one small group per
    pass) - counters cannot be read from inside the benchmark process - or None."""
    paths = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_rollout.json")) if re.fullmatch(r"r\d+_pmc_rollout\.json", os.path.basename(p)))
    if not paths:
        return None
    out = json.load(open(paths[-1]))
    out["source"] = os.path.relpath(paths[-1], ROOT)
    return out


def write_floor():
    """The write-only floors of the recording pattern (28 B per lane and step, time-major, one long-running kernel) from the newest
    committed tools/microbench/mb_floor output (profiles/rNN_floors.txt): {log2 lanes: best us per step over the store policies}."""
    paths = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r*_floors.txt")) if re.fullmatch(r"r\d+_floors\.txt", os.path.basename(p)))
    floors, lanes = {}, None
    for line in (open(paths[-1]) if paths else []):
        m = re.match(r"record\s+28 B written per lane and step, 2\^(\d+) lanes", line)
        if m:
            lanes = int(m.group(1))
            continue
        m = re.search(r"median\s+([0-9.]+) \(min\s+([0-9.]+)\) us/step", line)
        if lanes is not None and m and "written" in line:
            floors[lanes] = min(floors.get(lanes, 1e30), float(m.group(1)))
        elif not line.startswith("  "):
            lanes = None
    return floors, (os.path.relpath(paths[-1], ROOT) if paths else None)


def rollout_block(lib, device):
    """SURVEY section 8f row 1 / 8d "rollout mode", measured on THIS box with HIP events on the environment's stream: the fused
    rollout kernel (a whole 1000-step episode of BASELINE configs[1] per launch, the policy evaluated in the kernel).
    Returns only: ~0 B of HBM traffic per env-step - instruction-issue bound, so NO HBM fraction is quoted; the line gives env-steps/s
    and the vector-issue fraction the committed counters give for the same launch.  Recorded (generate_trajectory's layout, GT:11-15):
    28 B written per env-step, against the 8 TB/s line and against the write-only floor of the same store pattern."""
    import ctypes as C

    import torch

    from mbt_gym_amd import _native
    from mbt_gym_amd.agents.BaselineAgents import AvellanedaStoikovAgent, FixedSpreadAgent

    def timed(env, agent, episodes, **pointers):
        env.reset_device()
        env.rollout_device(agent, **pointers)  # warm (and the first touch of a recording's pages)
        env.synchronize()
        _native.check(lib.mbt_env_timer_begin(env._handle))
        for _ in range(episodes):
            env.reset_device()
            steps, done = env.rollout_device(agent, **pointers)
            assert done and steps == env.n_steps
        ms = C.c_float(0)
        _native.check(lib.mbt_env_timer_end(env._handle, C.byref(ms)))
        return ms.value / 1e3 / episodes

    out = {"what": "fused rollout kernel: one launch = one 1000-step episode of BASELINE configs[1], policy evaluated in the kernel (SURVEY 8f row 1)"}
    counters = rollout_counters()
    n = LANES_PER_GPU
    env = build_env(n, 0, device)
    try:
        for name, agent in (("avellaneda_stoikov_policy", AvellanedaStoikovAgent(0.1, env)), ("fixed_policy", FixedSpreadAgent(env, half_spread=QUOTE[0]))):
            t = timed(env, agent, 8)
            row = {"lanes": n, "env_steps_per_s": n * env.n_steps / t, "us_per_env_step_of_all_lanes": t / env.n_steps * 1e6, "ms_per_episode": t * 1e3,
                   "hbm_bytes_per_env_step": 0, "bound": "vector instruction issue (no HBM fraction applies)"}
            hit = (counters or {}).get(name)
            if hit is not None:
                row["valu_issue_fraction"] = hit["valu_issue_fraction"]
                row["valu_instructions_per_wave_and_step"] = hit["valu_instructions_per_wave_and_step"]
                row["counters"] = counters["source"]
            out["returns_only_" + name] = row
    finally:
        env.close()
    floors, floors_file = write_floor()
    for log2n in (18, 20):
        n = 1 << log2n
        env = build_env(n, 0, device)
        try:
            lanes = env.padded_lanes
            obs = torch.empty((env.n_steps + 1, lanes, 4), dtype=torch.float32, device=f"cuda:{device}")
            act = torch.empty((env.n_steps, lanes, 2), dtype=torch.float32, device=f"cuda:{device}")
            rew = torch.empty((env.n_steps, lanes), dtype=torch.float32, device=f"cuda:{device}")
            # kernel and write-only floor ALTERNATELY, in this process, against these buffers (the floor moves by +-15 % with where an
            # allocation lands - a figure from another process is a figure about another allocation): medians of five of each
            agent, pointers = AvellanedaStoikovAgent(0.1, env), dict(obs_ptr=obs.data_ptr(), act_ptr=act.data_ptr(), rew_ptr=rew.data_ptr())
            timed(env, agent, 1, **pointers)
            kernel_s, floor_s, ms = [], [], C.c_float(0)
            for _ in range(5):
                kernel_s.append(timed(env, agent, 1, **pointers))
                _native.check(lib.mbt_env_timer_begin(env._handle))
                _native.check(lib.mbt_env_record_floor_device(env._handle, env.n_steps, obs.data_ptr(), act.data_ptr(), rew.data_ptr()))
                _native.check(lib.mbt_env_timer_end(env._handle, C.byref(ms)))
                floor_s.append(ms.value / 1e3)
            env.reset_device()
            env.rollout_device(agent, **pointers)  # (the recording the mean return below is read from: the floor runs wrote over the last one)
            env.synchronize()
            t, floor_here = float(np.median(kernel_s)), float(np.median(floor_s))
            us = t / env.n_steps * 1e6
            row = {"lanes": n, "env_steps_per_s": n * env.n_steps / t, "us_per_env_step_of_all_lanes": us, "written_bytes_per_env_step": 28,
                   "write_GBps": 28.0 * n / us * 1e-3, "frac_of_8TBps": 28.0 * n / us * 1e-3 / HBM_PEAK_GBPS, "GB_per_episode": 28e-9 * n * env.n_steps,
                   "mean_return_of_the_recording": float(rew[:, :n].sum(dim=0).mean())}
            hit = (counters or {}).get(f"recorded_avellaneda_stoikov_2^{log2n}")
            if hit is not None and "wave_cycles_issuing" in hit:  # what the committed counters say the waves of this launch wait for
                row["counters"] = {"source": counters["source"], "wave_cycles_issuing": hit["wave_cycles_issuing"],
                                   "wave_cycles_stalled_at_issue": hit["wave_cycles_stalled_at_issue"], "wave_cycles_parked_on_waitcnt": hit["wave_cycles_parked_on_waitcnt"],
                                   "l2_write_request_stall_cycles_per_request": hit.get("l2_write_requests_stalled_cycles_per_request"),
                                   "dram_credit_stall_cycles_per_write_request": hit.get("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", 0.0) / max(hit.get("TCC_EA0_WRREQ_sum", 1.0), 1.0)}
            row["write_only_floor_us"] = floor_here / env.n_steps * 1e6
            row["floor_over_kernel"] = floor_here / t
            row["floor_source"] = ("mbt_env_record_floor_device: the rollout's own store pattern with no other work, timed alternately with the rollout in this "
                                   "process against the same trajectory buffers; medians of 5")
            row["us_per_step_min_max"] = {"kernel": [min(kernel_s) / env.n_steps * 1e6, max(kernel_s) / env.n_steps * 1e6],
                                          "floor": [min(floor_s) / env.n_steps * 1e6, max(floor_s) / env.n_steps * 1e6]}
            if log2n in floors:  # (the stand-alone micro-benchmark's figure of an earlier run, for comparison: another process, another allocation)
                row["write_only_floor_us_of_the_committed_microbenchmark"] = floors[log2n]
                row["microbenchmark_source"] = floors_file
            out[f"recorded_avellaneda_stoikov_2^{log2n}"] = row
            del obs, act, rew
        except Exception as exc:  # noqa: BLE001 - a device without room for the recording still reports the rest
            out[f"recorded_avellaneda_stoikov_2^{log2n}"] = {"error": str(exc)}
        finally:
            env.close()
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU), exactly what
    `python -m torch.distributed.run --nproc-per-node N` would set up, and pass rank 0's JSON line through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        deadline = time.time() + 1500
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is not None:
                    pending.remove(p)
                    rc = rc or code
            if rc != 0 or time.time() > deadline:
                break
            time.sleep(0.05)
    finally:
        for p in procs:  # a failed rank leaves the others waiting in a collective: stop exactly the processes started here
            if p.poll() is None:
                p.kill()
                rc = rc or 1
    return rc


def rccl_probe(rank, world, gpu, id_path):
    """`bench.py --probe rank world gpu path`: a DIAGNOSTIC, not part of the measured job - one rank of a throw-away job that
    makes a C-ABI RCCL communicator (the 128-byte id travels by file, like examples/sharded_returns.c) and all-reduces three
    doubles on an environment's stream.  Exit code 0 = the sum is right.  Start one per GPU by hand when a multi-GPU run
    misbehaves; the benchmark itself uses ONE communicator strategy (main: make_communicator)."""
    from mbt_gym_amd.distributed import RcclCommunicator

    def exchange(payload):
        if rank == 0:
            with open(id_path + ".tmp", "wb") as f:
                f.write(payload)
            os.replace(id_path + ".tmp", id_path)
            return payload
        deadline = time.time() + 60
        while not os.path.exists(id_path):
            if time.time() > deadline:
                raise TimeoutError("rank 0's RCCL id never appeared")
            time.sleep(0.02)
        with open(id_path, "rb") as f:
            return f.read()

    env = build_env(1024, rank * 1024, gpu)
    comm = RcclCommunicator(rank, world, gpu, exchange=exchange)
    sums = env.allreduce_return_sums(comm, [float(rank + 1), 0.0, 1.0])
    ok = sums[0] == world * (world + 1) / 2 and sums[2] == world and comm.count() == world
    print(f"rank {rank}: ranks seen {comm.count()}, sums {sums.tolist()} -> {'ok' if ok else 'WRONG'}", file=sys.stderr)
    env.close()
    comm.close()
    return 0 if ok else 4


class Watchdog:
    """A hard deadline around the steps of a multi-GPU run that can block inside native code (communicator creation, a
    collective): nothing inside a process can cancel those, so when the deadline passes the process says which step hung
    and exits - the launcher then stops the other ranks - instead of hanging the run for the caller's whole time limit."""

    def __init__(self, seconds, what, rank):
        import threading

        self.what, self.rank, self.seconds = what, rank, seconds
        self.timer = threading.Timer(seconds, self.expire)
        self.timer.daemon = True

    def expire(self):
        print(f"[rank {self.rank}] bench.py: '{self.what}' did not finish within {self.seconds:.0f} s - giving up (exit 17)", file=sys.stderr, flush=True)
        os._exit(17)

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def gpu_placement(gpu, pin):
    """Where this rank's GPU sits - PCI bus id, NUMA node - and, with `pin`, the host threads kept on that node's cores (the launch
    path is one host thread per GPU; a cross-socket hop adds to every launch).  Best effort: ({facts}, description)."""
    facts = {"device_ordinal": gpu, "pci_bus_id": None, "numa_node": None, "cores_pinned": None, "device_name": None}
    try:
        import torch

        props = torch.cuda.get_device_properties(gpu)
        facts["device_name"] = getattr(props, "gcnArchName", None) or props.name
        bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        facts["pci_bus_id"] = bdf
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        facts["numa_node"] = node
        if not pin:
            return facts, "not pinned"
        if node < 0:
            return facts, f"GPU {gpu} ({bdf}): no NUMA affinity reported"
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = cpus & os.sched_getaffinity(0)
        if not allowed:
            return facts, f"GPU {gpu} ({bdf}): NUMA node {node} has no core this process may use"
        os.sched_setaffinity(0, allowed)
        facts["cores_pinned"] = len(allowed)
        return facts, f"GPU {gpu} ({bdf}): NUMA node {node}, {len(allowed)} cores"
    except Exception as exc:  # noqa: BLE001 - affinity is an optimisation, never a requirement
        return facts, f"not pinned: {exc}"


def gpu_cfg0_figures(device):
    """BASELINE.json configs[0] (N = 1000, n_steps = 200) on the GPU, the two ways a caller runs it: the reference's loop
    (host agent + env.step(ndarray): latency-bound at this size) and the fused rollout (the agent's closed form on the device)."""
    from mbt_gym_amd.agents.BaselineAgents import AvellanedaStoikovAgent
    from mbt_gym_amd.gym.ModelDynamics import LimitOrderModelDynamics
    from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
    from mbt_gym_amd.stochastic_processes.arrival_models import PoissonArrivalModel
    from mbt_gym_amd.stochastic_processes.fill_probability_models import ExponentialFillFunction
    from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel

    n, n_steps = 1000, 200
    dt = 1.0 / n_steps

    def build(resident=False):
        dynamics = LimitOrderModelDynamics(
            midprice_model=BrownianMotionMidpriceModel(volatility=2.0, initial_price=100, terminal_time=1.0, step_size=dt, num_trajectories=n),
            arrival_model=PoissonArrivalModel(intensity=np.array([140.0, 140.0]), step_size=dt, num_trajectories=n),
            fill_probability_model=ExponentialFillFunction(fill_exponent=1.5, step_size=dt, num_trajectories=n), num_trajectories=n)
        return TradingEnvironment(terminal_time=1.0, n_steps=n_steps, model_dynamics=dynamics, initial_inventory=0, max_inventory=200, seed=SEED,
                                  num_trajectories=n, normalise_action_space=False, normalise_observation_space=False, device=device, resident_step=resident)

    env = build()
    agent = AvellanedaStoikovAgent(risk_aversion=0.1, env=env)

    def loop_episode():
        obs = env.reset()
        while True:
            obs, _, done, _ = env.step(agent.get_action(obs))
            if done[0]:
                return

    def fused_episode():
        env.reset_device()
        env.rollout(agent, record=False)
        env.synchronize()

    out = {}
    for name, episode in (("host_api_loop", loop_episode), ("fused_rollout", fused_episode)):
        episode()
        t0, count = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.5:
            episode()
            count += 1
        out[name + "_env_steps_per_s"] = n * n_steps * count / (time.perf_counter() - t0)
    # env.step(ndarray) on its own (a fixed action: no agent), and through the SB3 VecEnv adapter with its auto-reset (SBE:28-37)
    from mbt_gym_amd.gym.StableBaselinesTradingEnvironment import StableBaselinesTradingEnvironment

    action = np.tile(np.array([QUOTE], dtype=np.float32), (n, 1))
    vec_env = StableBaselinesTradingEnvironment(trading_env=env)
    for name, reset, step in (("env_step_us", env.reset, env.step), ("sb3_vec_env_step_us", vec_env.reset, vec_env.step)):
        reset()
        for _ in range(50):
            step(action)
        reset()
        t0 = time.perf_counter()
        for _ in range(150):
            step(action)
        out[name] = (time.perf_counter() - t0) / 150 * 1e6
    env.close()
    # the same calls with the opt-in resident kernel (TradingEnvironment(resident_step=True): env.step() rings the doorbell of a kernel that stays on the
    # device instead of launching one; it slows kernels on OTHER streams by 20-27 %, profiles/r05_resident_step.txt - hence opt-in)
    env = build(resident=True)
    agent = AvellanedaStoikovAgent(risk_aversion=0.1, env=env)
    vec_env = StableBaselinesTradingEnvironment(trading_env=env)
    resident = {}
    loop_episode()
    t0, count = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.5:
        loop_episode()
        count += 1
    resident["host_api_loop_env_steps_per_s"] = n * n_steps * count / (time.perf_counter() - t0)
    for name, reset, step in (("env_step_us", env.reset, env.step), ("sb3_vec_env_step_us", vec_env.reset, vec_env.step)):
        reset()
        for _ in range(50):
            step(action)
        reset()
        t0 = time.perf_counter()
        for _ in range(150):
            step(action)
        resident[name] = (time.perf_counter() - t0) / 150 * 1e6
    env.close()
    out["resident_step_opt_in"] = resident
    out["note"] = "N = 1000 x 200 steps: a launch-latency regime (140 KB of state), not a bandwidth one; reported beside the CPU port's configs0 figure"
    return out


def main():
    if len(sys.argv) == 6 and sys.argv[1] == "--probe":
        sys.exit(rccl_probe(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--lanes", type=int, default=LANES_PER_GPU, help="trajectories per GPU (default 2^20)")
    ap.add_argument("--prewarm-steps", type=int, default=-1, help="untimed clock warm-up before --warmup; -1 = ~55 ms worth, a fixed count for the size (default), 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration block (cfg2/3/4 and the precise_state tier; N = 1 only)")
    ap.add_argument("--cfg4-total-lanes", type=int, default=1 << 24,
                    help="N > 1: BASELINE.json configs[4] (limit + market orders) with this many lanes in TOTAL, sharded over the ranks "
                         "(strong scaling: 2^21 per GPU at N = 8), reported as an extra block of the line; 0 = skip")
    ap.add_argument("--cfg4-steps", type=int, default=1100, help="timed steps of the cfg4 block (after steps // 4 of warm-up): 1100 puts one episode end - "
                    "reduction, 24-byte all-reduce, reset, all enqueued in-stream - inside the timed region")
    ap.add_argument("--no-hbm-resident", action="store_true", help="skip the extra 2^24-lane measurement (N = 1 only)")
    ap.add_argument("--no-rollout", action="store_true", help="skip the fused-rollout block (N = 1 only)")
    ap.add_argument("--no-device-loop", action="store_true", help="skip the zero-copy consumer-loop block (torch policy on the device, eager vs HIP graph; N = 1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the launcher-side barrier (nccl = RCCL; gloo only for testing)")
    ap.add_argument("--single-device", action="store_true", help="testing: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--force-distributed", action="store_true", help="testing: take the multi-rank code path (process group, C-ABI communicator, collective check) even with one rank")
    ap.add_argument("--per-rank-hbm-lanes", type=int, default=HBM_RESIDENT_LANES,
                    help="N > 1: every rank also measures the step kernel at this many lanes (2^24: beyond the Infinity Cache) for the per-rank block of the line; 0 = skip")
    ap.add_argument("--comm-timeout", type=float, default=180.0, help="hard deadline (s) for creating the RCCL communicator and for each collective check")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    import torch

    from mbt_gym_amd import _native
    from mbt_gym_amd.build import build_native
    from mbt_gym_amd.distributed import RcclCommunicator, allreduce_return_sums, return_statistics, shard_bounds

    if rank == 0:
        build_native()  # no-op unless the library is missing or was built from other sources
    dist = None
    multi = world > 1 or args.force_distributed  # the multi-rank code path (a world of one takes it only when asked to: tests)
    if multi:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # (no launcher: a world of one, and nobody else has to know the port - any free one)
            with socket.socket() as probe:
                probe.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(probe.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    visible = torch.cuda.device_count()
    if world > 1 and not args.single_device and visible < world and args.backend == "nccl":
        raise SystemExit(f"--gpus {world} needs {world} visible GPUs, {visible} present (functional test on one GPU: --backend gloo --single-device)")
    # one process per GPU; a launcher that narrows each rank's visible devices leaves fewer ordinals than ranks
    gpu = 0 if args.single_device else local_rank % max(1, visible)
    torch.cuda.set_device(gpu)
    placement, affinity = gpu_placement(gpu, pin=multi and os.environ.get("MBT_BENCH_PIN", "1") != "0")
    seconds = {"rendezvous_and_first_barrier": None, "comm_init_rank": None, "first_collective": None}  # where this rank's set-up time went
    if multi:
        # (generous: on a fresh box the ranks finish their first `import torch` minutes apart, and the early ones wait here)
        with Watchdog(max(args.comm_timeout, 480.0), "torch.distributed rendezvous + first barrier", rank):
            t_phase = time.perf_counter()
            if args.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", gpu))
            else:
                dist.init_process_group(backend=args.backend)
            dist.barrier()  # every rank waits for rank 0's build check before it loads the library
            seconds["rendezvous_and_first_barrier"] = time.perf_counter() - t_phase
    lib = _native.load_library()

    n = args.lanes
    offset, count = shard_bounds(n * world, rank, world)
    assert count == n and offset == rank * n
    env = build_env(n, offset, gpu)

    # The one collective of the path, ONE strategy: an RCCL communicator made through the C ABI (mbt_comm_init_rank; the
    # 128-byte id travels over the launcher's process group) and attached to the environment, so that episode ends enqueue
    # their 24-byte all-reduce on the environment's stream.  Creation runs under a hard deadline.  If it FAILS (an error, not
    # a hang) every rank agrees to take torch.distributed for the return sums instead - on every rank or on none.  With the
    # gloo backend (single-device testing: RCCL refuses two ranks on one GPU) torch.distributed is the transport from the start.
    comm, transport, ranks_seen = None, "none (1 rank)", 1
    tdev = torch.device("cuda", gpu) if args.backend == "nccl" else torch.device("cpu")
    if multi:
        transport = f"torch.distributed/{args.backend}"
        ranks_seen = dist.get_world_size()
        if args.backend == "nccl":
            ok = True
            with Watchdog(args.comm_timeout, "mbt_comm_init_rank (C-ABI RCCL communicator)", rank):
                try:
                    t_phase = time.perf_counter()
                    comm = RcclCommunicator(rank, world, gpu)
                    seconds["comm_init_rank"] = time.perf_counter() - t_phase
                    env.set_communicator(comm)
                    t_phase = time.perf_counter()  # the communicator's FIRST collective (RCCL sets its channels up lazily: seconds, on some fabrics)
                    env.allreduce_return_sums(comm, [1.0, 0.0, 1.0])
                    seconds["first_collective"] = time.perf_counter() - t_phase
                except Exception as exc:  # noqa: BLE001
                    print(f"[rank {rank}] C-ABI RCCL communicator unavailable ({exc}); using torch.distributed", file=sys.stderr)
                    ok = False
                flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=tdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if flag.item() < 1.0:
                if comm is not None:
                    env.set_communicator(None)
                    comm.close()
                comm = None
            else:
                transport = "RCCL via mbt_env_set_communicator (C ABI), enqueued on the environment's stream"
                ranks_seen = comm.count()  # ncclCommCount: what RCCL itself says

    def sync_all(barrier=True):
        env.synchronize()
        torch.cuda.synchronize()
        if barrier and dist is not None:
            dist.barrier()

    def drain_log():
        out = []
        while True:
            sums = env.episode_log_pop(wait=True)
            if sums is None:
                return out
            out.append(sums if (comm is not None or not multi) else allreduce_return_sums(sums, device=tdev))

    # clocks up (untimed, not part of --warmup), then the warm-up the caller asked for
    prewarm, prewarm_target = 0, args.prewarm_steps if args.prewarm_steps >= 0 else default_prewarm_steps(n)
    with Watchdog(max(args.comm_timeout, 600.0), "warm-up steps (including the all-reduces of the episodes that end in them)", rank):
        t_warm = time.perf_counter()
        with phase("clock warm-up (untimed) + --warmup steps"):
            while prewarm < prewarm_target:
                prewarm += env.step_many_device(min(256, prewarm_target - prewarm), auto_reset=True)[0]
                env.synchronize()
            if args.warmup > 0:
                env.step_many_device(args.warmup, auto_reset=True)
            sync_all()
        warm_episodes = drain_log()
        seconds["warm_up_steps_and_their_collectives"] = time.perf_counter() - t_warm

        with phase(f"timed region: {args.steps} steps"):
            wall, event_s, episodes = timed_steps(env, lib, args.steps, sync_all)
        episode_returns = drain_log()  # (the log keeps the newest 16 episodes)
    event_s_own = event_s  # (this rank's own; `event_s` becomes the slowest rank's below)
    launch_us_min = launch_us_max = event_s / args.steps * 1e6  # per-rank mean launch-to-launch time: a straggler shows here
    if dist is not None:
        dist.barrier()
        t = torch.tensor([wall, event_s, -event_s], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        launch_us_max, launch_us_min = float(t[1]) / args.steps * 1e6, -float(t[2]) / args.steps * 1e6
        wall, event_s = float(t[0]), float(t[1])

    # mean return of the last finished episode over ALL shards (or of the partial episode if none finished)
    if episode_returns:
        sums = episode_returns[-1]
    elif warm_episodes:
        sums = warm_episodes[-1]
    else:
        local = env.episode_return_sums()
        sums = env.allreduce_return_sums(comm, local) if comm is not None else allreduce_return_sums(local, device=tdev)

    # the collective on its own (N > 1): a known-answer all-reduce through the same communicator - every rank contributes
    # [rank + 1, 0, 1], the sum must be [N (N + 1) / 2, 0, N] - and its latency (blocking form: copy in, all-reduce, copy out, wait)
    collective = None
    if multi:
        with Watchdog(args.comm_timeout, "known-answer all-reduce", rank):
            reduce_once = (lambda v: env.allreduce_return_sums(comm, v)) if comm is not None else (lambda v: allreduce_return_sums(v, device=tdev))
            t_phase = time.perf_counter()
            got = reduce_once([float(rank + 1), 0.0, 1.0])
            seconds["known_answer_collective"] = time.perf_counter() - t_phase
            correct = bool(got[0] == world * (world + 1) / 2 and got[2] == world)
            sync_all()
            reps = 50
            t0 = time.perf_counter()
            for _ in range(reps):
                reduce_once([1.0, 0.0, 1.0])
            per_call = (time.perf_counter() - t0) / reps
            # every rank must have finished - and all-reduced - the SAME number of episodes (the step counts are functions of the
            # arguments alone for exactly this reason): the line says what each end of the range saw
            finished = float(len(warm_episodes) + len(episode_returns))
            span = torch.tensor([finished, -finished], dtype=torch.float64, device=tdev)
            dist.all_reduce(span, op=dist.ReduceOp.MAX)
        collective = {"what": "24-byte all-reduce of [sum R, sum R^2, lanes], blocking form (H2D 24 B + all-reduce + D2H 24 B + wait)",
                      "us": per_call * 1e6, "known_answer_ok": correct, "per_episode_share_of_stepping": per_call / (N_STEPS * wall / args.steps),
                      "episodes_all_reduced_before_the_timed_region": len(warm_episodes),
                      "episodes_in_the_log_per_rank": {"min": int(-span[1].item()), "max": int(span[0].item())}}

    # BASELINE.json configs[4]: limit + market orders, --cfg4-total-lanes in TOTAL sharded over the ranks (strong scaling), the
    # episode-return all-reduce on the same communicator; timed like the headline (barrier, K launches in one call, max over ranks)
    cfg4 = None
    if multi and args.cfg4_total_lanes > 0:
        with Watchdog(max(args.comm_timeout, 300.0), "cfg4 (limit + market) sharded measurement", rank):
            off4, n4 = shard_bounds(args.cfg4_total_lanes, rank, world)
            env4, problem = None, ""
            try:  # set-up can fail on ONE rank (memory, a device fault): every rank learns of it before anyone enters a collective
                env4 = build_env(n4, off4, gpu, workload="cfg4")
                if comm is not None:
                    env4.set_communicator(comm)
                env4.step_many_device(max(50, args.cfg4_steps // 4), auto_reset=True)
                env4.synchronize()
            except Exception as exc:  # noqa: BLE001
                problem = f"rank {rank}: {exc}"
                print(f"[rank {rank}] cfg4 block skipped: {exc}", file=sys.stderr)
            ready = torch.tensor([0.0 if problem else 1.0], dtype=torch.float64, device=tdev)
            dist.all_reduce(ready, op=dist.ReduceOp.MIN)
            if ready.item() < 1.0:
                cfg4 = {"error": problem or "another rank could not set the workload up (see its stderr)"}
                if env4 is not None:
                    env4.close()
            else:
                def sync4(barrier=True):
                    env4.synchronize()
                    torch.cuda.synchronize()
                    if barrier and dist is not None:
                        dist.barrier()

                wall4, event4, _ = timed_steps(env4, lib, args.cfg4_steps, sync4)
                while env4.episode_log_pop(wait=True) is not None:
                    pass
                t4 = torch.tensor([wall4, event4, -event4], dtype=torch.float64, device=tdev)
                dist.all_reduce(t4, op=dist.ReduceOp.MAX)
                env4.close()
                launch4 = float(t4[1]) / args.cfg4_steps
                cfg4 = {"workload": WORKLOADS["cfg4"]["label"] + ", BASELINE.json configs[4]", "scaling": "strong", "num_trajectories_total": args.cfg4_total_lanes,
                        "num_trajectories_per_gpu": n4, "steps": args.cfg4_steps, "value": args.cfg4_total_lanes * args.cfg4_steps / float(t4[0]), "unit": "env-steps/s",
                        "ms_per_step": float(t4[0]) / args.cfg4_steps * 1e3, "credited_bytes_per_env_step": credited_bytes("cfg4"),
                        "avg_launch_us_slowest_rank": launch4 * 1e6, "avg_launch_us_fastest_rank": -float(t4[2]) / args.cfg4_steps * 1e6,
                        "frac_per_gpu": credited_bytes("cfg4") * n4 / launch4 / 1e9 / HBM_PEAK_GBPS, "kernel": kernel_name("cfg4", False, n4)}

    # One line has to explain a scaling curve that bends (the 8-GPU run is unattended): what every rank used and where its time went -
    # device, PCI bus id, NUMA node, what RCCL says the communicator spans, rendezvous / communicator / first-collective seconds, its own
    # launch-to-launch time in the timed region, and the step kernel's HBM-resident rate on ITS device (2^24 lanes: a slow HBM stack,
    # a throttled or shared device shows here and nowhere else).
    ranks_block = None
    if multi:
        mine = dict(placement, rank=rank, local_rank=local_rank, host=socket.gethostname(), rccl_comm_count=(comm.count() if comm is not None else None),
                    return_allreduce=transport, seconds=seconds, avg_launch_us=event_s_own / args.steps * 1e6, visible_devices=visible,
                    hip_visible_devices=os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES"))
        if args.per_rank_hbm_lanes > 0:
            with Watchdog(max(args.comm_timeout, 300.0), "per-rank HBM-resident measurement", rank):
                try:
                    big = hbm_resident_measurement(lib, gpu, {}, steps=200, warmup=50, lanes=args.per_rank_hbm_lanes)
                    mine["hbm_resident"] = {"lanes": big["lanes"], "avg_launch_us": big["avg_launch_us"], "frac": big["frac_events"], "GBps": big["achieved"],
                                            "concurrent_with_the_other_ranks": True}
                except Exception as exc:  # noqa: BLE001 - one rank short of memory must not cost the line
                    mine["hbm_resident"] = {"error": f"{type(exc).__name__}: {exc}"}
        with Watchdog(args.comm_timeout, "gathering the per-rank blocks", rank):
            gathered = [None] * dist.get_world_size()
            dist.all_gather_object(gathered, mine)
        ranks_block = gathered
    if rank == 0:
        reference, reference_file = rocprof_reference()
        headline = roofline_row("cfg1", False, n, event_s / args.steps, reference)
        total_lanes = n * world
        value = total_lanes * args.steps / wall
        launch_s = event_s / args.steps  # includes the reset / reduction launches of finished episodes (2 per 1000 steps)
        achieved = BYTES_PER_ENV_STEP * n / launch_s / 1e9
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
                "return_allreduce": transport, "rccl_ranks_seen": ranks_seen, "host_affinity_rank0": affinity,
                "episodes_finished_in_timed_region": episodes, "prewarm_steps": prewarm,
                "kernel_tuning": "this kernel (Brownian / Poisson / limit / PnL) runs capped at 5 workgroups per CU from 2^20 lanes "
                                 "up (mbt_env.hip: tune_for_size): measured +2.5 % for it, a loss for every heavier kernel, which keep full occupancy",
            },
            "roofline": {
                "bound": "hbm", "achieved": headline["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": headline["frac"],
                "frac_events": achieved / HBM_PEAK_GBPS, "frac_rocprof": headline["frac_rocprof"], "rocprof_summary": reference_file,
                "avg_launch_us_rocprof": headline["avg_launch_us_rocprof"], "kernel": headline["kernel"],
                "avg_launch_us_per_rank": {"min": launch_us_min, "max": launch_us_max},
                "traffic": pmc_traffic(n), "bytes_per_env_step": BYTES_PER_ENV_STEP, "avg_launch_us": launch_s * 1e6,
                "regime": "Infinity-Cache resident" if n * BYTES_PER_ENV_STEP < (200 << 20) else "HBM resident",
                "note": "`frac` = the LOWER of `frac_events` (algorithmic bytes, 44 B/env-step x lanes per launch, / mean launch-to-launch "
                        "time from HIP events on the kernel's stream over the timed region of THIS run) and `frac_rocprof` (the same bytes / "
                        "the kernel's average duration in the committed rocprofv3 --kernel-trace --stats summary of this command, "
                        "`rocprof_summary`); `achieved` = frac x peak.  At 2^20 lanes the 46 MB a launch touches stay in the 256 MB "
                        "Infinity Cache between launches, so this is NOT an HBM rate (it can exceed the ~6.3 TB/s HBM sustains); "
                        "`hbm_resident` is the same kernel where every byte crosses HBM.  Cross-check: per-dispatch durations under "
                        "rocprofv3 agree within 1-4 % in runs where the traced launches stay back to back; the tracer raises the host's "
                        "cost per launch to about the kernel's duration, and in runs where the queue runs dry kernels start on an idle "
                        "chip and take 0.5-1.8 us longer (profiles/r03_bench_kernel_trace_hist*.txt).",
            },
            "event_env_steps_per_s": total_lanes * args.steps / event_s,
            "mean_episode_return": return_statistics(sums)[0],
        }
        if TRACED_LAUNCH_GATE:
            out["config"]["launch_gate"] = (f"launches enqueued in bursts of {TRACED_LAUNCH_GATE} behind a gate kernel (a tracer is attached: the host's "
                                            "~11 us per traced launch would let the queue run dry); `value` is not a benchmark figure in this mode")
        if collective is not None:
            out["collective"] = collective
        if cfg4 is not None:
            out["cfg4_sharded"] = cfg4
        if ranks_block is not None:
            out["ranks"] = ranks_block
    env.close()
    if rank == 0:
        if world == 1 and not args.no_hbm_resident:
            try:
                with phase("block: hbm_resident (2^24 lanes)"):
                    out["roofline"]["hbm_resident"] = hbm_resident_measurement(lib, gpu, reference)
            except Exception as exc:  # noqa: BLE001 - e.g. a smaller device: the headline stands on its own
                out["roofline"]["hbm_resident"] = {"error": str(exc)}
        if world == 1 and not args.no_configs:
            with phase("block: roofline.configs (cfg2 / cfg3 / cfg4, precise_state)"):
                out["roofline"]["configs"] = configs_block(lib, gpu, reference)
        if world == 1 and not args.no_rollout:
            try:
                with phase("block: rollout (returns only, recorded + write-only floor)"):
                    out["rollout"] = rollout_block(lib, gpu)
            except Exception as exc:  # noqa: BLE001
                out["rollout"] = {"error": str(exc)}
        if world == 1 and not args.no_device_loop and TRACED:
            # (under a tracer the block would launch the headline's kernel at N = 1000 and 2^16 lanes and fold those launches into the
            # kernel's row of the summary `roofline.frac_rocprof` is read from)
            out["device_policy_loop"] = {"skipped": "a tracer is attached: the block's small-batch launches of the headline kernel would enter its row of the kernel statistics"}
        elif world == 1 and not args.no_device_loop:
            # a policy that lives on the device writes `action_device`, the environment steps on the same stream - one Python
            # call per step (host clock) against a torch.cuda.graph of [policy, mbt_env_step_device_captured] x 50 (device clock)
            try:
                from tools.bench_device_loop import device_policy_loop_block

                with phase("block: device_policy_loop (eager vs HIP graph)"):
                    out["device_policy_loop"] = device_policy_loop_block(gpu)
            except Exception as exc:  # noqa: BLE001
                out["device_policy_loop"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline:
            with phase("block: cpu_baseline (NumPy port on the host cores; cfg0 on the GPU)"):
                out["cpu_baseline"] = cpu_baseline()
            try:
                out["cpu_baseline"]["configs0"]["gpu"] = gpu_cfg0_figures(gpu)
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"]["configs0"]["gpu"] = {"error": str(exc)}
    if comm is not None:
        comm.close()
    if dist is not None:
        with Watchdog(args.comm_timeout, "final barrier", rank):
            dist.barrier()
            dist.destroy_process_group()
    if rank == 0:  # the very last thing written to stdout (RCCL prints its version banner there when communicators are made)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
