#!/usr/bin/env python3
"""Does the resident kernel's mailbox still land in host-writable device memory after the signal-handler probe went (round 6)?
us per env.step() at N = 1000 with the resident kernel: default placement vs MBT_RESIDENT_VRAM=0 (pinned host memory)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = """
import sys, time, numpy as np
sys.path.insert(0, %r)
import bench
from mbt_gym_amd.gym.ModelDynamics import LimitOrderModelDynamics
from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
from mbt_gym_amd.stochastic_processes.arrival_models import PoissonArrivalModel
from mbt_gym_amd.stochastic_processes.fill_probability_models import ExponentialFillFunction
from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel
n, n_steps = 1000, 200; dt = 1.0 / n_steps
dyn = LimitOrderModelDynamics(midprice_model=BrownianMotionMidpriceModel(volatility=2.0, initial_price=100, terminal_time=1.0, step_size=dt, num_trajectories=n),
    arrival_model=PoissonArrivalModel(intensity=np.array([140.0, 140.0]), step_size=dt, num_trajectories=n),
    fill_probability_model=ExponentialFillFunction(fill_exponent=1.5, step_size=dt, num_trajectories=n), num_trajectories=n)
env = TradingEnvironment(terminal_time=1.0, n_steps=n_steps, model_dynamics=dyn, initial_inventory=0, max_inventory=200, seed=50, num_trajectories=n,
                         normalise_action_space=False, normalise_observation_space=False, resident_step=True)
a = np.tile(np.array([[0.7, 0.7]], dtype=np.float32), (n, 1))
best = 1e9
for rep in range(5):
    env.reset()
    for _ in range(40): env.step(a)
    t0 = time.perf_counter()
    for _ in range(150): env.step(a)
    best = min(best, (time.perf_counter() - t0) / 150 * 1e6)
print(round(best, 2))
""" % ROOT
for label, knob in (("default placement (device memory through the BAR where the platform has it)", None), ("MBT_RESIDENT_VRAM=0 (pinned host memory)", "0")):
    env = dict(os.environ)
    if knob is not None:
        env["MBT_RESIDENT_VRAM"] = knob
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(f"{label}: {out.stdout.strip() or out.stderr[-300:]} us per env.step() at N = 1000, resident kernel")
