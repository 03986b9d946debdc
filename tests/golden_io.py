"""Loading of the committed golden fixtures (tests/golden/*.npz, made by tools/refgen/make_golden.py)."""
import glob
import json
import os

import numpy as np

from oracle.mbt_oracle import OracleConfig

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# environment trajectories; `agents_*.npz` (tools/refgen/make_agent_golden.py) hold agent outputs and are loaded by their own tests
CASES = sorted(name for name in (os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
               if not name.startswith("agents_"))
# fixtures with a NumPy-only class that draws from ITS OWN generator (a midprice model's update() with trading-with-speed dynamics,
# an arrival model's get_arrivals()): the generic per-fixture GPU tests inject noise into the kernels and cannot feed that generator -
# tests/test_gpu_host_callbacks.py replays it
DRAWS_ON_THE_HOST = ("user_cev_midprice_speed", "user_state_reading_arrivals")
KERNEL_NOISE_CASES = [name for name in CASES if name not in DRAWS_ON_THE_HOST]


def load_case(name):
    data = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    raw = json.loads(str(data.pop("config_json")))
    if isinstance(raw.get("initial_inventory"), list):
        raw["initial_inventory"] = tuple(raw["initial_inventory"])
    return OracleConfig(**raw), data


def step_size_changes(g):
    """{step index: new step size} of fixtures that use the step_size setter in mid-episode (TE:158-167)."""
    if "step_size_at" not in g:
        return {}
    return {int(k): float(v) for k, v in zip(g["step_size_at"], g["step_size_to"])}
