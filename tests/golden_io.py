"""Loading of the committed golden fixtures (tests/golden/*.npz, made by tools/refgen/make_golden.py)."""
import glob
import json
import os

import numpy as np

from oracle.mbt_oracle import OracleConfig

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_case(name):
    data = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    raw = json.loads(str(data.pop("config_json")))
    if isinstance(raw.get("initial_inventory"), list):
        raw["initial_inventory"] = tuple(raw["initial_inventory"])
    return OracleConfig(**raw), data
