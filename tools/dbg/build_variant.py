#!/usr/bin/env python3
"""Builds an EXPERIMENT copy of the library next to the real one: `python tools/dbg/build_variant.py NAME -DMBT_EXP_X=1 ...`
-> mbt_gym_amd/libmbtenv_NAME.so (same sources, same baked hash, extra preprocessor flags).  Scripts under tools/dbg/ load it
by setting mbt_gym_amd._native.LIB_PATH before the first use (MBT_LIB_VARIANT=NAME in tools/dbg/ab_policy.py), so two code
variants can be timed back to back on the same box.  Never used by the package, the tests or bench.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mbt_gym_amd import build as b  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
b.write_embedded_sources()
out = os.path.join(os.path.dirname(b.LIB_PATH), f"libmbtenv_{name}.so")
cmd = ["/opt/rocm/bin/hipcc"] + b.HIPCC_FLAGS + extra + [f'-DMBT_SOURCE_HASH="{b.source_hash()}"'] + [os.path.join(b.CSRC, s) for s in b.SOURCES] + b.LINK_FLAGS + ["-o", out]
subprocess.run(cmd, check=True, cwd=b.CSRC)
print(out)
