#!/usr/bin/env python3
"""Builds an EXPERIMENT copy of the library next to the real one: `python tools/dbg/build_variant.py NAME -DMBT_EXP_X=1 ...`
-> mbt_gym_amd/libmbtenv_NAME.so (same sources, same baked hash, extra preprocessor flags; every translation unit of
mbt_gym_amd/build.py, compiled in parallel, objects cached by content + flags under build/).  Scripts under tools/dbg/ load it by
setting mbt_gym_amd._native.LIB_PATH before the first use (MBT_LIB_VARIANTS in tools/dbg/ab_policy.py / ab_configs.py), so two code
variants can be timed back to back on the same box.  Never used by the package, the tests or bench.py."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
name, extra = sys.argv[1], sys.argv[2:]
os.environ["MBT_EXTRA_HIPCC_FLAGS"] = " ".join(extra)  # (read when mbt_gym_amd.build is imported)
from mbt_gym_amd import build as b  # noqa: E402

real = b.LIB_PATH
out = os.path.join(os.path.dirname(real), f"libmbtenv_{name}.so")
keep = real + ".keep"
if os.path.exists(real):
    shutil.copy2(real, keep)
try:
    b.build_native(force=False if not extra else True)  # (force: the staleness check looks at the sources, not at the flags)
    shutil.copy2(real, out)
finally:
    if os.path.exists(keep):
        os.replace(keep, real)
print(out)
