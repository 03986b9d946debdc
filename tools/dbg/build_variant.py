#!/usr/bin/env python3
"""Builds an EXPERIMENT copy of the library next to the real one: `python tools/dbg/build_variant.py NAME -DMBT_EXP_X=1 ...`
-> mbt_gym_amd/libmbtenv.NAME.so (same sources, extra preprocessor flags; every translation unit of mbt_gym_amd/build.py, compiled in
parallel, objects cached by content + flags under build/).  A process loads it with
    MBT_LIBRARY_VARIANT=NAME MBT_EXTRA_HIPCC_FLAGS="-DMBT_EXP_X=1 ..." python ...
(the flags are part of the source hash the binding checks), so two code variants can be timed back to back on the same box.  Never used by
the package, the tests or bench.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
name, extra = sys.argv[1], sys.argv[2:]
os.environ["MBT_EXTRA_HIPCC_FLAGS"] = " ".join(extra)  # (read when mbt_gym_amd.build is imported)
os.environ["MBT_BUILD_VARIANT"] = name
from mbt_gym_amd import build as b  # noqa: E402

print(b.build_native())
