#!/usr/bin/env python3
"""Builds an EXPERIMENT copy of the library next to the real one: `python tools/dbg/build_variant.py NAME -DMBT_EXP_X=1 ...`
-> mbt_gym_amd/libmbtenv.NAME.so (same sources, extra preprocessor flags; every translation unit of mbt_gym_amd/build.py, compiled in
parallel, objects cached by content + flags under build/).  A process loads it with
    MBT_LIBRARY_VARIANT=NAME MBT_EXTRA_HIPCC_FLAGS="-DMBT_EXP_X=1 ..." python ...
(or `MBT_LIBRARY_VARIANT=NAME` alone: the flags, which are part of the source hash the binding checks, are left beside the library in
libmbtenv.NAME.flags), so two code variants can be timed back to back on the same box.  Never used by
the package, the tests or bench.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
name, extra = sys.argv[1], sys.argv[2:]
os.environ["MBT_EXTRA_HIPCC_FLAGS"] = " ".join(extra)  # (read when mbt_gym_amd.build is imported)
os.environ["MBT_BUILD_VARIANT"] = name
from mbt_gym_amd import build as b  # noqa: E402

path = b.build_native()
with open(path[:-len(".so")] + ".flags", "w") as f:  # (what `MBT_LIBRARY_VARIANT=NAME` needs to know to accept the library: _native.py)
    f.write(" ".join(extra) + "\n")
print(path)
