#!/bin/bash
# x ** p of the float32 tier (step_kernel.hpp: power_f32): the shipped form against experiment builds, timed back to back on one box, and the
# device function's error for each (tests/test_gpu_rewards.py).   python tools/dbg/build_variant.py vexp -DMBT_EXP_POW_VEXP=1 first.
for v in ${VARIANTS:-"" vexp}; do
  [ -n "$v" ] && [ ! -f mbt_gym_amd/libmbtenv.$v.so ] && continue
  for rep in 1 2; do
    MBT_LIBRARY_VARIANT=$v MBT_BENCH_STEPS=2000 MBT_BENCH_ONLY="speed power impact" python tests/perf/bench_configs.py 2>/dev/null | python -c "
import json, sys
d = json.load(sys.stdin)
print('variant ${v:-shipped}:', ', '.join(f'{k[:28]}: {v[\"us_per_step\"]}' for k, v in d.items()))"
  done
  MBT_LIBRARY_VARIANT=$v python - <<'PY'
import ctypes as C
import numpy as np
from mbt_gym_amd import _native
lib = _native.load_library()
rng = np.random.default_rng(5)
for p in (0.6, 1.5, 4.7):
    x = np.exp(rng.uniform(np.log(1e-3), np.log(1e3), size=1 << 21)).astype(np.float32)
    got = np.empty_like(x)
    _native.check(lib.mbt_power_f32_device(0, x.ctypes.data_as(C.POINTER(C.c_float)), p, got.ctypes.data_as(C.POINTER(C.c_float)), len(x)))
    exact = np.power(x.astype(np.longdouble), np.longdouble(p))
    ulps = np.abs(got.astype(np.longdouble) - exact) / np.spacing(np.abs(exact.astype(np.float32)))
    print(f"   p = {p}: max {float(ulps.max()):.4f} ulp, not correctly rounded: {float((got != exact.astype(np.float32)).mean()):.2e}")
PY
done
