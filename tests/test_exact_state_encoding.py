"""The 8-byte exact representation of a float64 state value (include/mbt_env.h: mbt_exact_split / mbt_exact_join), which
the precise_state kernels keep their state in: hi = float32(x) in the row, an int32 remainder beside it.  Host-side
restatement of the device functions (csrc/step_kernel.hpp: exact_split / exact_join) - no GPU needed; the device pair is
pinned against it by tests/test_gpu_precise.py (state64 equals the float64 oracle bit for bit over whole episodes)."""
import ctypes as C

import numpy as np
import pytest

from mbt_gym_amd import _native


@pytest.fixture(scope="module")
def lib():
    return _native.load_library()


def split(lib, x):
    hi, lo = C.c_float(0), C.c_int32(0)
    lib.mbt_exact_split(float(x), C.byref(hi), C.byref(lo))
    return hi.value, lo.value


def restated_split(x):
    """The encoding in NumPy: hi = float32(x); lo = (x - hi) * 2^(53 - e), e = floor(log2 |hi|)."""
    hi = np.float32(x)
    if hi == 0 or not np.isfinite(hi) or abs(hi) < np.finfo(np.float32).tiny:
        return float(hi), 0
    e = int(np.floor(np.log2(abs(float(hi)))))
    return float(hi), int(np.ldexp(np.float64(x) - np.float64(hi), 53 - e))


def test_round_trip_is_exact_over_random_doubles(lib):
    rng = np.random.default_rng(1)
    samples = np.concatenate([
        rng.normal(size=20000) * 10.0 ** rng.integers(-30, 30, size=20000),  # every magnitude a state column can take
        rng.uniform(-1e5, 1e5, size=20000),                                  # cash
        100.0 + rng.normal(size=20000),                                      # midprice
        np.float64(np.float32(rng.uniform(-50, 50, size=2000))),             # float32-representable: remainder 0
    ])
    for x in samples:
        hi, lo = split(lib, x)
        assert hi == float(np.float32(x))  # the row shows np.float32(x), rounded to nearest
        assert abs(lo) <= 2 ** 29
        assert lib.mbt_exact_join(hi, lo) == x
        assert (hi, lo) == restated_split(x)
    hi, lo = split(lib, np.float64(np.float32(3.14159)))
    assert lo == 0


def test_binade_edges_zero_and_non_finite(lib):
    """hi rounds UP to a power of two (x just below it: the remainder is a multiple of 2^(e-53), not of 2^(e-52)); exact
    powers of two; the largest remainders (x half a float32 ulp from hi); zero; values float32 cannot hold as normals."""
    cases = [np.nextafter(2.0, 0.0), np.nextafter(1.0, 0.0), np.nextafter(1024.0, 0.0), -np.nextafter(2.0, 0.0), 1.0, 2.0, -0.5,
             1.0 + 2.0 ** -24, 1.0 + 2.0 ** -24 - 2.0 ** -52, 1.0 - 2.0 ** -25, 1.0 - 2.0 ** -25 + 2.0 ** -53,
             3.0 + 2.0 ** -23 + 2.0 ** -51, 1e30, -1e-30, 0.1 + 0.2 - 0.3]
    for x in cases:
        hi, lo = split(lib, x)
        assert lib.mbt_exact_join(hi, lo) == x, x
        assert abs(lo) <= 2 ** 29
    assert split(lib, 0.0) == (0.0, 0)
    hi, lo = split(lib, 1e-40)  # a float32 denormal: the remainder is dropped, the loss is below 2^-126
    assert lo == 0 and abs(lib.mbt_exact_join(hi, lo) - 1e-40) < 2.0 ** -126
    hi, lo = split(lib, 1e300)  # beyond float32: inf, nothing to add
    assert np.isinf(hi) and lo == 0
