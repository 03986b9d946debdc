#!/usr/bin/env python3
"""The float32 tier's x ** p (csrc/step_kernel.hpp: power_f32), modelled in NumPy with the kernel's own constants and checked against long double.

What is modelled: |x| = 2^k m, m in [sqrt(1/2), sqrt(2)); l0 = float32 log2(m), DELIBERATELY off by up to two float32 ulps of 1/2 (v_log_f32 is
good to one; the refinement must not depend on it); d = m 2^(-l0) - 1 and log2 m = l0 + d / ln 2 in double; y = p (k + log2 m); 2^rint(y) 2^(y - rint(y));
2^t on |t| <= 0.52 is the degree-8 polynomial whose coefficients this script reads OUT OF THE HEADER.  Prints the largest error in float32 ulps and the
share of results that differ from the correctly rounded one; `check()` is what tests/test_host_logic.py calls on a smaller sample.

    python tools/microbench/pow_f32_model.py [arguments per exponent, default 2 000 000]"""
import os
import re
import sys

import numpy as np

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "mbt_gym_amd", "csrc", "step_kernel.hpp")


def header_coefficients():
    """Highest degree first, as exp2_on_half_unit's Horner chain holds them, the final 1.0 included."""
    text = open(HEADER).read()
    body = text[text.index("double exp2_on_half_unit(double t)"):]
    body = body[:body.index("\n}\n")]
    found = [float.fromhex(h) for h in re.findall(r"0x1\.[0-9a-f]+p[-+]\d+", body)]
    assert len(found) == 8, found
    return found + [1.0]


def exp2_on_half_unit(t, coefficients):
    r = np.full_like(t, coefficients[0])
    for c in coefficients[1:]:
        r = r * t + c
    return r


def power_f32(x, p, rng, coefficients):
    ax = np.abs(x).astype(np.float32)
    m, k = np.frexp(ax)
    m, k = m.astype(np.float32), k.astype(np.int32)
    low = m < np.float32(0.70710678)
    m, k = np.where(low, m * np.float32(2), m), np.where(low, k - 1, k)
    l0 = np.log2(m.astype(np.float64)).astype(np.float32)
    l0 = (l0 + rng.integers(-2, 3, size=l0.shape).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
    d = m.astype(np.float64) * exp2_on_half_unit(-l0.astype(np.float64), coefficients) - 1.0
    y = p * (k.astype(np.float64) + (d * 1.4426950408889634 + l0.astype(np.float64)))
    y = np.clip(y, -2000.0, 2000.0)
    whole = np.rint(y)
    return np.ldexp(exp2_on_half_unit(y - whole, coefficients), whole.astype(np.int64)).astype(np.float32)


def check(samples, exponents=(0.5, 0.6, 1.5, 2.5, 3.0, 0.25, 4.7, -1.3), seed=1):
    rng, coefficients, rows = np.random.default_rng(seed), header_coefficients(), []
    for p in exponents:
        x = np.exp(rng.uniform(np.log(1e-4), np.log(1e3), size=samples)).astype(np.float32)
        got = power_f32(x, p, rng, coefficients)
        exact = np.power(x.astype(np.longdouble), np.longdouble(p))
        rounded = exact.astype(np.float32)
        ulps = np.abs(got.astype(np.longdouble) - exact) / np.spacing(np.abs(rounded))
        rows.append((p, float(ulps.max()), float((got != rounded).mean())))
    return rows


if __name__ == "__main__":
    for p, worst, differing in check(int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000):
        print(f"p = {p:5}: max error {worst:.6f} ulp, {differing:.2e} of the results are not the correctly rounded one")
