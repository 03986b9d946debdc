"""The fill decision is exact against float64 for EVERY uniform the generator can produce.

The kernel decides `u < exp(-kappa * depth)` (FILL:34, FILL:57-58) as `depth < t(u)` with a float32 threshold bracket
computed from v_log_f32 and re-decides in double only inside the bracket (csrc/step_kernel.hpp: fill_thresholds).
Here all 2^24 uniforms m * 2^-24 are injected, one per lane, with quotes placed ON the decision boundary - the float32
depth nearest to the exact threshold -ln(u)/kappa and its neighbours - which is where a sloppy bracket would flip
decisions.  Expected: NumPy float64, the reference's expression.

One caveat defines "exact": float64 `exp` is not correctly rounded in NumPy (nor in any libm), so where the uniform
equals exp(-kappa * depth) to within one float64 ulp the reference's own answer depends on the NumPy build.  Such ties
exist (u = 1 - 96 * 2^-24 against the float32 depth nearest its threshold: the exact exponential exceeds u by 6.2e-17,
0.56 ulp; NumPy 2.2 rounds it down to u, a correctly rounded exp rounds it up).  Cases within one ulp are excluded:
a few dozen among the 10^8 boundary cases tried, of which that one actually decides differently."""
import numpy as np
import pytest

from oracle.mbt_oracle import OracleConfig
from tests.env_factory import make_env

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kappa", [1.5, 0.05, 40.0])
def test_every_uniform_against_boundary_depths(kappa):
    n = 1 << 24
    cfg = OracleConfig(num_trajectories=n, n_steps=10, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0, arrival="poisson",
                       intensity=(5.0, 5.0), fill_exponent=kappa, dynamics="limit", reward="pnl", initial_inventory=0, max_inventory=1000,
                       max_depth=1e9, seed=1, normalise_action_space=False, normalise_observation_space=False)
    env = make_env(cfg, noise="injected")
    env.record_events(True)
    env.reset()
    u = (np.arange(n, dtype=np.float64) / float(1 << 24)).astype(np.float32)  # every 24-bit uniform, 0 included
    with np.errstate(divide="ignore"):
        boundary = (-np.log(u.astype(np.float64)) / kappa).astype(np.float32)  # u = 0 -> inf
    boundary[0] = np.float32(745.0 / kappa)  # exp underflows to zero near here: 0 < exp(-745.2) is still true, 0 < exp(-746) is not
    up = np.nextafter(boundary, np.float32(np.inf))
    down = np.nextafter(boundary, np.float32(-np.inf))
    u_fill = np.stack([u, u], axis=1)
    u_arr = np.zeros((n, 2), np.float32)  # an order arrives on both sides in every lane
    z = np.zeros(n, np.float32)
    wrong, ties = [], 0
    for bid, ask in ((boundary, up), (down, np.nextafter(up, np.float32(np.inf))), (boundary * np.float32(1.0000005), boundary * np.float32(0.9999995))):
        depths = np.stack([bid, ask], axis=1).astype(np.float32)
        env.set_noise(u_arr, u_fill, z)
        env.step(depths)
        got = env.last_fills
        p = np.exp(-kappa * depths.astype(np.float64))
        want = u_fill.astype(np.float64) < p
        tie = (u_fill.astype(np.float64) < np.nextafter(p, 0.0)) != (u_fill.astype(np.float64) < np.nextafter(p, 2.0))
        ties += int(np.count_nonzero(tie))
        for lane, side in zip(*np.nonzero((got != want) & ~tie)):
            wrong.append((int(lane), int(side), float(u[lane]), float(depths[lane, side]), bool(got[lane, side]), bool(want[lane, side])))
        assert np.all(env.last_arrivals)
    assert not wrong, wrong[:10]
    assert ties <= 64, ties
    env.close()
