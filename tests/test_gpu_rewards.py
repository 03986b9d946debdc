"""The reference's own automated tests (mbt_gym/rewards/tests/testRewardFunctions.py:33-135) run against the
package's RewardFunction classes, whose calculate() executes on the device in double precision."""
import copy

import numpy as np
import pytest

from mbt_gym_amd.rewards.RewardFunctions import CjCriterion, CjMmCriterion, PnL, RunningInventoryPenalty
from oracle.mbt_oracle import cj_mm_criterion, pnl_reward, running_inventory_penalty
from tests.test_oracle_rewards import ALPHA, CUR, DT, EPISODE, NXT, PHI, T_END

pytestmark = pytest.mark.gpu
ACTION = np.array([[1, 1]])


def test_pnl_per_step_reward_is_exact():
    want = (NXT[:, 0] + NXT[:, 1] * NXT[:, 3]) - (CUR[:, 0] + CUR[:, 1] * CUR[:, 3])
    got = PnL().calculate(current_state=CUR, action=ACTION, next_state=NXT)
    assert got == want  # assertEqual in the reference's test


def test_running_inventory_penalty_per_step_reward():
    want = PnL().calculate(CUR, ACTION, NXT) - PHI * DT * abs(NXT[:, 1]) ** 2
    got = RunningInventoryPenalty(PHI, ALPHA).calculate(CUR, ACTION, NXT)
    assert abs(want.item() - got.item()) < 5e-6
    assert CjCriterion is RunningInventoryPenalty


def _telescope(states, start=0):
    cj = CjMmCriterion(PHI, ALPHA, terminal_time=T_END)
    target = RunningInventoryPenalty(PHI, ALPHA)
    cj.reset(states[start])
    a = b = 0.0
    for i in range(start, len(states) - 1):
        terminal = states[i + 1][:, 2] == 1
        a += cj.calculate(states[i], ACTION, states[i + 1], terminal).item()
        b += target.calculate(states[i], ACTION, states[i + 1], terminal).item()
    assert abs(a - b) < 5e-6


def test_cjmm_agrees_with_the_non_deconstructed_version():
    _telescope(EPISODE)


def test_cjmm_nonzero_initial_inventory():
    states = copy.deepcopy(EPISODE)
    states[0][:, 1] = 2
    states[0][:, 0] = -100
    states[-1] = copy.deepcopy(states[-2])
    states[-1][:, 2] = 1.0
    _telescope(states)


def test_cjmm_partial_trajectory():
    _telescope(EPISODE, start=2)


def test_device_calculate_is_bitwise_the_float64_formula():
    rng = np.random.default_rng(0)
    cur = np.column_stack([rng.normal(0, 500, 257), rng.integers(-9, 10, 257), np.full(257, 0.3), rng.normal(100, 2, 257)])
    nxt = cur + np.column_stack([rng.normal(0, 100, 257), rng.integers(-1, 2, 257), np.full(257, 0.005), rng.normal(0, 0.1, 257)])
    np.testing.assert_array_equal(PnL().calculate(cur, None, nxt), pnl_reward(cur, nxt))
    rip = RunningInventoryPenalty(0.01, 0.5)
    np.testing.assert_array_equal(rip.calculate(cur, None, nxt, True), running_inventory_penalty(cur, nxt, True, 0.01, 0.5, 2.0))
    cj = CjMmCriterion(0.01, 0.001, terminal_time=1.0)
    init = cur.copy()
    init[:, 2] = 0.1
    cj.reset(init)
    want = cj_mm_criterion(cur, nxt, 0.01, 0.001, 2.0, init[:, 1], 1.0 - init[:, 2])
    np.testing.assert_allclose(cj.calculate(cur, None, nxt), want, rtol=0, atol=1e-12)
    with pytest.raises(Exception):
        CjMmCriterion().calculate(cur, None, nxt)  # before reset(): the reference fails too (None ** p)


def test_process_objects_driven_on_their_own_walk_the_references_path():
    """The reference's plugin objects can be used outside an environment (SP:33-35, ARR:27-29, FILL:28-34).  Ours then keep a
    host copy of their state, draw from their own numpy Generator like the reference's classes, and have each call's
    arithmetic evaluated on the device in double, in the reference's operation order: seeded alike they produce the values
    the reference's NumPy statements produce (restated here line by line; exp() is the one libm call)."""
    from mbt_gym_amd.stochastic_processes.arrival_models import HawkesArrivalModel, PoissonArrivalModel, PoissonArrivalNonLinearModel
    from mbt_gym_amd.stochastic_processes.fill_probability_models import ExponentialFillFunction
    from mbt_gym_amd.stochastic_processes.midprice_models import (BrownianMotionMidpriceModel, GeometricBrownianMotionMidpriceModel, OuJumpMidpriceModel,
                                                                 OuMidpriceModel)

    n, dt, seed = 500, 0.01, 7
    # midprices: MID:60-65, MID:140-143, MID:95-103, MID:264-270
    bm = BrownianMotionMidpriceModel(drift=0.3, volatility=2.0, initial_price=100.0, step_size=dt, num_trajectories=n, seed=seed)
    rng, s = np.random.default_rng(seed), np.full((n, 1), 100.0)
    for _ in range(5):
        s = s + 0.3 * dt * np.ones((n, 1)) + 2.0 * np.sqrt(dt) * rng.normal(size=(n, 1))
        np.testing.assert_array_equal(bm.update(None, None, None), s)
    bm.reset()
    np.testing.assert_array_equal(bm.current_state, np.full((n, 1), 100.0))
    ou = OuMidpriceModel(mean_reversion_level=99.0, mean_reversion_speed=0.05, volatility=1.5, initial_price=100.0, step_size=dt, num_trajectories=n, seed=seed)
    rng, s = np.random.default_rng(seed), np.full((n, 1), 100.0)
    for _ in range(5):
        s = s + (-0.05 * (s - 99.0 * np.ones((n, 1))) + 1.5 * np.sqrt(dt) * rng.normal(size=(n, 1)))
        np.testing.assert_array_equal(ou.update(None, None, None), s)
    gbm = GeometricBrownianMotionMidpriceModel(drift=0.1, volatility=0.2, initial_price=50.0, step_size=dt, num_trajectories=n, seed=seed)
    rng, s = np.random.default_rng(seed), np.full((n, 1), 50.0)
    for _ in range(5):
        s = s + 0.1 * s * dt + 0.2 * s * np.sqrt(dt) * rng.normal(size=(n, 1))
        np.testing.assert_array_equal(gbm.update(None, None, None), s)
    jump = OuJumpMidpriceModel(mean_reversion_level=100.0, mean_reversion_speed=0.02, volatility=1.0, jump_size=0.25, initial_price=100.0, step_size=dt,
                               num_trajectories=n, seed=seed)
    rng, s, events = np.random.default_rng(seed), np.full((n, 1), 100.0), np.random.default_rng(1)
    for _ in range(5):
        arrivals, fills = events.integers(0, 2, size=(n, 2)).astype(bool), events.integers(0, 2, size=(n, 2))
        j = (0.25 * (fills[:, 1] * arrivals[:, 1]) - 0.25 * (fills[:, 0] * arrivals[:, 0])).reshape(-1, 1)
        s = s - 0.02 * (s - 100.0 * np.ones((n, 1))) + 1.0 * np.sqrt(dt) * rng.normal(size=(n, 1)) + j
        np.testing.assert_array_equal(jump.update(arrivals, fills, None), s)
    # arrivals: ARR:54-56, ARR:81-83, ARR:110-123
    for cls, thr in ((PoissonArrivalModel, np.array([140.0, 90.0]) * dt), (PoissonArrivalNonLinearModel, 1.0 - np.exp(-np.array([140.0, 90.0]) * dt))):
        model, rng = cls(intensity=np.array([140.0, 90.0]), step_size=dt, num_trajectories=n, seed=seed), np.random.default_rng(seed)
        for _ in range(3):
            np.testing.assert_array_equal(model.get_arrivals(), rng.uniform(size=(n, 2)) < thr)
    hawkes = HawkesArrivalModel(baseline_arrival_rate=np.array([[10.0, 14.0]]), step_size=dt, jump_size=40.0, mean_reversion_speed=60.0, num_trajectories=n, seed=seed)
    rng, lam = np.random.default_rng(seed), np.repeat(np.array([[10.0, 14.0]]), n, axis=0)
    for _ in range(6):
        arrivals = rng.uniform(size=(n, 2)) < lam * dt
        np.testing.assert_array_equal(hawkes.get_arrivals(), arrivals)
        lam = lam + 60.0 * (np.ones((n, 2)) * np.array([[10.0, 14.0]]) - lam) * dt * np.ones((n, 2)) + 40.0 * arrivals
        np.testing.assert_array_equal(hawkes.update(arrivals, None, None), lam)
    # fills: FILL:28-34, FILL:57-58 (exp: device libm against NumPy's - a draw within an ulp of the probability may differ)
    fill, rng = ExponentialFillFunction(fill_exponent=1.5, step_size=dt, num_trajectories=n, seed=seed), np.random.default_rng(seed)
    depths = np.random.default_rng(2).uniform(0.0, 2.0, size=(n, 2))
    unif = rng.uniform(size=(n, 2))
    got, want = fill.get_fills(depths), unif < np.exp(-1.5 * depths)
    assert np.all((got == want) | (np.abs(unif - np.exp(-1.5 * depths)) < 1e-15))
    with pytest.raises(AssertionError):
        fill.get_fills(depths[:, :1])


def test_the_clip_cash_fixture_raises_the_float32_tiers_clip_warning_at_its_next_reset():
    """`clip_cash` (a max_cash small enough for TE:283-289 to fire): the float32 tier says so at the episode boundary, once;
    precise_state has nothing to say."""
    import warnings

    from mbt_gym_amd.gym.TradingEnvironment import Float32ClipWarning
    from tests.env_factory import make_env
    from tests.golden_io import load_case

    cfg, g = load_case("clip_cash")
    for precise in (False, True):
        env = make_env(cfg, noise="injected", precise_state=precise)
        with warnings.catch_warnings():
            warnings.simplefilter("error", Float32ClipWarning)
            env.reset()  # a fresh environment: no clip yet
        for k in range(g["actions"].shape[0]):
            env.set_noise(g["u_arr"][k], g["u_fill"][k], g["z"][k])
            env.step(g["actions"][k])
        assert env.clip_count > 0
        if precise:
            with warnings.catch_warnings():
                warnings.simplefilter("error", Float32ClipWarning)
                env.reset()
        else:
            with pytest.warns(Float32ClipWarning, match="clipped to max_inventory / max_cash"):
                env.reset()
            with warnings.catch_warnings():
                warnings.simplefilter("error", Float32ClipWarning)
                env.reset()  # once per environment
        env.close()


@pytest.mark.parametrize("p", [0.5, 0.6, 1.5, 2.5, 3.0, 4.0, -1.3])
def test_the_float32_tiers_power_is_the_float64_power_rounded_once(p):
    """mbt_power_f32_device = step_kernel.hpp: power_f32, the `x ** p` of the float32 kernels for exponents other than 1 and 2 (IMP:55-56
    `action ** exponent`; RW:59-68, :101-104, :133-137 `inventory ** exponent`): within 0.5002 float32 ulp of NumPy's float64 power over six
    decades of magnitude, both signs (a negative base keeps NumPy's rule: its sign by the exponent's parity, NaN for a fractional exponent),
    zeros, and - through the library's pow, out of line - subnormal, infinite and NaN bases."""
    import ctypes as C

    from mbt_gym_amd import _native

    lib = _native.load_library()
    rng = np.random.default_rng(int(abs(p) * 100))
    x = np.exp(rng.uniform(np.log(1e-3), np.log(1e3), size=1 << 20)).astype(np.float32)
    x[::3] *= -1.0
    special = np.array([0.0, -0.0, 1.0, -1.0, 1e-41, -1e-41, np.inf, -np.inf, np.nan, np.float32(1.17549435e-38), np.float32(3.4e38)], dtype=np.float32)
    x = np.concatenate([special, x])
    got = np.empty_like(x)
    _native.check(lib.mbt_power_f32_device(0, x.ctypes.data_as(C.POINTER(C.c_float)), p, got.ctypes.data_as(C.POINTER(C.c_float)), len(x)))
    with np.errstate(all="ignore"):
        exact = np.power(x.astype(np.longdouble), np.longdouble(p))
        want = exact.astype(np.float32)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(np.signbit(got[~np.isnan(want)]), np.signbit(want[~np.isnan(want)]))
    finite = np.isfinite(want) & (np.abs(want) >= np.float32(1.17549435e-38))  # (float32 subnormal results: one subnormal ulp, below)
    ulps = np.abs(got[finite].astype(np.longdouble) - exact[finite]) / np.spacing(np.abs(want[finite]))
    assert float(ulps.max()) < 0.5002, (p, float(ulps.max()))
    rest = ~finite & ~np.isnan(want)
    np.testing.assert_allclose(got[rest], want[rest], rtol=0, atol=1.5e-45)  # 0, inf, and subnormal results to one subnormal ulp
