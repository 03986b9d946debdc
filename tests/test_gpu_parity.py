"""Parity of the HIP step kernel (through the C ABI and the public Python API) with the reference.

Inputs: the committed golden fixtures (what the REAL reference produced for the same injected noise and actions)
and, redundantly, the float64 oracle run live on the same inputs.  Stated tolerances - each at most 2x what
tests/perf/parity_report.py measures (profiles/r02_parity_report.txt):
  arrivals, fills (post mask), market-order flags, inventory, dones ....... bit-exact
  rewards ................................................................ |err| <= 1e-5 + 1e-6 |r|   (float32 output:
      the relative term matters only for |r| >> 1 - a terminal penalty on real-valued inventory, a GBM price move
      times the inventory; measured: <= 6.3e-6 for |r| <= 13, relative <= 7.6e-7 beyond)
      EXCEPT on lane-steps where the cash / inventory clip of TE:283-289 changed a value (the reference prints its
      whole state there): the reward then contains the LEVEL of the float32 cash / midprice state, not just the step's
      increments: |err| <= 1.2e-4 (measured 5.8e-5).  `precise_state=True` removes this exception
      (tests/test_gpu_precise.py: <= 1e-5 on every lane-step, measured 3.7e-6).
  cash ................................................................... |err| <= 1e-6 * (largest |cash| the lane
                                                                           has held this episode) + 1e-4
  midprice ............................................................... |err| <= 2e-4   (S ~ 100: ulp = 7.6e-6,
                                                                           random-walk of the per-step rounding; measured 8.4e-5)
  Hawkes intensities ..................................................... EQUAL to np.float32(reference) (round 5: the two
      intensity columns are held exactly - float32 row + int32 remainder - and advanced in double in the reference's order,
      ARR:110-123; `hawkes_float32_intensities=True` restores float32 state: |err| <= 2e-5 + 3e-7 |lambda|)
  time ................................................................... 1e-6 abs
  normalised observations ................................................ 5e-5 abs (midprice drift / half-width 8)
The state is float32 in HBM (it IS the float32 observation the API returns), so cash and midprice carry the
accumulated float32 rounding of an episode; decisions never depend on it and rewards only where a clip fires.
"""
import numpy as np
import pytest

from oracle.mbt_oracle import InjectedNoise, OracleEnv
from tests.env_factory import make_env
from tests.golden_io import KERNEL_NOISE_CASES as CASES, load_case, step_size_changes

pytestmark = pytest.mark.gpu

REWARD_ATOL, REWARD_RTOL = 1e-5, 1e-6   # north_star's 1e-5, plus the float32 output's own rounding of a reward >> 1
REWARD_ATOL_CLIPPED = 1.2e-4             # lane-steps where TE:283-289 clipped: 2x the measured 5.8e-5


def _is_speed(name):
    return name.startswith("speed_") or name.endswith("_speed")


def _check_obs(name, k, got, want, normalised, max_inventory, cash_scale=None, exact_intensities=False):
    if normalised:
        np.testing.assert_allclose(got, want, rtol=0, atol=5e-5, err_msg=f"{name} step {k}: normalised obs")
        # inventory is an integer count: exact after de-normalisation
        q_got = np.rint((got[:, 1].astype(np.float64) + 1) * max_inventory - max_inventory)
        q_want = np.rint((want[:, 1] + 1) * max_inventory - max_inventory)
        np.testing.assert_array_equal(q_got, q_want, err_msg=f"{name} step {k}: inventory")
        return
    if _is_speed(name):  # real-valued inventory (MD:267) and the impact state: float32 accumulation
        np.testing.assert_allclose(got[:, 1], want[:, 1], rtol=1e-6, atol=1e-6, err_msg=f"{name} step {k}: inventory")
        if want.shape[1] > 4:
            np.testing.assert_allclose(got[:, 4], want[:, 4], rtol=1e-5, atol=1e-7, err_msg=f"{name} step {k}: impact state")
    else:
        np.testing.assert_array_equal(got[:, 1].astype(np.float64), want[:, 1], err_msg=f"{name} step {k}: inventory")
    np.testing.assert_allclose(got[:, 2], want[:, 2], rtol=0, atol=1e-6, err_msg=f"{name} step {k}: time")
    cash_tol = 1e-4 + 1e-6 * (np.abs(want[:, 0]) if cash_scale is None else cash_scale)
    assert np.all(np.abs(got[:, 0] - want[:, 0]) <= cash_tol), f"{name} step {k}: cash {np.max(np.abs(got[:, 0] - want[:, 0]))}"
    np.testing.assert_allclose(got[:, 3], want[:, 3], rtol=0, atol=2e-4, err_msg=f"{name} step {k}: midprice")
    if want.shape[1] == 5 and not _is_speed(name):  # the second factor of a user's two-column midprice (float32 state, O(1) values)
        np.testing.assert_allclose(got[:, 4], want[:, 4], rtol=2e-6, atol=2e-5, err_msg=f"{name} step {k}: second midprice factor")
    if want.shape[1] > 5:
        if exact_intensities:  # the built-in Hawkes model in the default tier: the reference's float64 intensities, rounded once
            np.testing.assert_array_equal(got[:, 4:6], want[:, 4:6].astype(np.float32), err_msg=f"{name} step {k}: intensities")
        # float32 state otherwise: 2e-5 absolute around the baselines (10..50), float32 relative accuracy where arrivals have driven an
        # intensity to ~150 (ulp 1.5e-5 there)
        np.testing.assert_allclose(got[:, 4:], want[:, 4:], rtol=3e-7, atol=2e-5, err_msg=f"{name} step {k}: intensities")


@pytest.mark.parametrize("record", [True, False])
@pytest.mark.parametrize("name", CASES)
def test_step_matches_reference_fixture(name, record):
    cfg, g = load_case(name)
    env = make_env(cfg, noise="injected")
    exact = cfg.arrival == "hawkes"
    oracle = OracleEnv(cfg, InjectedNoise(g["u_arr"], g["u_fill"], g["z"], g.get("z_user")))
    if record:
        env.record_events(True)
    obs0 = env.reset()
    oracle.reset()
    assert obs0.dtype == np.float32 and obs0.shape == g["obs0"].shape
    _check_obs(name, -1, obs0, g["obs0"], cfg.normalise_observation_space, cfg.max_inventory, exact_intensities=exact)
    cash_scale = np.abs(g["obs0"][:, 0]) if not cfg.normalise_observation_space else None
    changes = step_size_changes(g)
    for k in range(g["actions"].shape[0]):
        if k in changes:  # the step_size setter in mid-episode (TE:158-167): host-side kernel parameters only
            env.step_size = changes[k]
            oracle.set_step_size(changes[k])
        env.set_noise(g["u_arr"][k], g["u_fill"][k], g["z"][k], g["z_user"][k] if "z_user" in g else None)
        obs, rew, dones, infos = env.step(g["actions"][k])
        o_obs, o_rew, o_done = oracle.step(g["actions"][k].astype(np.float64))
        # the live oracle and the stored reference outputs agree exactly (CPU test), so either is the target
        np.testing.assert_array_equal(o_rew, g["rewards"][k])
        if record and not _is_speed(name):
            np.testing.assert_array_equal(env.last_arrivals.astype(np.uint8), g["arrivals"][k], err_msg=f"{name} step {k}: arrivals")
            np.testing.assert_array_equal(env.last_fills.astype(np.uint8), g["fills"][k], err_msg=f"{name} step {k}: fills")
        if cash_scale is not None:
            cash_scale = np.maximum(cash_scale, np.abs(g["obs"][k][:, 0]))
        _check_obs(name, k, obs, g["obs"][k], cfg.normalise_observation_space, cfg.max_inventory, cash_scale, exact_intensities=exact)
        # lanes where the clip of TE:283-289 changed cash or inventory, from the oracle; the kernel's event bits agree
        clipped = oracle.last_clipped
        if record:
            np.testing.assert_array_equal((env.last_events >> 6) != 0, clipped, err_msg=f"{name} step {k}: clip flags")
        err = np.abs(rew - g["rewards"][k])
        assert np.all(err[clipped] <= REWARD_ATOL_CLIPPED), f"{name} step {k}: reward on clipped lanes {err[clipped].max()}"
        tol = REWARD_ATOL + REWARD_RTOL * np.abs(g["rewards"][k])
        assert np.all(err[~clipped] <= tol[~clipped]), f"{name} step {k}: rewards off by {err[~clipped].max()}"
        assert dones.shape == (cfg.num_trajectories,) and bool(dones[0]) == bool(g["done"][k])
        assert len(infos) == cfg.num_trajectories
    env.close()


def test_step_before_noise_is_an_error():
    from mbt_gym_amd._native import NativeError

    cfg, g = load_case("as_limit_pnl")
    env = make_env(cfg, noise="injected")
    env.reset()
    with pytest.raises(NativeError):
        env.step(g["actions"][0])
    env.close()
