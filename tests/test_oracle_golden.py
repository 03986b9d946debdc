"""The oracle (oracle/mbt_oracle.py) against the reference's own outputs - bit for bit, float64.

The fixtures hold what the real reference produced in the build container for the injected noise
and actions they also hold; the oracle must reproduce every array exactly.  CPU only.
"""
import numpy as np
import pytest

from oracle.mbt_oracle import (
    InjectedNoise,
    NumpyProtocolNoise,
    OracleConfig,
    OracleEnv,
    avellaneda_stoikov_action,
    results_table,
    rollout,
)
from tests.golden_io import CASES, load_case, step_size_changes


def test_fixture_set_is_complete():
    assert set(CASES) >= {
        "as_limit_pnl", "cjp_running", "cjp_cjmm", "hawkes_ou", "limit_and_market", "default_normalised", "clip_cash",
        "gbm_nonlinear_touch", "bmjump_exputility", "oujump_hawkes_running", "constant_midprice",
        "speed_temp_perm_cjoe", "speed_power_running", "speed_temp_transient_pnl", "speed_transient_pnl",
        "step_size_change_hawkes", "step_size_change_speed", "user_linear_sde_midprice",
        "user_fill_and_reward", "user_fill_hawkes_market_normalised", "user_reward_touch", "user_seasonal_arrivals", "user_cev_midprice",
        "user_cross_hawkes", "user_two_factor_midprice", "user_two_factor_midprice_normalised", "user_reward_speed", "user_cev_midprice_speed", "user_impact_speed", "user_adaptive_fill", "user_state_reading_arrivals",
    }


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_bit_for_bit(name):
    cfg, g = load_case(name)
    env = OracleEnv(cfg, InjectedNoise(g["u_arr"], g["u_fill"], g["z"], g.get("z_user")))
    np.testing.assert_array_equal(env.obs_lo, g["obs_lo"])
    np.testing.assert_array_equal(env.obs_hi, g["obs_hi"])
    np.testing.assert_array_equal(env.act_lo, g["act_lo"])
    np.testing.assert_array_equal(env.act_hi, g["act_hi"])
    assert env.max_cash == float(g["max_cash"])
    obs0 = env.reset()
    # tuple initial inventories come from default_rng(seed).integers (TE:271-272): same protocol here
    np.testing.assert_array_equal(env.state[:, 1], g["q0"])
    assert float(env.state[0, 2]) == float(g["t0"])
    np.testing.assert_array_equal(obs0, g["obs0"])
    changes = step_size_changes(g)
    for k in range(g["actions"].shape[0]):
        if k in changes:
            env.set_step_size(changes[k])
        obs, rew, done = env.step(g["actions"][k].astype(np.float64))
        if env.last_arrivals is not None:  # speed dynamics have neither arrivals nor fills (MD:47-48)
            np.testing.assert_array_equal(env.last_arrivals.astype(np.uint8), g["arrivals"][k], err_msg=f"arrivals step {k}")
            np.testing.assert_array_equal(np.asarray(env.last_fills, dtype=np.uint8), g["fills"][k], err_msg=f"fills step {k}")
        np.testing.assert_array_equal(obs, g["obs"][k], err_msg=f"obs step {k}")
        np.testing.assert_array_equal(rew, g["rewards"][k], err_msg=f"rewards step {k}")
        assert bool(done[0]) == bool(g["done"][k])


def _as_env(n):
    return OracleConfig(
        num_trajectories=n, n_steps=200, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0,
        arrival="poisson", intensity=(140.0, 140.0), fill_exponent=1.5, dynamics="limit", reward="pnl",
        initial_inventory=0, max_inventory=200, seed=50, normalise_action_space=False, normalise_observation_space=False,
    )


# Published by the reference: notebooks/Test_1_-_replicate_AS_original_results.ipynb:219-231 (gamma=0.1)
# and :338-350 (gamma=0.01): mean spread, mean PnL, std PnL, mean terminal inventory, std terminal inventory.
# The notebook rolls out twice without reseeding (cells 9 and 10); the table is the SECOND rollout.
NB1_TABLE = {
    0.1: (1.49177, 64.872139, 6.692567, 0.201, 2.893544),
    0.01: (1.349009, 68.754417, 8.720076, 0.23, 5.095989),
}


@pytest.mark.parametrize("gamma", [0.1, 0.01])
def test_published_avellaneda_stoikov_table_to_every_digit(gamma):
    cfg = _as_env(1000)
    env = OracleEnv(cfg, NumpyProtocolNoise(cfg.seed))
    policy = lambda obs: avellaneda_stoikov_action(cfg, gamma, obs)  # noqa: E731
    rollout(env, policy)
    table = results_table(*rollout(env, policy))
    for got, want in zip(table, NB1_TABLE[gamma]):
        assert round(float(got), 6) == pytest.approx(want, abs=5.1e-7), (table, NB1_TABLE[gamma])
