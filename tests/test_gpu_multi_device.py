"""MultiDeviceTradingEnvironment: one logical environment sharded over devices from one host process.  On the one-GPU test
box the shards share device 0 (the code path - one handle, stream and host thread per shard, rows concatenated in global lane
order - is the same); what is asserted is that sharding is invisible: every output equals that of ONE environment holding
all the lanes, bit for bit, including random initial inventories drawn on the host and the SB3 adapter's auto-reset."""
import numpy as np
import pytest

from mbt_gym_amd.gym.MultiDeviceTradingEnvironment import MultiDeviceTradingEnvironment
from mbt_gym_amd.gym.StableBaselinesTradingEnvironment import StableBaselinesTradingEnvironment
from oracle.mbt_oracle import OracleConfig
from tests.env_factory import make_env

pytestmark = pytest.mark.gpu


def _cfg(n, **kw):
    base = dict(num_trajectories=n, n_steps=25, terminal_time=1.0, midprice="ou", ou_level=100.0, ou_speed=0.02, volatility=2.0, initial_price=100.0,
                arrival="hawkes", intensity=(20.0, 15.0), hawkes_jump=20.0, hawkes_speed=15.0, fill_exponent=1.5, dynamics="limit_and_market",
                market_half_spread=0.4, reward="cjmm", phi=0.01, alpha=0.05, initial_inventory=(-3, 4), max_inventory=5, seed=77,
                normalise_action_space=False, normalise_observation_space=False)
    base.update(kw)
    return OracleConfig(**base)


def _sharded(cfg, devices):
    import copy

    def make_shard(num_trajectories, device, trajectory_offset):
        shard_cfg = copy.copy(cfg)
        shard_cfg.num_trajectories = num_trajectories
        return make_env(shard_cfg, device=device, trajectory_offset=trajectory_offset)

    return MultiDeviceTradingEnvironment(make_shard, cfg.num_trajectories, devices=devices, seed=cfg.seed)


@pytest.mark.parametrize("n,devices", [(5000, [0, 0, 0]), (1024, [0, 0]), (3000, [0, 0, 0, 0, 0])])
@pytest.mark.parametrize("kw", [dict(), dict(normalise_action_space=True, normalise_observation_space=True, initial_inventory=1, reward="running"),
                                dict(dynamics="speed", arrival="none", midprice="bm", impact="temp_perm", temporary_impact=0.02, permanent_impact=0.01,
                                     impact_step_size=1 / 25, reward="cjoe", initial_inventory=12, max_inventory=1000, volatility=0.3)])
def test_sharding_over_devices_is_invisible(n, devices, kw):
    cfg = _cfg(n, **kw)
    single, multi = make_env(cfg), _sharded(cfg, devices)
    assert len(multi.shards) == min(len(devices), -(-n // 1024)) and multi.num_trajectories == n
    rng = np.random.default_rng(1)
    for episode in range(2):  # the second episode: fresh host draws (initial inventories) on both sides, Philox stream continuing
        obs_s, obs_m = single.reset(), multi.reset()
        np.testing.assert_array_equal(obs_m, obs_s)
        for k in range(cfg.n_steps):
            if cfg.dynamics == "speed":
                action = rng.uniform(-0.2, 0.4, size=(n, 1)).astype(np.float32) * (1.0 if cfg.normalise_action_space else 5.0)
            elif cfg.normalise_action_space:
                action = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)
            else:
                action = np.concatenate([rng.uniform(0, 1.5, size=(n, 2)), rng.choice([0.0, 1.0], p=[0.9, 0.1], size=(n, 2))], axis=1).astype(np.float32)
            o_s, r_s, d_s, i_s = single.step(action)
            o_m, r_m, d_m, i_m = multi.step(action)
            np.testing.assert_array_equal(o_m, o_s, err_msg=f"episode {episode} step {k}")
            np.testing.assert_array_equal(r_m, r_s)
            np.testing.assert_array_equal(d_m, d_s)
            assert len(i_m) == len(i_s) == n
        assert d_m.all()
        np.testing.assert_array_equal(multi.state, single.state)
        sums_s, sums_m = single.episode_return_sums(), multi.episode_return_sums()
        assert sums_m[2] == sums_s[2] == n and sums_m[0] == pytest.approx(sums_s[0], rel=1e-9, abs=1e-6)
    single.close(), multi.close()


def test_fused_rollout_and_the_sb3_adapter_over_shards():
    from mbt_gym_amd.agents.BaselineAgents import FixedActionAgent

    cfg = _cfg(4000, dynamics="limit", reward="pnl", initial_inventory=0, arrival="poisson", intensity=(100.0, 100.0), midprice="bm")
    single, multi = make_env(cfg), _sharded(cfg, [0, 0, 0])
    fixed = np.array([0.6, 0.7], np.float32)
    single.reset(), multi.reset()
    out_s = single.rollout(FixedActionAgent(fixed, single))
    out_m = multi.rollout(FixedActionAgent(fixed, multi.shards[0]))
    for a, b in zip(out_m[:3], out_s[:3]):
        np.testing.assert_array_equal(a, b)
    assert out_m[3:] == out_s[3:]
    # the VecEnv adapter on top: step_wait auto-resets when the shared clock ends the episode (SBE:28-37)
    vec_s, vec_m = StableBaselinesTradingEnvironment(single), StableBaselinesTradingEnvironment(multi)
    assert vec_m.num_envs == vec_s.num_envs == 4000
    np.testing.assert_array_equal(vec_m.reset(), vec_s.reset())
    action = np.tile(fixed, (4000, 1))
    for k in range(cfg.n_steps + 3):
        vec_s.step_async(action), vec_m.step_async(action)
        o_s, r_s, d_s, i_s = vec_s.step_wait()
        o_m, r_m, d_m, i_m = vec_m.step_wait()
        np.testing.assert_array_equal(o_m, o_s)
        np.testing.assert_array_equal(r_m, r_s)
        np.testing.assert_array_equal(d_m, d_s)
        if d_s[0]:
            np.testing.assert_array_equal(i_m[17]["terminal_observation"], i_s[17]["terminal_observation"])
    vec_s.close(), vec_m.close()


def test_a_random_callable_initial_inventory_is_drawn_once_for_all_shards():
    """TE:275-279: a callable `initial_inventory` is evaluated ONCE per reset and every lane gets that value.  Sharded, each
    shard used to evaluate it for itself - a random callable then gave lanes that depended on the number of devices."""
    draws = np.random.default_rng(8)
    calls = []

    def random_inventory():
        calls.append(1)
        return float(draws.integers(-3, 4))

    cfg = _cfg(3072, initial_inventory=0, reward="running")
    env = _sharded(cfg, [0, 0, 0])
    env.initial_inventory = random_inventory
    for _ in range(6):
        before = len(calls)
        obs = env.reset()
        assert len(calls) == before + 1
        assert np.unique(obs[:, 1]).size == 1  # one value, in every shard
    env.close()
