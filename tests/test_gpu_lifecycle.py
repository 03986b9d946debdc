"""Handle lifecycle on the device: environments of every kind created, used and destroyed in a loop give their memory
back; a closed environment refuses work instead of touching freed memory; close() is idempotent."""
import numpy as np
import pytest

from mbt_gym_amd import _native
from tests.env_factory import make_env
from tests.random_configs import random_config, random_speed_config

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch

    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _exercise(cfg, rng):
    from mbt_gym_amd.agents.BaselineAgents import FixedActionAgent
    from oracle.mbt_oracle import action_bounds

    env = make_env(cfg)
    lo, hi = action_bounds(cfg)
    if cfg.dynamics == "touch":
        fixed = np.array([1.0, 0.0], np.float32)
    elif cfg.normalise_action_space:
        fixed = np.full(cfg.action_dim, 0.2, np.float32)
    else:
        fixed = (0.3 * hi).astype(np.float32)
    env.track_lane_returns(True)
    env.record_events(True)
    obs = env.reset()
    action = np.tile(fixed, (cfg.num_trajectories, 1))
    for _ in range(3):
        env.step(action)
    env.reset()
    obs_r, act_r, rew_r, steps, done = env.rollout(FixedActionAgent(fixed, env))  # trajectory staging buffers, policy upload
    assert done and obs_r.shape[0] == steps + 1
    env.episode_return_sums()
    env.close()
    env.close()  # idempotent


def test_create_use_destroy_returns_device_memory():
    rng = np.random.default_rng(2024)
    configs = []
    for i in range(60):
        n = int(rng.choice([64, 1500, 40000]))
        configs.append(random_speed_config(rng, n) if i % 4 == 3 else random_config(rng, n))
    for cfg in configs[:6]:  # first use pays for one-off allocations (module load, HIP's own pools)
        _exercise(cfg, rng)
    before = _free_bytes()
    for cfg in configs:
        _exercise(cfg, rng)
    after = _free_bytes()
    assert before - after < 32 << 20, f"{(before - after) / 2**20:.1f} MiB of device memory did not come back after 60 environments"


def test_user_plugin_environments_come_and_go():
    """The run-time compiled route: modules are cached per process (one compilation per distinct source), handles are not."""
    from tests.golden_io import load_case

    cfg, _ = load_case("user_fill_and_reward")
    for _ in range(3):
        make_env(cfg).close()
    before = _free_bytes()
    for _ in range(25):
        env = make_env(cfg)
        env.reset()
        env.step(np.full((cfg.num_trajectories, cfg.action_dim), 0.3, np.float32))
        env.close()
    assert before - _free_bytes() < 16 << 20


def test_a_closed_environment_refuses_work():
    rng = np.random.default_rng(5)
    cfg = random_config(rng, 256)
    env = make_env(cfg)
    env.reset()
    env.close()
    action = np.zeros((256, cfg.action_dim), np.float32)
    with pytest.raises(_native.NativeError):
        env.step(action)
    with pytest.raises(_native.NativeError):
        env.reset()
    env.close()


def test_two_host_threads_each_with_its_own_environment():
    """The ABI's threading contract is one host thread per handle: two threads stepping two environments at the same time
    (own streams, thread-local error strings, shared module / library state) get exactly what each gets alone."""
    import threading

    from oracle.mbt_oracle import OracleConfig

    def cfg(seed):
        return OracleConfig(num_trajectories=5000, n_steps=120, terminal_time=1.0, midprice="ou", ou_level=100.0, ou_speed=0.02, volatility=2.0,
                            initial_price=100.0, arrival="hawkes", intensity=(20.0, 15.0), hawkes_jump=20.0, hawkes_speed=40.0, fill_exponent=1.5,
                            dynamics="limit_and_market", market_half_spread=0.4, reward="running", phi=0.01, alpha=0.05, initial_inventory=1,
                            max_inventory=4, seed=seed, normalise_action_space=False, normalise_observation_space=False)

    action = np.tile(np.array([[0.4, 0.6, 0.0, 1.0]], np.float32), (5000, 1))

    def run(seed, out):
        env = make_env(cfg(seed))
        env.reset()
        total = np.zeros(5000)
        for _ in range(120):
            obs, rew, dones, _ = env.step(action)
            total += rew
        out[seed] = (obs.copy(), total)
        env.close()

    alone, together = {}, {}
    for seed in (11, 12):
        run(seed, alone)
    threads = [threading.Thread(target=run, args=(seed, together)) for seed in (11, 12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for seed in (11, 12):
        np.testing.assert_array_equal(together[seed][0], alone[seed][0])
        np.testing.assert_array_equal(together[seed][1], alone[seed][1])
    assert not np.array_equal(alone[11][0], alone[12][0])


def test_empty_and_single_lane_batches():
    """num_trajectories = 0 is refused at construction (the reference builds empty arrays and fails later, in its first
    reduction); one lane - the reference's default - runs: the partner lane of the pair and the rest of the tile are padding."""
    rng = np.random.default_rng(8)
    cfg = random_config(rng, 1)
    cfg.normalise_action_space = cfg.normalise_observation_space = False
    cfg.dynamics, cfg.fill = "limit", "exponential"
    env = make_env(cfg)
    obs = env.reset()
    assert obs.shape == (1, env.observation_dim)
    obs, rew, dones, infos = env.step(np.full((1, 2), 0.3, np.float32))
    assert obs.shape == (1, env.observation_dim) and rew.shape == (1,) and dones.shape == (1,) and infos == {}  # TE:320-321: a dict for N = 1
    env.close()
    cfg.num_trajectories = 0
    with pytest.raises((_native.NativeError, AssertionError, ValueError)):
        make_env(cfg)
