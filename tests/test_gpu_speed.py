"""Trading-with-speed (optimal execution) dynamics beyond the fixture parity of test_gpu_parity.py: the quad noise
stream, Philox mode tied to injected mode, sharding / ragged sizes, the fused rollout against the step loop, and the
Cartea-Jaimungal optimal-execution schedule against an exact expectation."""
import numpy as np
import pytest

from mbt_gym_amd import _native
from mbt_gym_amd.agents.BaselineAgents import CarteaJaimungalOeAgent, FixedActionAgent
from oracle.mbt_oracle import InjectedNoise, OracleEnv
from oracle.philox_ref import philox4x32_10
from tests.env_factory import make_env
from tests.golden_io import load_case

pytestmark = pytest.mark.gpu
SPEED_CASES = ["speed_temp_perm_cjoe", "speed_power_running", "speed_temp_transient_pnl", "speed_transient_pnl"]


def _quad_normals_reference(seed, offset, step, n):
    """Quad stream of csrc/speed_kernel.hpp: ctr = (quad.lo, quad.hi, step, 3); words (0,1) / (2,3) -> two Box-Muller pairs;
    the quad of a 1024-lane tile's slot j holds lanes j, j + 256, j + 512, j + 768."""
    n_pad = -(-n // 1024) * 1024
    quads = np.arange(n_pad // 4, dtype=np.uint64) + np.uint64(offset // 4)
    w = philox4x32_10(((quads & np.uint64(0xFFFFFFFF)).astype(np.uint32), (quads >> np.uint64(32)).astype(np.uint32),
                       np.full(quads.shape, step, np.uint32), np.full(quads.shape, 3, np.uint32)), (seed & 0xFFFFFFFF, seed >> 32))
    z = np.empty(n_pad)
    for pair, (wr, wt) in enumerate(((w[0], w[1]), (w[2], w[3]))):
        r = np.sqrt(-2.0 * np.log(((wr >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0**-24))
        th = 2.0 * np.pi * (wt >> np.uint32(8)).astype(np.float64) * 2.0**-24
        local = np.arange(n_pad // 4)
        lane0 = (local // 256) * 1024 + local % 256
        z[lane0 + 256 * (2 * pair)], z[lane0 + 256 * (2 * pair + 1)] = r * np.cos(th), r * np.sin(th)
    return z[:n]


def test_quad_stream_matches_the_restatement():
    for seed, offset, step, n in [(31, 0, 0, 1000), (2**40 + 5, 1 << 20, 77, 4099), (9, 3072, 5, 2500)]:
        z = _native.rng_fill_quad(seed, offset, step, n)
        np.testing.assert_allclose(z, _quad_normals_reference(seed, offset, step, n), rtol=0, atol=3e-5)
    assert abs(float(_native.rng_fill_quad(1, 0, 0, 1 << 16).std()) - 1.0) < 0.02


@pytest.mark.parametrize("name", SPEED_CASES)
def test_philox_mode_equals_injected_mode_and_the_oracle(name):
    cfg, g = load_case(name)
    cfg.seed, steps, n = 99, 30, cfg.num_trajectories
    actions = g["actions"][:steps]
    z = np.stack([_native.rng_fill_quad(99, 0, k, n) for k in range(steps)])
    env_p, env_i = make_env(cfg, noise="philox"), make_env(cfg, noise="injected")
    oracle = OracleEnv(cfg, InjectedNoise(np.zeros((steps, n, 2)), np.zeros((steps, n, 2)), z))
    env_p.reset(), env_i.reset(), oracle.reset()
    for k in range(steps):
        env_i.set_noise(None, None, z[k])
        op, rp, _, _ = env_p.step(actions[k])
        oi, ri, _, _ = env_i.step(actions[k])
        oo, ro, _ = oracle.step(actions[k].astype(np.float64))
        np.testing.assert_array_equal(op, oi)
        np.testing.assert_array_equal(rp, ri)
        np.testing.assert_allclose(op, oo, rtol=2e-6, atol=3e-4)
        assert np.all(np.abs(rp - ro) <= 1e-5 + 2e-6 * np.abs(ro))
    env_p.close()
    env_i.close()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 258, 1030])
def test_ragged_sizes_and_sharding(n):
    cfg, _ = load_case("speed_temp_perm_cjoe")
    cfg.num_trajectories, cfg.seed = n, 4
    env = make_env(cfg)
    env.reset()
    act = np.full((n, 1), 1.5, np.float32)
    total = np.zeros(n)
    for _ in range(10):
        obs, rew, dones, _ = env.step(act)
        assert obs.shape == (n, 5) and rew.shape == (n,)
        total += rew
    assert env.episode_return_sums()[0] == pytest.approx(total.sum(), abs=1e-3)
    cfg.num_trajectories = 1032
    big = make_env(cfg)
    big.reset()
    for _ in range(10):
        obs_big, _, _, _ = big.step(np.full((1032, 1), 1.5, np.float32))
    np.testing.assert_array_equal(obs, obs_big[:n])
    if n == 1030:  # a shard starting on the second tile (global lane 1024) reproduces lanes 1024.. of the whole run
        cfg.num_trajectories = 6
        tail = make_env(cfg, trajectory_offset=1024)
        tail.reset()
        for _ in range(10):
            obs_tail, _, _, _ = tail.step(np.full((6, 1), 1.5, np.float32))
        np.testing.assert_array_equal(obs_tail, obs_big[1024:1030])
        tail.close()
    env.close()
    big.close()


@pytest.mark.parametrize("name", SPEED_CASES)
def test_fixed_speed_rollout_equals_the_step_loop(name):
    cfg, g = load_case(name)
    cfg.seed = 12
    steps = cfg.n_steps
    env_a, env_b = make_env(cfg), make_env(cfg)
    agent = FixedActionAgent(np.array([0.8], np.float32), env_a)
    env_a.reset()
    obs_r, act_r, rew_r, n_done, done = env_a.rollout(agent)
    assert n_done == steps and done
    obs = [env_b.reset()]
    for k in range(steps):
        o, r, d, _ = env_b.step(agent.get_action(None))
        np.testing.assert_array_equal(o, obs_r[k + 1])
        np.testing.assert_array_equal(r, rew_r[k])
    np.testing.assert_array_equal(obs[0], obs_r[0])
    assert np.all(act_r == np.float32(0.8)) and d[0]
    np.testing.assert_array_equal(env_a.state, env_b.state)
    env_a.close()
    env_b.close()


def test_cartea_jaimungal_execution_schedule_known_answer():
    """2^20 lanes liquidating 10 units with the closed-form CJ speed through the fused rollout (time-table policy).
    The rewards are affine in the midprice noise, so the expected total reward equals the total reward of the
    NOISE-FREE float64 oracle run; the Monte-Carlo mean must hit it within 5 standard errors."""
    cfg, _ = load_case("speed_temp_perm_cjoe")
    n = 1 << 20
    cfg.num_trajectories, cfg.seed, cfg.drift = n, 2025, 0.0
    env = make_env(cfg)
    agent = CarteaJaimungalOeAgent(phi=0.01, alpha=0.05, env=env)
    schedule = agent.schedule()
    assert schedule.shape == (cfg.n_steps + 1, 1) and np.all(schedule[:-1] < 0)  # selling a long position
    env.track_lane_returns(True)
    env.reset()
    _, _, _, steps, done = env.rollout(agent, record=False)
    assert steps == cfg.n_steps and done
    total, total_sq, count = env.episode_return_sums()
    mean, std = total / count, np.sqrt(total_sq / count - (total / count) ** 2)
    cfg.num_trajectories = 1
    k = cfg.n_steps
    quiet = OracleEnv(cfg, InjectedNoise(np.zeros((k, 1, 2)), np.zeros((k, 1, 2)), np.zeros((k, 1))))
    quiet.reset()
    exact = sum(float(quiet.step(schedule[j].reshape(1, 1).astype(np.float64))[1][0]) for j in range(k))
    assert mean == pytest.approx(exact, abs=5 * std / np.sqrt(n) + 1e-4), (mean, exact, std)
    q_T = env.state[:, 1]
    assert np.allclose(q_T, quiet.state[0, 1], atol=1e-4)  # the inventory path is deterministic
    # the same schedule through the host agent and the step path, bit for bit, on a small batch
    cfg.num_trajectories = 64
    env_s, env_r = make_env(cfg), make_env(cfg)
    agent_s, agent_r = CarteaJaimungalOeAgent(0.01, 0.05, env_s), CarteaJaimungalOeAgent(0.01, 0.05, env_r)
    env_r.reset()
    obs_r, act_r, rew_r, _, _ = env_r.rollout(agent_r)
    obs = env_s.reset()
    for j in range(k):
        a = agent_s.get_action(obs)
        np.testing.assert_array_equal(a, act_r[j])
        obs, r, _, _ = env_s.step(a)
        np.testing.assert_array_equal(obs, obs_r[j + 1])
        np.testing.assert_array_equal(r, rew_r[j])
    for e in (env, env_s, env_r):
        e.close()
