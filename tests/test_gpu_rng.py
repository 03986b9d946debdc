"""The production noise path: the device Philox4x32-10 against Random123's known answers and the NumPy restatement;
Philox mode tied bit-for-bit to the parity-tested injected mode; invariance to sharding of the trajectory axis."""
import numpy as np
import pytest

from mbt_gym_amd import _native
from oracle.mbt_oracle import InjectedNoise, OracleEnv
from oracle.philox_ref import pair_stream_noise
from tests.env_factory import make_env
from tests.golden_io import load_case
from tests.test_philox_oracle import KAT

pytestmark = pytest.mark.gpu


def test_device_philox_known_answers():
    for ctr, key, want in KAT:
        assert tuple(_native.philox4x32_10(ctr, key)) == want


@pytest.mark.parametrize("offset,n,step,seed", [(0, 1000, 0, 50), (1 << 20, 4097, 12345, 2**40 + 17), (2**33, 64, 3, 7)])
def test_device_stream_matches_the_restatement(offset, n, step, seed):
    u_arr, u_fill, z = _native.rng_fill(seed, offset, step, n)
    r_arr, r_fill, r_z = pair_stream_noise(seed, offset, step, n)
    np.testing.assert_array_equal(u_arr, r_arr)
    np.testing.assert_array_equal(u_fill, r_fill)
    # Box-Muller on v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32 vs float64 libm
    # (v_sin_f32 / v_cos_f32 carry ~2e-6 absolute error, scaled by the radius r <= 5.8)
    np.testing.assert_allclose(z, r_z, rtol=0, atol=3e-5)


def _roll(env, actions, noise=None):
    out = []
    env.reset()
    for k, a in enumerate(actions):
        if noise is not None:
            env.set_noise(*[x[k] for x in noise])
        obs, rew, dones, _ = env.step(a)
        out.append((obs.copy(), rew.copy(), bool(dones[0])))
    return out


@pytest.mark.parametrize("name", ["as_limit_pnl", "hawkes_ou", "limit_and_market", "cjp_cjmm", "default_normalised",
                                  "gbm_nonlinear_touch", "bmjump_exputility", "oujump_hawkes_running", "constant_midprice", "exo_fill_bm_poisson",
                                  "exo_fill_hawkes_market", "exo_fill_normalised"])
def test_philox_mode_equals_injected_mode_on_the_same_draws(name):
    """The kernel body is shared: feeding the injected-noise instantiation with the draws the Philox instantiation
    makes must give bit-identical states and rewards - this carries the parity result over to production mode."""
    cfg, g = load_case(name)
    steps = min(40, g["actions"].shape[0])
    seed = 1234
    cfg.seed = seed
    actions = g["actions"][:steps]
    draws = [_native.rng_fill(seed, 0, k, cfg.num_trajectories) for k in range(steps)]
    noise = [np.stack(x) for x in zip(*draws)]
    env_p = make_env(cfg, noise="philox")
    env_i = make_env(cfg, noise="injected")
    got_p, got_i = _roll(env_p, actions), _roll(env_i, actions, noise)
    for (op, rp, dp), (oi, ri, di) in zip(got_p, got_i):
        np.testing.assert_array_equal(op, oi)
        np.testing.assert_array_equal(rp, ri)
        assert dp == di
    # and the float64 oracle on those draws: decisions exact, rewards within 1e-5
    oracle = OracleEnv(cfg, InjectedNoise(*noise))
    oracle.reset()
    for k in range(steps):
        o_obs, o_rew, _ = oracle.step(actions[k].astype(np.float64))
        if not cfg.normalise_observation_space:
            np.testing.assert_array_equal(got_p[k][0][:, 1].astype(np.float64), o_obs[:, 1])
        err = np.abs(got_p[k][1] - o_rew)
        # as in test_gpu_parity.py: 1e-5 (+ float32 rounding of a large reward); 1.2e-4 where the clip of TE:283-289 fires
        tol = {"limit_and_market": 1.2e-4, "exo_fill_hawkes_market": 1.2e-4}.get(name, 1e-5) + 1e-6 * np.abs(o_rew)
        assert np.all(err <= tol), err.max()
    env_p.close()
    env_i.close()


def test_results_do_not_depend_on_how_the_trajectory_axis_is_sharded():
    cfg, g = load_case("as_limit_pnl")
    cfg.num_trajectories, cfg.seed = 4096, 99
    action = np.tile(np.array([[0.5, 0.9]], np.float32), (4096, 1))
    whole = make_env(cfg)
    ref = _roll(whole, [action] * 6)
    cfg.num_trajectories = 2048
    for shard in range(2):
        part = make_env(cfg, trajectory_offset=shard * 2048)
        got = _roll(part, [action[:2048]] * 6)
        for (o, r, _), (ow, rw, _) in zip(got, ref):
            np.testing.assert_array_equal(o, ow[shard * 2048:(shard + 1) * 2048])
            np.testing.assert_array_equal(r, rw[shard * 2048:(shard + 1) * 2048])
        part.close()
    whole.close()


def test_reset_does_not_reseed_but_seed_does():
    """Like the reference (no reseed on reset): a second episode continues the stream; seed() restarts it."""
    cfg, _ = load_case("as_limit_pnl")
    cfg.num_trajectories, cfg.seed = 512, 5
    action = np.tile(np.array([[0.7, 0.7]], np.float32), (512, 1))
    env = make_env(cfg)
    first = _roll(env, [action] * 3)
    second = _roll(env, [action] * 3)
    assert not np.array_equal(first[0][0], second[0][0])
    env.seed(5)
    again = _roll(env, [action] * 3)
    for (a, ra, _), (b, rb, _) in zip(first, again):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(ra, rb)
    env.close()


@pytest.mark.parametrize("n", [1, 3, 255])
def test_ragged_sizes(n):
    """Odd and tiny lane counts (the kernel works on pairs; the pad lane must never leak into results)."""
    cfg, g = load_case("as_limit_pnl")
    cfg.num_trajectories, cfg.seed = n, 8
    env = make_env(cfg)
    env.track_lane_returns(True)
    action = np.tile(np.array([[0.4, 0.6]], np.float32), (n, 1))
    env.reset()
    total = np.zeros(n, np.float64)
    for _ in range(20):
        obs, rew, dones, infos = env.step(action)
        assert obs.shape == (n, 4) and rew.shape == (n,) and dones.shape == (n,)
        total += rew
    sums = env.episode_return_sums()
    assert sums[2] == n
    assert sums[0] == pytest.approx(total.sum(), abs=1e-3)
    assert sums[1] == pytest.approx(float((total**2).sum()), rel=1e-4, abs=1e-3)
    # lanes of a size-n run are the first n lanes of any larger run
    cfg.num_trajectories = 256
    big = make_env(cfg)
    big.reset()
    for _ in range(20):
        obs_big, _, _, _ = big.step(np.tile(np.array([[0.4, 0.6]], np.float32), (256, 1)))
    np.testing.assert_array_equal(obs, obs_big[:n])
    env.close()
    big.close()


def test_stream_statistics_and_independence():
    """The pair layout uses every bit of two Philox blocks: the four uniforms of a lane take the top 24 bits of its block's
    words, the two normals of the pair come from the LOW bytes of both blocks.  Moments of each output and the
    correlations between them (same lane, partner lane, next step, neighbouring lane) at 2^22 draws: all within 5 standard
    errors of their ideal values."""
    n = 1 << 22
    ua, uf, z = _native.rng_fill(2024, 0, 0, n)
    ua2, uf2, z2 = _native.rng_fill(2024, 0, 1, n)
    se = 1 / np.sqrt(n)
    for u in (ua[:, 0], ua[:, 1], uf[:, 0], uf[:, 1]):
        u = u.astype(np.float64)
        assert abs(u.mean() - 0.5) < 5 * se / np.sqrt(12)
        assert abs(u.var() - 1 / 12) < 5 * se * np.sqrt(1 / 180)
        assert abs(np.histogram(u, bins=64, range=(0, 1))[0] / n - 1 / 64).max() < 5 * np.sqrt(1 / 64 / n)
    zz = z.astype(np.float64)
    assert abs(zz.mean()) < 5 * se and abs(zz.var() - 1) < 5 * se * np.sqrt(2)
    assert abs((zz**3).mean()) < 5 * se * np.sqrt(15) and abs((zz**4).mean() - 3) < 5 * se * np.sqrt(96)
    assert 4.9 < np.abs(zz).max() < 6.5  # 24-bit radius: the tail reaches ~5.8 sigma
    def corr(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        return float(np.mean((a - a.mean()) * (b - b.mean())) / (a.std() * b.std()))
    partner = np.arange(n).reshape(-1, 2, 256)[:, ::-1, :].reshape(-1)  # lane g <-> g +- 256 inside its 512-lane tile
    pairs = [(zz, ua[:, 0]), (zz, ua[:, 1]), (zz, uf[:, 0]), (zz, uf[:, 1]), (ua[:, 0], ua[:, 1]), (ua[:, 0], uf[:, 0]), (uf[:, 0], uf[:, 1]),
             (zz, zz[partner]), (zz, ua[partner, 0]), (zz, uf[partner, 1]), (zz, z2), (ua[:, 0], ua2[:, 0]), (uf[:, 1], uf2[:, 1]),
             (zz[:-1], zz[1:]), (ua[:-1, 0], ua[1:, 0]), (zz**2, zz[partner] ** 2)]
    for k, (a, b) in enumerate(pairs):
        assert abs(corr(a, b)) < 5 * se, (k, corr(a, b))
