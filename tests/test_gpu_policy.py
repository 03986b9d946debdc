"""Learned policies evaluated inside the kernels (csrc/policy_mlp.hpp): a linear map on the vector unit, a [D -> H -> H -> A]
MLP on the matrix cores (v_mfma_f32_16x16x16_f16).

Numerics contract, stated here and in the header: operands (observation, weights, hidden activations) are rounded to fp16,
products and sums are fp32, tanh is 1 - 2 / (1 + 2^(2 log2(e) x)) on the hardware exp2 / rcp.  The NumPy restatement below
rounds where the kernel rounds; actions agree with it to 2e-3 (fp32 summation order, one-ulp fp16 rounding flips of
hidden units) and with the plain fp32 network to 2e-2.  The fused rollout is BIT-identical to "policy kernel, step kernel"
repeated, and the environment under a learned policy still matches the float64 oracle (decisions exact, rewards 1e-5) when the
oracle is fed the device's own actions and draws."""
import numpy as np
import pytest

from mbt_gym_amd import _native
from oracle.mbt_oracle import InjectedNoise, OracleConfig, OracleEnv
from tests.env_factory import make_env

pytestmark = pytest.mark.gpu


def _cfg(n, **kw):
    base = dict(num_trajectories=n, n_steps=40, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0, arrival="poisson",
                intensity=(140.0, 140.0), fill_exponent=1.5, dynamics="limit", reward="pnl", initial_inventory=0, max_inventory=20, seed=5,
                normalise_action_space=True, normalise_observation_space=True)
    base.update(kw)
    return OracleConfig(**base)


def _random_mlp(rng, d, hidden, a, scale=1.0):
    def layer(out, inp):
        return (rng.normal(0, scale / np.sqrt(inp), size=(out, inp)).astype(np.float32), rng.normal(0, 0.1, size=out).astype(np.float32))
    return [layer(hidden, d), layer(hidden, hidden), layer(a, hidden)]


def _f16(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def mlp_reference(obs, layers, activation, lo, hi, emulate_fp16=True):
    """NumPy restatement of policy_mlp.hpp: fp16 operands (when emulating), fp32 accumulation, clip to the action space."""
    r = _f16 if emulate_fp16 else (lambda a: np.asarray(a, dtype=np.float32))
    act = (lambda x: np.tanh(x)) if activation == "tanh" else (lambda x: np.maximum(x, 0.0))
    (w1, b1), (w2, b2), (w3, b3) = layers
    x = r(np.concatenate([obs, np.ones((len(obs), 1), np.float32)], axis=1))
    w1p = r(np.concatenate([w1, b1[:, None]], axis=1))  # the first bias rides on the constant-one feature: it is fp16 too
    h1 = r(act(x.astype(np.float64) @ w1p.T.astype(np.float64)).astype(np.float32))
    h2 = r(act(h1.astype(np.float64) @ r(w2).T.astype(np.float64) + b2).astype(np.float32))
    out = (h2.astype(np.float64) @ r(w3).T.astype(np.float64) + b3).astype(np.float32)
    return np.clip(out, lo, hi)


def _action_space(env):
    return env.action_space.low.astype(np.float32), env.action_space.high.astype(np.float32)


@pytest.mark.parametrize("activation", ["tanh", "relu"])
@pytest.mark.parametrize("kw,hidden", [(dict(), 64), (dict(dynamics="limit_and_market", market_half_spread=0.4), 64),
                                       (dict(arrival="hawkes", intensity=(10.0, 10.0), hawkes_speed=20.0, midprice="ou", ou_level=100.0, ou_speed=0.02), 32),
                                       (dict(normalise_action_space=False), 48)])
def test_mlp_policy_kernel_matches_the_numpy_restatement(kw, hidden, activation):
    n = 3000  # not a multiple of the 512-lane tile: pad rows are evaluated and never reported
    cfg = _cfg(n, **kw)
    env = make_env(cfg)
    rng = np.random.default_rng(11)
    raw = not cfg.normalise_observation_space
    layers = _random_mlp(rng, env.observation_dim, hidden, env.action_dim, scale=0.05 if raw else 1.5)
    if not cfg.normalise_action_space:  # raw depths live in [0, max_depth]: centre the outputs inside the Box instead of on its lower edge
        layers[2] = (layers[2][0], layers[2][1] + np.float32(1.5))
    policy = _native.mlp_policy(layers, activation)
    env.reset()
    warm = np.tile(np.array([[0.0] * env.action_dim], np.float32), (n, 1)) if not raw else np.tile(np.array([[0.5] * env.action_dim], np.float32), (n, 1))
    for _ in range(7):  # move away from the reset state so that rows differ
        obs, _, _, _ = env.step(warm)
    env.policy_device(policy)
    env.synchronize()
    import torch

    got = torch.as_tensor(env.action_device, device="cuda").cpu().numpy()
    lo, hi = _action_space(env)
    want = mlp_reference(obs, layers, activation, lo, hi)
    plain = mlp_reference(obs, layers, activation, lo, hi, emulate_fp16=False)
    assert got.shape == want.shape == (n, env.action_dim)
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-3 * scale)
    np.testing.assert_allclose(got, plain, rtol=0, atol=2e-2 * scale)
    assert np.std(want) > 0.01 * (hi[0] - lo[0]), "the test network must not be saturated or constant"
    env.close()


def test_linear_policy_is_fp32_exact_up_to_summation_order():
    n = 2048
    cfg = _cfg(n)
    env = make_env(cfg)
    rng = np.random.default_rng(3)
    w, b = rng.normal(0, 0.5, size=(2, 4)).astype(np.float32), rng.normal(0, 0.1, size=2).astype(np.float32)
    policy = _native.linear_policy(w, b)
    env.reset()
    for _ in range(5):
        obs, _, _, _ = env.step(np.zeros((n, 2), np.float32))
    env.policy_device(policy)
    env.synchronize()
    import torch

    got = torch.as_tensor(env.action_device, device="cuda").cpu().numpy()
    want = np.clip(obs.astype(np.float64) @ w.T.astype(np.float64) + b, -1.0, 1.0)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    env.close()


@pytest.mark.parametrize("kind", ["mlp_tanh", "mlp_relu", "linear"])
@pytest.mark.parametrize("kw", [dict(), dict(dynamics="limit_and_market", market_half_spread=0.4, reward="running", phi=0.01, alpha=0.05, max_inventory=4)])
def test_fused_rollout_with_a_learned_policy_is_bit_identical_to_the_policy_and_step_kernels(kw, kind):
    n = 4096 + 300
    cfg = _cfg(n, **kw)
    fused, loop = make_env(cfg), make_env(cfg)
    rng = np.random.default_rng(23)
    d, a = fused.observation_dim, fused.action_dim
    if kind == "linear":
        policy = _native.linear_policy(rng.normal(0, 0.8, size=(a, d)).astype(np.float32), rng.normal(0, 0.2, size=a).astype(np.float32))
    else:
        policy = _native.mlp_policy(_random_mlp(rng, d, 64, a, scale=1.5), kind.split("_")[1])
    fused.reset(), loop.reset()
    steps, done = fused.rollout_device(policy)
    assert (steps, done) == (cfg.n_steps, True)
    for k in range(cfg.n_steps):
        loop.policy_device(policy)
        finished = loop.step_device()
    assert finished
    np.testing.assert_array_equal(fused.state, loop.state)
    np.testing.assert_array_equal(fused.observation_host(), loop.observation_host())
    # (the sums are accumulated per step by the step kernel and per lane by the rollout: same values, different summation order)
    assert fused.episode_return_sums()[0] == pytest.approx(loop.episode_return_sums()[0], rel=1e-6)
    fused.close(), loop.close()


def test_environment_under_an_in_kernel_mlp_policy_matches_the_oracle():
    """The recorded rollout: actions are what the NumPy restatement computes from the recorded observations, and the
    float64 oracle driven by the device's own actions and Philox draws reproduces inventory exactly and rewards to 1e-5."""
    n, seed = 2000, 5
    cfg = _cfg(n, reward="running", phi=0.01, alpha=0.02)
    env = make_env(cfg)
    rng = np.random.default_rng(7)
    layers = _random_mlp(rng, 4, 64, 2, scale=1.5)
    env.reset()
    obs_t, act_t, rew_t, steps, done = env.rollout(_native.mlp_policy(layers, "tanh"))
    assert steps == cfg.n_steps and done and obs_t.shape == (steps + 1, n, 4) and act_t.shape == (steps, n, 2)
    lo, hi = _action_space(env)
    for k in (0, 1, steps // 2, steps - 1):
        np.testing.assert_allclose(act_t[k], mlp_reference(obs_t[k], layers, "tanh", lo, hi), rtol=0, atol=2e-3)
    assert np.std(act_t) > 0.1  # a policy that actually reacts to the state
    draws = [_native.rng_fill(seed, 0, k, n) for k in range(steps)]
    oracle = OracleEnv(cfg, InjectedNoise(*[np.stack(x) for x in zip(*draws)]))
    o_obs = oracle.reset()
    np.testing.assert_allclose(obs_t[0], o_obs, rtol=0, atol=1e-6)
    for k in range(steps):
        o_obs, o_rew, _ = oracle.step(act_t[k].astype(np.float64))
        q_got = np.rint((obs_t[k + 1][:, 1].astype(np.float64) + 1) * cfg.max_inventory - cfg.max_inventory)
        q_want = np.rint((o_obs[:, 1] + 1) * cfg.max_inventory - cfg.max_inventory)
        np.testing.assert_array_equal(q_got, q_want, err_msg=f"step {k}: inventory")
        clipped = oracle.last_clipped
        err = np.abs(rew_t[k] - o_rew)
        assert np.all(err[~clipped] <= 1e-5 + 1e-6 * np.abs(o_rew[~clipped])), f"step {k}: {err[~clipped].max()}"
    env.close()


def test_learned_policy_refusals():
    from mbt_gym_amd._native import NativeError

    env = make_env(_cfg(512))
    env.reset()
    rng = np.random.default_rng(1)
    with pytest.raises(NativeError, match="floats"):
        env.policy_device(_native.mlp_policy(_random_mlp(rng, 5, 64, 2)))  # wrong observation width
    with pytest.raises(AssertionError):
        _native.mlp_policy(_random_mlp(rng, 4, 64, 2)[:2])
    wide = _random_mlp(rng, 4, 80, 2)
    with pytest.raises(NativeError, match="hidden width"):
        env.policy_device(_native.mlp_policy(wide))
    with pytest.raises(NativeError, match="learned"):
        env.policy_device(_native.MbtPolicy(kind=_native.POLICY_FIXED))
    env.close()
    touch = make_env(_cfg(512, dynamics="touch", market_half_spread=0.25, normalise_action_space=False, normalise_observation_space=False))
    touch.reset()
    with pytest.raises(NativeError, match="binary actions"):
        touch.policy_device(_native.mlp_policy(_random_mlp(rng, 4, 64, 2)))
    touch.close()


def test_raw_observations_are_refused_by_the_mlp_kernel_and_sent_to_the_host_loop_by_the_agents():
    """The matrix cores read the observation row as fp16: a raw midprice of 100 would be quantised to 0.0625, raw cash
    overflows at 65504.  The library refuses an MLP policy on an environment whose observations are not normalised (nor
    bounded by 4); the agents that would have routed to it (`has_device_policy`) keep the reference's host loop instead,
    with the network in float32.  A linear policy is float32 arithmetic and takes raw observations."""
    from mbt_gym_amd._native import NativeError
    from mbt_gym_amd.agents.SbAgent import SbAgent
    from mbt_gym_amd.gym.helpers.generate_trajectory import generate_trajectory
    from tests.test_host_logic import _fake_sb3_model

    n = 1024
    cfg = _cfg(n, n_steps=12, normalise_action_space=False, normalise_observation_space=False)
    env = make_env(cfg)
    env.reset()
    rng = np.random.default_rng(4)
    with pytest.raises(NativeError, match="fp16"):
        env.policy_device(_native.mlp_policy(_random_mlp(rng, 4, 64, 2, scale=0.05)))
    with pytest.raises(NativeError, match="fp16"):
        env.rollout(_native.mlp_policy(_random_mlp(rng, 4, 64, 2, scale=0.05)))
    env.policy_device(_native.linear_policy(rng.normal(0, 0.01, size=(2, 4)).astype(np.float32), np.array([0.5, 0.5], np.float32)))
    env.synchronize()
    agent = SbAgent(_fake_sb3_model(4, 64, 2, "Tanh", seed=3, env=env))
    assert not agent.has_device_policy
    obs_t, act_t, rew_t = generate_trajectory(env, agent)  # the reference's loop: model.predict on the host
    for k in (0, 5, 11):
        np.testing.assert_allclose(act_t[:, :, k], agent.get_action(np.ascontiguousarray(obs_t[:, :, k])), rtol=0, atol=1e-6)
    assert SbAgent(_fake_sb3_model(4, 64, 2, "Tanh", seed=3, env=make_env(_cfg(n)))).has_device_policy
    env.close()


def test_generate_trajectory_with_an_sb3_shaped_agent_runs_fused_and_agrees_with_the_host_loop():
    """The reference's caller, unchanged: generate_trajectory(env, SbAgent(model)) (GT:8-38, agents/SbAgent.py).  With an
    SB3-shaped MlpPolicy actor the episode runs in one launch with the policy in-kernel; the actions it records are what
    model.predict returns for the recorded observations (fp16-operand tolerance), and the episode statistics agree with the
    reference-style loop (model.predict on the host + env.step per time step) statistically."""
    from mbt_gym_amd.agents.SbAgent import SbAgent
    from mbt_gym_amd.gym.helpers.generate_trajectory import generate_trajectory
    from tests.test_host_logic import _fake_sb3_model

    n = 4096
    cfg = _cfg(n, n_steps=30)
    env, twin = make_env(cfg), make_env(cfg)
    agent = SbAgent(_fake_sb3_model(4, 64, 2, "Tanh", seed=3, env=env))
    assert agent.has_device_policy
    obs_f, act_f, rew_f = generate_trajectory(env, agent)                 # fused: one launch
    obs_h, act_h, rew_h = generate_trajectory(twin, SbAgent(_fake_sb3_model(4, 64, 2, "Tanh", seed=3, env=twin)), fused=False)
    assert obs_f.shape == obs_h.shape == (n, 4, 31) and act_f.shape == act_h.shape == (n, 2, 30) and rew_f.shape == rew_h.shape == (n, 1, 30)
    for k in (0, 10, 29):
        np.testing.assert_allclose(act_f[:, :, k], agent.get_action(np.ascontiguousarray(obs_f[:, :, k])), rtol=0, atol=2e-2)
    np.testing.assert_allclose(act_f[:, :, 0], act_h[:, :, 0], rtol=0, atol=2e-2)  # same first observation
    total_f, total_h = rew_f.sum(axis=(1, 2)), rew_h.sum(axis=(1, 2))
    assert total_f.mean() == pytest.approx(total_h.mean(), abs=5 * total_h.std() / np.sqrt(n) + 1e-3)
    env.close(), twin.close()


@pytest.mark.parametrize("kw,clip", [(dict(), False), (dict(dynamics="limit_and_market", market_half_spread=0.4), True)])
def test_stochastic_policy_draws_its_exploration_noise_from_its_own_philox_blocks(kw, clip):
    """action = mean + std * eps (SB3's PPO, the reference's PolicyGradientAgent, AG-PG:34-47): eps comes from Philox blocks
    with counter word 3 = 8, 9 - restated in oracle/philox_ref.py - so a stochastic rollout is reproducible from the seed,
    independent of the environment's draws, identical between the policy kernel and the fused rollout, and the recorded
    actions are what the consumer needs for its policy gradient."""
    from oracle.philox_ref import policy_exploration_noise

    n, seed = 3000, 5
    cfg = _cfg(n, n_steps=12, **kw)
    env, loop = make_env(cfg), make_env(cfg)
    a_dim = env.action_dim
    rng = np.random.default_rng(2)
    layers = _random_mlp(rng, env.observation_dim, 64, a_dim, scale=0.5)
    std = np.array([0.05, 0.1, 0.2, 0.3][:a_dim])
    policy = _native.mlp_policy(layers, "tanh", action_std=std, clip=clip)
    env.reset()
    obs_t, act_t, rew_t, steps, done = env.rollout(policy)
    assert steps == cfg.n_steps and done
    lo, hi = _action_space(env)
    for k in (0, 5, 11):
        mean = mlp_reference(obs_t[k], layers, "tanh", -np.inf, np.inf)
        want = mean + std * policy_exploration_noise(seed, 0, k, n, a_dim)
        if clip:
            want = np.clip(want, lo, hi)
        np.testing.assert_allclose(act_t[k], want, rtol=0, atol=3e-3)  # fp16 operands of the mean + hardware Box-Muller (3e-5 x std)
    if not clip:
        eps = (act_t - np.stack([mlp_reference(o, layers, "tanh", -np.inf, np.inf) for o in obs_t[:-1]])) / std
        assert abs(eps.mean()) < 0.02 and eps.std() == pytest.approx(1.0, abs=0.03)
    # the step loop with the policy kernel draws the same noise: bit-identical to the fused rollout
    loop.reset()
    for _ in range(cfg.n_steps):
        loop.policy_device(policy)
        loop.step_device()
    np.testing.assert_array_equal(loop.state, env.state)
    # std = 0 is the deterministic policy
    quiet = make_env(cfg)
    quiet.reset()
    _, act_q, _, _, _ = quiet.rollout(_native.mlp_policy(layers, "tanh", action_std=0.0, clip=True))
    np.testing.assert_allclose(act_q[0], np.clip(mlp_reference(obs_t[0], layers, "tanh", -np.inf, np.inf), lo, hi), rtol=0, atol=2e-3)
    env.close(), loop.close(), quiet.close()


@pytest.mark.timeout(300)
def test_policy_gradient_example_improves_the_return():
    """examples/policy_gradient_on_device.py: the reference's PolicyGradientAgent.train loop (PG:49-73) with the sampling in
    the kernel.  A smoke test of the whole consumer path (stochastic in-kernel MLP -> recorded tensors -> torch log-probs of
    the kernel's samples -> update): the mean episode return under the inventory-penalised reward improves."""
    import importlib.util
    import os
    import sys

    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("policy_gradient_on_device", os.path.join(root, "examples", "policy_gradient_on_device.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    argv = sys.argv
    try:
        sys.argv = ["policy_gradient_on_device.py", "13", "40"]
        history = module.main()
    finally:
        sys.argv = argv
    assert len(history) == 40 and np.mean(history[-5:]) > np.mean(history[:5])


@pytest.mark.parametrize("kind", ["exogenous_fill", "precise_state", "user_plugins", "speed", "speed_with_impact_state"])
def test_learned_policy_rollout_on_environments_without_a_fused_kernel(kind):
    """The fused learned rollout exists for the float32 tiers of the built-in models.  The exogenous-depth fill model (two
    more observation columns), precise_state and run-time compiled user plugins take the policy as a kernel of its own in
    front of every step instead - behind the same rollout call, with the same recording layout and the same exploration
    counters: the recording equals what the caller's own "policy_device, step_device" loop produces, bit for bit."""
    from tests.golden_io import load_case

    n = 1500
    if kind == "exogenous_fill":
        cfg = _cfg(n, fill="exogenous", exo_depth=(0.25, 0.375), exo_depth_lo=(0.0, 0.1), exo_depth_hi=(0.6, 0.7), base_fill_probability=0.8,
                   arrival="hawkes", intensity=(15.0, 10.0), hawkes_speed=20.0, hawkes_jump=10.0)
        kw = {}
    elif kind == "precise_state":
        cfg, kw = _cfg(n, dynamics="limit_and_market", market_half_spread=0.4, reward="running", phi=0.01, alpha=0.05, max_inventory=4), dict(precise_state=True)
    elif kind.startswith("speed"):  # optimal execution: one real-valued action (the trading speed), 1024-lane tiles, D = 4 or 5
        cfg = OracleConfig(num_trajectories=n, n_steps=40, terminal_time=1.0, midprice="bm", volatility=0.2, initial_price=100.0, arrival="none",
                           dynamics="speed", impact="temp_power" if kind == "speed" else "temp_transient", temporary_impact=0.02, transient_impact=0.3,
                           resilience=1.5, kernel_coefficient=0.2, impact_step_size=1.0 / 40, reward="cjoe", phi=0.01, alpha=0.05, initial_inventory=12,
                           max_inventory=1000, seed=5, normalise_action_space=True, normalise_observation_space=True)
        kw = {}
    else:
        cfg, _ = load_case("user_fill_and_reward")
        cfg.num_trajectories, kw = n, {}
        cfg.normalise_observation_space = cfg.normalise_action_space = True  # (the matrix cores read the observation as fp16)
    recorded, loop = make_env(cfg, **kw), make_env(cfg, **kw)
    rng = np.random.default_rng(31)
    d, a = recorded.observation_dim, recorded.action_dim
    raw = not cfg.normalise_observation_space
    policy = _native.mlp_policy(_random_mlp(rng, d, 64, a, scale=0.05 if raw else 1.5), "tanh", action_std=[0.05] * a)
    recorded.reset(), loop.reset()
    obs_t, act_t, rew_t, steps, done = recorded.rollout(policy)
    assert done and steps == cfg.n_steps and obs_t.shape == (steps + 1, n, d) and act_t.shape == (steps, n, a) and rew_t.shape == (steps, n)
    np.testing.assert_array_equal(obs_t[0], loop.observation_host())
    import torch

    for k in range(steps):
        loop.policy_device(policy)
        loop.synchronize()
        np.testing.assert_array_equal(torch.as_tensor(loop.action_device, device="cuda").cpu().numpy(), act_t[k], err_msg=f"{kind} step {k}: action")
        finished = loop.step_device()
        np.testing.assert_array_equal(loop.observation_host(), obs_t[k + 1], err_msg=f"{kind} step {k}: observation")
        loop.synchronize()
        np.testing.assert_array_equal(torch.as_tensor(loop.reward_device, device="cuda").cpu().numpy(), rew_t[k], err_msg=f"{kind} step {k}: reward")
    assert finished
    assert np.std(act_t) > 1e-3
    np.testing.assert_array_equal(recorded.state, loop.state)
    assert recorded.episode_return_sums()[0] == loop.episode_return_sums()[0]
    recorded.close(), loop.close()
