"""NumPy-only plugin subclasses in env.step(): the host-callback route (include/mbt_env.h "host-callback plugins").

The classes are the ones tests/numpy_only_plugins.py defines - plain NumPy against the reference's plugin contract, no device
expression - bound to THIS package's base classes.  The fixtures they are checked against are the REAL reference running the
very same class source bound to ITS base classes (tools/refgen/make_golden.py: user_plugin_cases).  On the same injected draws:
  arrivals, fills (post mask), inventory, dones ............ bit-exact
  rewards, host-computed (RewardFunction.calculate) ........ EQUAL to float32(reference) (the float64 states handed to the
                                                             user's code are the reference's own: precise_state is implied)
  rewards, built-in, default tier .......................... the default tier's stated bounds (tests/test_gpu_parity.py)
"""
import warnings

import numpy as np
import pytest

import mbt_gym_amd.gym.index_names as index_names
from mbt_gym_amd import _native
from mbt_gym_amd.gym.ModelDynamics import LimitAndMarketOrderModelDynamics, LimitOrderModelDynamics
from mbt_gym_amd.gym.TradingEnvironment import HostCallbackWarning, TradingEnvironment
from mbt_gym_amd.rewards.RewardFunctions import CjMmCriterion, RewardFunction, RunningInventoryPenalty
from mbt_gym_amd.stochastic_processes.arrival_models import ArrivalModel, HawkesArrivalModel, PoissonArrivalModel
from mbt_gym_amd.stochastic_processes.fill_probability_models import ExponentialFillFunction, FillProbabilityModel
from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel, OuMidpriceModel
from tests.golden_io import load_case
from tests.numpy_only_plugins import Replay, define

pytestmark = pytest.mark.gpu

USER = define(FillProbabilityModel, ArrivalModel, RewardFunction, index_names)  # the user's classes, unmodified, on OUR base classes
RAW = dict(normalise_action_space=False, normalise_observation_space=False)


def _quiet(build):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", HostCallbackWarning)
        return build()


def _fill_and_reward(g, **kw):  # tools/refgen/make_golden.py: case "user_fill_and_reward", constructor call for constructor call
    n, ns = 32, 80
    md = LimitOrderModelDynamics(
        midprice_model=BrownianMotionMidpriceModel(volatility=2.0, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
        arrival_model=PoissonArrivalModel(intensity=np.array([60.0, 45.0]), step_size=1 / ns, num_trajectories=n),
        fill_probability_model=USER.UserPowerLawFill(1.25, 1.5, step_size=1 / ns, num_trajectories=n), num_trajectories=n)
    return TradingEnvironment(terminal_time=1.0, n_steps=ns, seed=61, initial_inventory=(-2, 3), max_inventory=5, num_trajectories=n,
                              reward_function=USER.UserExponentialInventoryCost(0.05, 0.3, 0.02), model_dynamics=md, noise="injected", **RAW, **kw)


def _fill_hawkes_market(g, **kw):  # case "user_fill_hawkes_market_normalised"
    n, ns = 24, 70
    md = LimitAndMarketOrderModelDynamics(
        midprice_model=OuMidpriceModel(mean_reversion_level=100.0, mean_reversion_speed=0.02, volatility=1.5, initial_price=100.0, terminal_time=1.0,
                                       step_size=1 / ns, num_trajectories=n),
        arrival_model=HawkesArrivalModel(baseline_arrival_rate=np.array([[15.0, 10.0]]), step_size=1 / ns, jump_size=20.0, mean_reversion_speed=30.0,
                                         terminal_time=1.0, num_trajectories=n),
        fill_probability_model=USER.UserPowerLawFill(1.25, 1.5, step_size=1 / ns, num_trajectories=n), num_trajectories=n, fixed_market_half_spread=0.4)
    return TradingEnvironment(terminal_time=1.0, n_steps=ns, seed=62, initial_inventory=0, max_inventory=8, num_trajectories=n,
                              reward_function=RunningInventoryPenalty(0.01, 0.05), model_dynamics=md, noise="injected",
                              normalise_action_space=True, normalise_observation_space=True, **kw)


def _seasonal_arrivals(g, **kw):  # case "user_seasonal_arrivals"
    n, ns = 32, 100
    arrivals = USER.UserSeasonalArrivals([40.0, 30.0], 0.8, 0.5, step_size=1 / ns, num_trajectories=n)
    md = LimitOrderModelDynamics(
        midprice_model=BrownianMotionMidpriceModel(volatility=1.5, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
        arrival_model=arrivals, fill_probability_model=ExponentialFillFunction(fill_exponent=1.5, step_size=1 / ns, num_trajectories=n),
        num_trajectories=n, max_depth=None)
    env = TradingEnvironment(terminal_time=1.0, n_steps=ns, seed=64, initial_inventory=0, max_inventory=6, num_trajectories=n,
                             reward_function=RunningInventoryPenalty(0.02, 0.05), model_dynamics=md, noise="injected", **RAW, **kw)
    arrivals.rng = Replay(uniforms=g["u_arr"])  # the user's get_arrivals() draws from ITS generator (ARR:55): the fixture's draws, as the reference got them
    return env


def _cross_hawkes(g, **kw):  # case "user_cross_hawkes": an arrival model that OWNS two state columns, advanced by its own update() on the host
    n, ns = 32, 90
    arrivals = USER.UserCrossExcitingHawkes([18.0, 12.0], 25.0, 14.0, 6.0, step_size=1 / ns, terminal_time=1.0, num_trajectories=n)
    md = LimitOrderModelDynamics(
        midprice_model=OuMidpriceModel(mean_reversion_level=100.0, mean_reversion_speed=0.02, volatility=1.5, initial_price=100.0, terminal_time=1.0,
                                       step_size=1 / ns, num_trajectories=n),
        arrival_model=arrivals, fill_probability_model=ExponentialFillFunction(fill_exponent=1.5, step_size=1 / ns, num_trajectories=n), num_trajectories=n)
    env = TradingEnvironment(terminal_time=1.0, n_steps=ns, seed=66, initial_inventory=(-2, 3), max_inventory=6, num_trajectories=n,
                             reward_function=CjMmCriterion(0.02, 0.05, terminal_time=1.0), model_dynamics=md, noise="injected", **RAW, **kw)
    arrivals.rng = Replay(uniforms=g["u_arr"])
    return env


def _cev_midprice(g, **kw):  # case "user_cev_midprice": a MidpriceModel subclass whose update() is NumPy (a non-linear increment)
    n, ns = 32, 80
    mid = USER.UserCevMidprice(0.05, 0.6, 0.75, 50.0, 20.0, 80.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n)
    md = LimitOrderModelDynamics(
        midprice_model=mid, arrival_model=PoissonArrivalModel(intensity=np.array([50.0, 50.0]), step_size=1 / ns, num_trajectories=n),
        fill_probability_model=ExponentialFillFunction(fill_exponent=1.5, step_size=1 / ns, num_trajectories=n), num_trajectories=n)
    env = TradingEnvironment(terminal_time=1.0, n_steps=ns, seed=65, initial_inventory=2, max_inventory=8, num_trajectories=n,
                             reward_function=RunningInventoryPenalty(0.01, 0.02), model_dynamics=md, noise="injected", **RAW, **kw)
    mid.rng = Replay(normals=g["z"])  # the user's update() draws from ITS generator (SP:27)
    return env


def _two_factor_midprice(g, normalised=False, **kw):  # cases "user_two_factor_midprice(_normalised)": a midprice model that owns TWO columns
    n, ns = 32, 80
    mid = USER.UserShortTermAlphaMidprice(1.2, 8.0, 3.0, 0.75, 100.0, 90.0, 110.0, -10.0, 10.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n)
    md = LimitOrderModelDynamics(
        midprice_model=mid, arrival_model=PoissonArrivalModel(intensity=np.array([45.0, 60.0]), step_size=1 / ns, num_trajectories=n),
        fill_probability_model=ExponentialFillFunction(fill_exponent=1.5, step_size=1 / ns, num_trajectories=n), num_trajectories=n)
    env = TradingEnvironment(terminal_time=1.0, n_steps=ns, seed=67, initial_inventory=1, max_inventory=5, num_trajectories=n,
                             reward_function=RunningInventoryPenalty(0.01, 0.05), model_dynamics=md, noise="injected",
                             normalise_action_space=normalised, normalise_observation_space=normalised, **kw)
    # update() draws rng.normal(size=(N, 2)) in one call: column 0 is the fixture's `z`, column 1 `z_user[..., 0]` (tools/refgen/make_golden.py: user_normals)
    mid.rng = Replay(normals=np.stack([g["z"], g["z_user"][:, :, 0]], axis=-1))
    return env


def _speed(name):  # cases "user_reward_speed", "user_cev_midprice_speed", "user_impact_speed": NumPy-only plugins on the trading-with-speed kernels, built by the
    def build(g, **kw):  # shared factory exactly as the generic fixture tests build them (tests/env_factory.py picks the NumPy classes there)
        from tests.env_factory import make_env

        env = make_env(load_case(name)[0], noise="injected", **kw)
        if name == "user_cev_midprice_speed":
            env.model_dynamics.midprice_model.rng = Replay(normals=g["z"])
        return env

    return build


def _state_reading_arrivals(g, **kw):  # case "user_state_reading_arrivals": update() reads the NEW time, midprice and inventory off the matrix it is handed
    from tests.env_factory import make_env

    env = make_env(load_case("user_state_reading_arrivals")[0], noise="injected", **kw)
    env.model_dynamics.arrival_model.rng = Replay(uniforms=g["u_arr"])
    return env


CASES = {"user_state_reading_arrivals": _state_reading_arrivals, "user_reward_speed": _speed("user_reward_speed"), "user_cev_midprice_speed": _speed("user_cev_midprice_speed"),
         "user_impact_speed": _speed("user_impact_speed"), "user_adaptive_fill": _speed("user_adaptive_fill"),
         "user_cev_midprice": _cev_midprice, "user_two_factor_midprice": _two_factor_midprice,
         "user_two_factor_midprice_normalised": lambda g, **kw: _two_factor_midprice(g, normalised=True, **kw),
         "user_fill_and_reward": _fill_and_reward, "user_fill_hawkes_market_normalised": _fill_hawkes_market, "user_seasonal_arrivals": _seasonal_arrivals,
         "user_cross_hawkes": _cross_hawkes}


@pytest.mark.parametrize("precise", [False, True])
@pytest.mark.parametrize("name", sorted(CASES))
def test_numpy_only_subclasses_run_in_step_and_match_the_reference(name, precise):
    cfg, g = load_case(name)
    env = _quiet(lambda: CASES[name](g, precise_state=precise))
    exact = env.precise_state  # implied by a host-computed reward
    host_midprice = "midprice" in name
    speed = name.endswith("_speed")
    assert exact == (precise or name == "user_fill_and_reward" or host_midprice or speed)  # host-formed rewards imply the float64 tier
    env.record_events(True)
    obs = env.reset()
    n, normalised = cfg.num_trajectories, cfg.normalise_observation_space
    if exact:
        np.testing.assert_array_equal(obs, g["obs0"].astype(np.float32))
    q_of = (lambda o: np.rint((o[:, 1].astype(np.float64) + 1) * cfg.max_inventory - cfg.max_inventory)) if normalised else (lambda o: o[:, 1].astype(np.float64))
    for k in range(g["actions"].shape[0]):
        env.set_noise(None if speed else g["u_arr"][k], None if speed else g["u_fill"][k], g["z"][k])  # (a host midprice draws its normals itself, from the replayed generator: z is not read)
        obs, rew, dones, infos = env.step(g["actions"][k])
        if speed:  # no order flow (MD:273-275): inventory is real-valued state, the reference's to the bit in this tier
            np.testing.assert_array_equal(env.state64, g["obs"][k], err_msg=f"{name} step {k}: float64 state")
            np.testing.assert_array_equal(rew, g["rewards"][k].astype(np.float32), err_msg=f"{name} step {k}: rewards")
            np.testing.assert_array_equal(obs, g["obs"][k].astype(np.float32), err_msg=f"{name} step {k}: observation")
            assert bool(dones[0]) == bool(g["done"][k])
            continue
        if host_midprice and not normalised:  # the model's own columns, advanced by ITS update() on the host in float64: the reference's values
            np.testing.assert_array_equal(env.state64[:, 3:], g["obs"][k][:, 3:], err_msg=f"{name} step {k}: float64 midprice columns")
        np.testing.assert_array_equal(env.last_arrivals.astype(np.uint8), g["arrivals"][k], err_msg=f"{name} step {k}: arrivals")
        np.testing.assert_array_equal(env.last_fills.astype(np.uint8), g["fills"][k], err_msg=f"{name} step {k}: fills")
        np.testing.assert_array_equal(q_of(obs), q_of(g["obs"][k]), err_msg=f"{name} step {k}: inventory")
        if name == "user_cross_hawkes" or (exact and name == "user_state_reading_arrivals"):  # the model's own columns, advanced by ITS update() on the host in float64: the reference's values, rounded once
            np.testing.assert_array_equal(obs[:, 4:6], g["obs"][k][:, 4:6].astype(np.float32), err_msg=f"{name} step {k}: the arrival model's state columns")
            if exact:
                np.testing.assert_array_equal(env.state64[:, 4:6], g["obs"][k][:, 4:6], err_msg=f"{name} step {k}: float64 state columns")
        assert bool(dones[0]) == bool(g["done"][k]) and dones.shape == (n,) and len(infos) == n
        err = np.abs(rew.astype(np.float64) - g["rewards"][k])
        if exact:  # the reference's float64 arithmetic on the device, the user's own NumPy on the reference's own float64 states
            np.testing.assert_array_equal(rew, g["rewards"][k].astype(np.float32), err_msg=f"{name} step {k}: rewards")
            np.testing.assert_array_equal(obs, g["obs"][k].astype(np.float32), err_msg=f"{name} step {k}: observation")
        else:  # default tier, built-in reward: the tier's own bounds (clip lanes carry the float32 level of cash / midprice)
            clipped = (env.last_events >> 6) != 0
            assert np.all(err[~clipped] <= 1e-5 + 1e-6 * np.abs(g["rewards"][k][~clipped])), f"{name} step {k}: rewards off by {err[~clipped].max()}"
            assert np.all(err[clipped] <= 1.2e-4), f"{name} step {k}: rewards on clipped lanes off by {err[clipped].max()}"
        assert float(err.max()) <= (1e-5 + 1e-6 * float(np.abs(g["rewards"][k]).max()) if exact else 1.2e-4)
    env.close()


def test_stepping_right_after_the_constructor_and_after_a_batch_resize():
    """TE:74 materialises the state in the constructor and TE:173-178 re-allocates it: a host-callback environment steps from both
    without a reset() in between (the state before the step is read from the device), and keeps recording the events its plugins'
    update() is handed."""
    cfg, g = load_case("user_cross_hawkes")
    env = _quiet(lambda: CASES["user_cross_hawkes"](g))
    arrivals = env.model_dynamics.arrival_model
    before = arrivals.current_state.copy()
    for k in range(3):  # no reset(): the constructor's state
        env.set_noise(g["u_arr"][k], g["u_fill"][k], g["z"][k])
        obs, rew, _, _ = env.step(g["actions"][k])
        assert arrivals.current_state.shape == (cfg.num_trajectories, 2) and env.last_arrivals.shape == (cfg.num_trajectories, 2)
        np.testing.assert_array_equal(obs[:, 4:6], arrivals.current_state.astype(np.float32))  # its update() ran and its columns were filed
    assert not np.array_equal(before, arrivals.current_state)
    env.num_trajectories = 64  # TE:173-178
    assert arrivals.current_state.shape == (64, 2)
    arrivals.rng = np.random.default_rng(3)  # (the replayed draws of the fixture are 32 lanes wide)
    env.set_noise(np.full((64, 2), 0.9, np.float32), np.full((64, 2), 0.9, np.float32), np.zeros(64, np.float32))
    obs, rew, dones, infos = env.step(np.full((64, 2), 0.5, np.float32))
    assert obs.shape == (64, 6) and rew.shape == (64,) and len(infos) == 64 and env.last_arrivals.shape == (64, 2)
    np.testing.assert_array_equal(obs[:, 4:6], arrivals.current_state.astype(np.float32))
    env.close()


@pytest.mark.parametrize("kind", ["reward_speed", "midprice_speed", "impact_speed", "adaptive_fill", "state_reading_arrivals"])
@pytest.mark.parametrize("case", range(int(__import__("os").environ.get("MBT_FUZZ_SCALE", "1")) * 4))
def test_numpy_only_user_classes_on_random_markets_equal_the_oracle(case, kind):
    """The host-callback route over random markets (tests/random_configs.py: random_numpy_only_config), in the float64 tier, against
    the oracle - whose restatement of these user classes is pinned to the REFERENCE running the classes themselves on the same
    distribution of markets (tests/test_oracle_vs_reference_live.py).  State, observations, rewards: array-equal."""
    from oracle.mbt_oracle import InjectedNoise, OracleEnv
    from tests.env_factory import make_env
    from tests.random_configs import NUMPY_ONLY_KINDS, random_numpy_only_actions, random_numpy_only_config

    rng = np.random.default_rng(int(__import__("os").environ.get("MBT_FUZZ_SEED", "0")) + 29000 + 100 * NUMPY_ONLY_KINDS.index(kind) + case)
    n = int(rng.choice([5, 64, 700]))
    cfg = random_numpy_only_config(rng, n, kind)
    if cfg.dynamics == "speed" and cfg.impact == "temp_power" and cfg.impact_exponent != 1.0:
        cfg.impact_exponent = 1.0  # (pow of device libm against NumPy's: an ulp of float64 - the bit-for-bit claim is for exponents 1 and 2)
    steps = cfg.n_steps - int(round(cfg.start_time / cfg.step_size))
    actions = random_numpy_only_actions(rng, cfg, steps)
    u_arr = (rng.integers(0, 1 << 24, size=(steps, n, 2)) / float(1 << 24)).astype(np.float32)
    u_fill = (rng.integers(0, 1 << 24, size=(steps, n, 2)) / float(1 << 24)).astype(np.float32)
    z = rng.normal(size=(steps, n)).astype(np.float32)
    env = _quiet(lambda: make_env(cfg, noise="injected", precise_state=True))
    if kind == "midprice_speed":
        env.model_dynamics.midprice_model.rng = Replay(normals=z)
    if kind == "state_reading_arrivals":
        env.model_dynamics.arrival_model.rng = Replay(uniforms=u_arr)
    oracle = OracleEnv(cfg, InjectedNoise(u_arr, u_fill, z))
    obs, o_obs = env.reset(), oracle.reset()
    np.testing.assert_array_equal(obs, o_obs.astype(np.float32), err_msg=f"{kind} case {case}: reset")
    speed = cfg.dynamics == "speed"
    tag = f"{kind} case {case} ({cfg.midprice}/{cfg.arrival}/{cfg.dynamics}/{cfg.impact}/{cfg.reward} norm={cfg.normalise_observation_space} N={n})"
    for k in range(steps):
        env.set_noise(None if speed else u_arr[k], None if speed else u_fill[k], z[k])
        obs, rew, dones, _ = env.step(actions[k])
        o_obs, o_rew, o_done = oracle.step(actions[k].astype(np.float64))
        if cfg.reward == "exp_utility":  # exp of device libm against NumPy's (tests/test_gpu_precise.py: REWARD_VIA_LIBM)
            np.testing.assert_allclose(rew, np.broadcast_to(o_rew, (n,)), rtol=2.0 ** -23, atol=1e-12, err_msg=f"{tag} step {k}: reward")
        else:
            np.testing.assert_array_equal(rew, np.broadcast_to(o_rew, (n,)).astype(np.float32), err_msg=f"{tag} step {k}: reward")
        np.testing.assert_array_equal(obs, o_obs.astype(np.float32), err_msg=f"{tag} step {k}: observation")
        if not cfg.normalise_observation_space:
            np.testing.assert_array_equal(env.state64, o_obs, err_msg=f"{tag} step {k}: float64 state")
        assert bool(dones[0]) == bool(o_done[0])
    env.close()


def test_the_bring_your_own_numpy_plugins_example_runs(capsys, monkeypatch):
    """examples/bring_your_own_numpy_plugins.py: three NumPy-only classes (fill model, reward, price impact model) in env.step();
    selling ten units at constant speed under the square-root law costs 10 c sqrt(10) on average."""
    import importlib.util
    import os
    import sys

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "bring_your_own_numpy_plugins.py")
    spec = importlib.util.spec_from_file_location("bring_your_own_numpy_plugins", path)
    example = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(example)
    monkeypatch.setattr(sys, "argv", [path, "4096"])
    _quiet(example.main)
    lines = capsys.readouterr().out.strip().splitlines()
    assert lines[0].startswith("market making") and lines[1].startswith("optimal execution")
    cost = float(lines[1].split("mean episode return")[1].split(",")[0])
    assert cost == pytest.approx(-10 * 0.05 * np.sqrt(10), abs=0.5)  # (Monte-Carlo error of the Brownian part: ~0.1 at 4096 lanes)


def test_host_callback_plugins_say_so_and_have_no_fused_rollout():
    cfg, g = load_case("user_fill_and_reward")
    with pytest.warns(HostCallbackWarning, match="UserPowerLawFill"):
        env = CASES["user_fill_and_reward"](g)
    env.reset()
    with pytest.raises(_native.NativeError, match="step by step"):
        env.rollout(_native.MbtPolicy(kind=_native.POLICY_FIXED), record=False)
    # the C ABI refuses a step whose host inputs are missing (the Python layer always supplies them)
    done = __import__("ctypes").c_int32(0)
    act = np.zeros((cfg.num_trajectories, 2), np.float32)
    env.set_noise(g["u_arr"][0], g["u_fill"][0], g["z"][0])
    rc = _native.load_library().mbt_env_step_host(env._handle, act.ctypes.data, None, None, __import__("ctypes").byref(done))
    assert rc == -4 and b"mbt_env_set_host_fill_probabilities" in _native.load_library().mbt_last_error()
    env.close()


@pytest.mark.parametrize("name", ["user_fill_and_reward", "user_reward_speed"])
def test_host_computed_rewards_feed_the_episode_statistics(name):
    """What the host files with mbt_env_set_host_rewards is what the device-side accounting sees: reward buffer, return sums - also
    on the speed kernels, which file their own reward first and have it REPLACED (include/mbt_env.h, MBT_REW_HOST)."""
    import torch

    cfg, g = load_case(name)
    env = _quiet(lambda: CASES[name](g))
    env.reset()
    total = np.zeros(cfg.num_trajectories, np.float64)
    speed = name.endswith("_speed")
    for k in range(10):
        env.set_noise(None if speed else g["u_arr"][k], None if speed else g["u_fill"][k], g["z"][k])
        _, rew, _, _ = env.step(g["actions"][k])
        total += rew
        np.testing.assert_array_equal(torch.as_tensor(env.reward_device, device="cuda").cpu().numpy(), rew)
        np.testing.assert_array_equal(rew, g["rewards"][k].astype(np.float32))
    assert env.episode_return_sums()[0] == pytest.approx(total.sum(), rel=1e-6)
    env.close()


def test_a_subclass_that_overrides_get_fills_supplies_the_fills_themselves():
    """FILL:28-34 overridden: the subclass draws its own fills; the device takes them as they are (p = 1 or 0 against u in [0, 1))."""
    n, ns = 512, 20

    class EveryOtherLane(FillProbabilityModel):
        def __init__(self, num_trajectories):
            super().__init__(np.array([[]]), np.array([[]]), 1 / ns, 0.0, np.array([[]]), num_trajectories, None)

        def get_fills(self, depths):
            fills = np.zeros((self.num_trajectories, 2), dtype=bool)
            fills[::2, 0] = True
            fills[1::2, 1] = True
            return fills

        max_depth = 4.0

        def update(self, arrivals, fills, actions, state=None):
            pass

    md = LimitOrderModelDynamics(
        midprice_model=BrownianMotionMidpriceModel(volatility=1.0, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
        arrival_model=PoissonArrivalModel(intensity=np.array([1e9, 1e9]), step_size=1 / ns, num_trajectories=n),  # every lane sees an arrival
        fill_probability_model=EveryOtherLane(n), num_trajectories=n)
    env = _quiet(lambda: TradingEnvironment(terminal_time=1.0, n_steps=ns, model_dynamics=md, num_trajectories=n, seed=5, max_inventory=100, **RAW))
    env.reset()
    obs, _, _, _ = env.step(np.full((n, 2), 0.5, np.float32))
    np.testing.assert_array_equal(obs[::2, 1], 1.0)   # bid filled: bought one
    np.testing.assert_array_equal(obs[1::2, 1], -1.0)  # ask filled: sold one
    env.close()
