"""The device route for user-defined plugins (include/mbt_env.h: mbt_env_create_jit, mbt_jit_check).

CPU part: the expressions are compiled by hiprtc without a GPU (`check_device_expressions`), diagnostics surface as
exceptions.  GPU part: the run-time compiled kernels against the oracle fed with the kernel's own Philox draws, the fused
rollout against the step loop, and the refusals.  Parity with the REAL reference running the same user-defined classes
is in test_gpu_parity.py (fixtures user_*)."""
import numpy as np
import pytest

from oracle.mbt_oracle import InjectedNoise, OracleConfig, OracleEnv
from tests.env_factory import make_env


def _cfg(n, **kw):
    base = dict(num_trajectories=n, n_steps=50, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0, arrival="poisson",
                intensity=(80.0, 60.0), fill="user_power_law", fill_scale=1.25, fill_power=1.5, dynamics="limit",
                reward="user_exp_inventory_cost", phi=0.05, eta=0.3, alpha=0.02, initial_inventory=0, max_inventory=6, seed=17,
                normalise_action_space=False, normalise_observation_space=False)
    base.update(kw)
    return OracleConfig(**base)


def test_device_expressions_compile_without_a_gpu(no_device):
    from mbt_gym_amd import _native
    from tests.user_plugins import ExponentialInventoryCost

    env = make_env(_cfg(64))
    env.check_device_expressions()  # hiprtc cross-compiles: a plugin can be checked where it is written
    assert env.action_space.high[0] == pytest.approx(99.0 ** (1 / 1.5) / 1.25, rel=1e-6)  # the user's max_depth bounds the action (MD:118-121)
    good = ExponentialInventoryCost.device_expression
    try:
        ExponentialInventoryCost.device_expression = "pnl - undeclared_symbol * q_next"
        with pytest.raises(_native.NativeError, match="undeclared_symbol"):
            env.check_device_expressions()
    finally:
        ExponentialInventoryCost.device_expression = good
    precise = make_env(_cfg(64, arrival="hawkes", intensity=(15.0, 10.0), hawkes_speed=20.0), precise_state=True)
    precise.check_device_expressions()  # the same plugins around the float64 (precise_state) instantiation
    built_in = make_env(_cfg(64, fill="exponential", reward="pnl"))
    built_in.check_device_expressions()  # nothing to compile: a no-op
    all_three = make_env(_cfg(64, arrival="user_seasonal", intensity=(40.0, 30.0), seasonal_amplitude=0.8, seasonal_period=0.5))
    all_three.check_device_expressions()  # a user arrival model, fill model and reward in one kernel


def test_numpy_only_plugin_classes_take_the_host_callback_route(no_device):
    """A subclass of the plugin base classes that only has NumPy code - what a user of the reference writes (FILL:22-34, ARR:27-29,
    RW:10-13) - is accepted: its method keeps running on the host between launches and the kernel takes its results (MBT_FILL_HOST /
    MBT_ARR_HOST / MBT_REW_HOST).  The run-time instantiation compiles without a GPU.  What has neither a device form nor a
    host-callable method of the contract is still refused, and so is a NumPy-only arrival model with more than two columns of its own."""
    import warnings

    import mbt_gym_amd.gym.index_names as index_names
    from mbt_gym_amd import _native
    from mbt_gym_amd.gym.ModelDynamics import LimitOrderModelDynamics
    from mbt_gym_amd.gym.TradingEnvironment import HostCallbackWarning, TradingEnvironment, UnsupportedOnDevice, host_callback_role
    from mbt_gym_amd.rewards.RewardFunctions import DeviceExpressionReward, PnL, RewardFunction
    from mbt_gym_amd.stochastic_processes.arrival_models import ArrivalModel, PoissonArrivalModel
    from mbt_gym_amd.stochastic_processes.fill_probability_models import DeviceExpressionFillModel, ExponentialFillFunction, FillProbabilityModel
    from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel
    from tests.numpy_only_plugins import define

    user = define(FillProbabilityModel, ArrivalModel, RewardFunction, index_names)
    n, ns = 8, 20
    fill, arrivals, reward = user.UserPowerLawFill(1.25, 1.5, 1 / ns, n), user.UserSeasonalArrivals([40.0, 30.0], 0.8, 0.5, 1 / ns, n), user.UserExponentialInventoryCost(0.05, 0.3, 0.02)
    assert [host_callback_role(p) for p in (fill, arrivals, reward, PnL(), ExponentialFillFunction(), PoissonArrivalModel())] == ["fill", "arrival", "reward", None, None, None]

    def build(**kw):
        md = LimitOrderModelDynamics(midprice_model=BrownianMotionMidpriceModel(step_size=1 / ns, num_trajectories=n), arrival_model=kw.pop("arrival", arrivals),
                                     fill_probability_model=kw.pop("fill", fill), num_trajectories=n)
        return TradingEnvironment(n_steps=ns, model_dynamics=md, reward_function=kw.pop("reward", reward), num_trajectories=n, normalise_action_space=False,
                                  normalise_observation_space=False, **kw)

    with pytest.warns(HostCallbackWarning, match="UserPowerLawFill"):
        env = build()
    cfg = env._device_config(n, 1.0)
    assert (cfg.fill_kind, cfg.arrival_kind, cfg.reward_kind) == (_native.FILL_HOST, _native.ARR_HOST, _native.REW_HOST)
    assert cfg.precise_state == 1 and env.precise_state  # calculate() is handed float64 states: the float64 tier is implied
    assert env.action_space.high[0] == pytest.approx(99.0 ** (1 / 1.5) / 1.25, rel=1e-6)  # the user's max_depth bounds the action (MD:118-121)
    env.check_device_expressions()  # one run-time instantiation (Variant::HOST = fill | arrival | reward), compiled by hiprtc without a GPU
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", HostCallbackWarning)
        only_fill = build(arrival=PoissonArrivalModel(step_size=1 / ns, num_trajectories=n), reward=PnL())
    assert not only_fill.precise_state and only_fill._device_config(n, 1.0).reward_kind == _native.REW_PNL
    only_fill.check_device_expressions()

    class StatefulNumpyArrivals(ArrivalModel):  # NumPy-only AND two state columns of its own: accepted - ITS update() advances them on the host
        def __init__(self, columns=2):
            super().__init__(np.zeros((1, columns)), np.ones((1, columns)), 1 / ns, 1.0, np.full((1, columns), 0.5), n, None)

        def get_arrivals(self):
            return np.zeros((n, 2), dtype=bool)

        def update(self, arrivals, fills, action, state=None):
            self.current_state = self.current_state * 0.5

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", HostCallbackWarning)
        stateful = build(arrival=StatefulNumpyArrivals(), reward=PnL())
    code = stateful._user_code()
    assert code.state_columns == 2 and list(code.state_owner) == [1, 1] and list(code.state_initial) == [0.5, 0.5] and not code.state_update[0]
    assert stateful.observation_dim == 6 and stateful._device_config(n, 1.0).arrival_kind == _native.ARR_HOST
    stateful.check_device_expressions()  # the kernel carries the two columns through; mbt_env_set_host_state_columns files the host's values
    model = stateful.model_dynamics.arrival_model
    model.update(None, None, None)
    np.testing.assert_array_equal(model.current_state, np.full((n, 2), 0.25))  # its own state, on the host, although it is attached
    with pytest.raises(UnsupportedOnDevice, match="at most two columns"):
        build(arrival=StatefulNumpyArrivals(columns=3), reward=PnL())

    # a MidpriceModel subclass whose update() is NumPy (SP:33-35): MBT_MID_HOST - the kernel holds the midprice still, the model's
    # own update() moves it on the host after the launch, and the reward (any class) is formed on the host from the float64 states
    from mbt_gym_amd.rewards.RewardFunctions import RunningInventoryPenalty

    def with_midprice(mid, **kw):
        md = LimitOrderModelDynamics(midprice_model=mid, arrival_model=kw.pop("arrival", PoissonArrivalModel(step_size=1 / ns, num_trajectories=n)),
                                     fill_probability_model=ExponentialFillFunction(step_size=1 / ns, num_trajectories=n), num_trajectories=n)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", HostCallbackWarning)
            return TradingEnvironment(n_steps=ns, model_dynamics=md, reward_function=kw.pop("reward", RunningInventoryPenalty(0.01, 0.02)), num_trajectories=n,
                                      normalise_action_space=False, normalise_observation_space=False, **kw)

    cev = user.UserCevMidprice(0.05, 0.6, 0.75, 50.0, 20.0, 80.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n)
    assert host_callback_role(cev) == "midprice" and host_callback_role(BrownianMotionMidpriceModel()) is None
    env = with_midprice(cev)
    cfg = env._device_config(n, 1.0)
    assert (cfg.midprice_kind, cfg.reward_kind, cfg.initial_price, cfg.precise_state) == (_native.MID_HOST, _native.REW_HOST, 50.0, 1)
    assert cfg.midprice_step_size == 1 / ns  # the model's OWN step size (SP:21): what speed dynamics trade per step (MD:265)
    assert env._user_code() is None and env._host_owned_columns()[:2] == (3, 4)
    env.check_device_expressions()
    two = with_midprice(user.UserShortTermAlphaMidprice(1.2, 8.0, 3.0, 0.75, 100.0, 90.0, 110.0, -10.0, 10.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                        arrival=StatefulNumpyArrivals(columns=1))
    code = two._user_code()
    assert code.state_columns == 2 and list(code.state_owner) == [0, 1] and list(code.state_initial) == [0.0, 0.5] and not code.state_update[0]
    assert two.observation_dim == 6 and two._host_owned_columns()[:2] == (3, 6)  # midprice, alpha, the arrival model's column: one block
    two.check_device_expressions()
    with pytest.raises(UnsupportedOnDevice, match="at most two"):
        with_midprice(user.UserShortTermAlphaMidprice(1.2, 8.0, 3.0, 0.75, 100.0, 90.0, 110.0, -10.0, 10.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                      arrival=StatefulNumpyArrivals(columns=2))
    from tests.user_plugins import ExponentialInventoryCost  # a reward that exists as a device expression only: no calculate() to form it on the host

    with pytest.raises(UnsupportedOnDevice, match="device expression only"):
        with_midprice(user.UserCevMidprice(0.05, 0.6, 0.75, 50.0, 20.0, 80.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n), reward=ExponentialInventoryCost(0.1, 0.5, 0.05))
    lib_cfg = env._device_config(n, 1.0)
    lib_cfg.reward_kind = _native.REW_PNL  # the C ABI refuses a host midprice whose reward the kernel would have to form
    assert _native.load_library().mbt_jit_check(__import__("ctypes").byref(lib_cfg), __import__("ctypes").byref(_native.MbtUserCode())) == -1
    assert b"MBT_REW_HOST" in _native.load_library().mbt_last_error()

    # a fill model that OWNS a state column (state_owner = 2: behind the arrival model's), and what the one-block rule refuses
    adaptive = user.UserAdaptiveFill(1.5, 4.0, 0.5, 0.5, 8.0, step_size=1 / ns, num_trajectories=n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", HostCallbackWarning)
        stateful_fill = build(fill=adaptive, arrival=PoissonArrivalModel(step_size=1 / ns, num_trajectories=n), reward=PnL())
    code = stateful_fill._user_code()
    assert code.state_columns == 1 and list(code.state_owner)[:1] == [2] and list(code.state_initial)[:1] == [1.5] and not code.state_update[0]
    assert stateful_fill.observation_dim == 5 and stateful_fill._host_owned_columns()[:2] == (4, 5) and not stateful_fill.precise_state
    stateful_fill.check_device_expressions()
    from tests.user_plugins import CrossExcitingHawkes  # a DEVICE-resident arrival model with two columns, between a host midprice and a host fill model

    with pytest.raises(UnsupportedOnDevice, match="not adjacent"):
        md = LimitOrderModelDynamics(midprice_model=user.UserCevMidprice(0.05, 0.6, 0.75, 50.0, 20.0, 80.0, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                                     arrival_model=CrossExcitingHawkes(baseline=(18.0, 12.0), speed=25.0, jump=14.0, cross=6.0, step_size=1 / ns, terminal_time=1.0, num_trajectories=n),
                                     fill_probability_model=user.UserAdaptiveFill(1.5, 4.0, 0.5, 0.5, 8.0, step_size=1 / ns, num_trajectories=n), num_trajectories=n)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", HostCallbackWarning)
            TradingEnvironment(n_steps=ns, model_dynamics=md, num_trajectories=n, normalise_action_space=False, normalise_observation_space=False)

    # a PriceImpactModel subclass whose get_impact() is NumPy (IMP:25-27), trading-with-speed dynamics: MBT_IMPACT_HOST(_STATE) - the
    # caller's impacts go to the kernel before every step; its state column lives on the host
    from mbt_gym_amd.gym.ModelDynamics import TradinghWithSpeedModelDynamics
    from mbt_gym_amd.stochastic_processes.price_impact_models import PriceImpactModel, TemporaryPowerPriceImpact

    speed_user = define(FillProbabilityModel, ArrivalModel, RewardFunction, index_names, PriceImpactModel=PriceImpactModel)
    sqrt_impact = speed_user.UserSquareRootImpact(0.05, 2.0, 0.3, 10.0, step_size=1 / ns, terminal_time=1.0, num_trajectories=n)
    assert host_callback_role(sqrt_impact) == "impact" and host_callback_role(TemporaryPowerPriceImpact()) is None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", HostCallbackWarning)
        execution = TradingEnvironment(n_steps=ns, num_trajectories=n, normalise_action_space=False, normalise_observation_space=False, model_dynamics=TradinghWithSpeedModelDynamics(
            midprice_model=BrownianMotionMidpriceModel(step_size=1 / ns, num_trajectories=n), price_impact_model=sqrt_impact, num_trajectories=n))
    cfg = execution._device_config(n, 1.0)
    assert (cfg.impact_kind, cfg.initial_transient_impact, cfg.precise_state, cfg.reward_kind) == (_native.IMPACT_HOST_STATE, 0.0, 1, _native.REW_PNL)
    assert execution.observation_dim == 5 and execution._host_owned_columns()[:2] == (4, 5) and execution.action_space.high[0] == 10.0
    execution.check_device_expressions()  # nothing to compile: the speed kernels are built ahead of time
    lib_cfg = execution._device_config(n, 1.0)
    lib_cfg.precise_state = 0  # float64 impacts belong to the float64 tier: refused otherwise, before any device is touched
    handle = __import__("ctypes").c_void_p()
    assert _native.load_library().mbt_env_create(__import__("ctypes").byref(lib_cfg), __import__("ctypes").byref(handle)) == -1
    assert b"precise_state" in _native.load_library().mbt_last_error()

    class NothingToRun(FillProbabilityModel):  # neither a device form nor _get_fill_probabilities / get_fills
        def __init__(self):
            super().__init__(np.array([[]]), np.array([[]]), 1 / ns, 0.0, np.array([[]]), n, None)

        max_depth = 3.0

    nothing = build(fill=NothingToRun(), arrival=PoissonArrivalModel(step_size=1 / ns, num_trajectories=n), reward=PnL())
    with pytest.raises(UnsupportedOnDevice, match="no HIP implementation"):
        nothing._device_config(n, 1.0)  # (what the constructor does when it creates the device handle)
    with pytest.raises(TypeError, match="device_expression"):
        type("NoExpression", (DeviceExpressionReward,), {})()
    with pytest.raises(TypeError, match="device_expression"):
        type("NoExpression", (DeviceExpressionFillModel,), {})()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(dynamics="limit_and_market", market_half_spread=0.4, arrival="hawkes", intensity=(15.0, 10.0), hawkes_speed=20.0),
                                dict(fill="exponential"), dict(reward="running"),
                                dict(arrival="user_seasonal", intensity=(40.0, 30.0), seasonal_amplitude=0.8, seasonal_period=0.5),
                                dict(arrival="user_seasonal", intensity=(40.0, 30.0), seasonal_amplitude=0.8, seasonal_period=0.5, fill="exponential", reward="pnl",
                                     dynamics="touch", market_half_spread=0.25),
                                dict(midprice="user_cev", drift=0.05, volatility=0.6, cev_gamma=0.75, initial_price=50.0, midprice_lo=20.0, midprice_hi=80.0),
                                dict(midprice="user_cev", drift=0.05, volatility=0.6, cev_gamma=0.75, initial_price=50.0, midprice_lo=20.0, midprice_hi=80.0,
                                     fill="exponential", reward="pnl"),
                                # processes that OWN state columns: a two-column arrival model, a two-column midprice (with two more normals per step), both
                                dict(arrival="user_cross_hawkes", intensity=(18.0, 12.0), hawkes_speed=25.0, hawkes_jump=14.0, hawkes_cross=6.0),
                                dict(arrival="user_cross_hawkes", intensity=(18.0, 12.0), hawkes_speed=25.0, hawkes_jump=14.0, hawkes_cross=6.0, fill="exponential",
                                     reward="cjmm", dynamics="limit_and_market", market_half_spread=0.4, normalise_action_space=True, normalise_observation_space=True),
                                dict(midprice="user_alpha", volatility=1.2, alpha_kappa=8.0, alpha_xi=3.0, alpha_eps=0.75, midprice_lo=90.0, midprice_hi=110.0,
                                     alpha_lo=-10.0, alpha_hi=10.0, fill="exponential", reward="running"),
                                dict(midprice="user_alpha", volatility=1.2, alpha_kappa=8.0, alpha_xi=3.0, alpha_eps=0.75, midprice_lo=90.0, midprice_hi=110.0,
                                     alpha_lo=-10.0, alpha_hi=10.0, dynamics="touch", market_half_spread=0.25, fill="exponential", reward="pnl")])
def test_user_plugins_on_philox_noise_match_the_oracle_and_the_fused_rollout(kw):
    from mbt_gym_amd import _native

    n, seed = 3000, 17
    cfg = _cfg(n, **kw)
    steps = cfg.n_steps
    a_dim = 4 if cfg.dynamics == "limit_and_market" else 2
    action = np.tile(np.array([[0.3, 0.5, 0.0, 1.0][:a_dim]], np.float32), (n, 1))
    if cfg.dynamics == "touch":
        action = np.tile(np.array([[1.0, 1.0]], np.float32), (n, 1))  # post on both sides
    env, fused = make_env(cfg), make_env(cfg)
    draws = [_native.rng_fill(seed, 0, k, n) for k in range(steps)]
    z_user = np.stack([_native.rng_fill_user(seed, 0, k, n) for k in range(steps)]) if cfg.midprice == "user_alpha" else None
    oracle = OracleEnv(cfg, InjectedNoise(*[np.stack(x) for x in zip(*draws)], z_user))
    if cfg.normalise_action_space:
        action = np.tile(np.array([[-0.6, -0.4, -1.0, 0.3][:a_dim]], np.float32), (n, 1))
    env.reset(), fused.reset(), oracle.reset()
    assert env.observation_dim == cfg.state_dim
    for k in range(steps):
        obs, rew, dones, _ = env.step(action)
        o_obs, o_rew, _ = oracle.step(action.astype(np.float64))
        if cfg.normalise_observation_space:
            np.testing.assert_allclose(obs, o_obs, rtol=0, atol=1e-4, err_msg=f"step {k}: normalised observation")
        else:
            np.testing.assert_array_equal(obs[:, 1], o_obs[:, 1], err_msg=f"step {k}: inventory")
            np.testing.assert_allclose(obs[:, 4:], o_obs[:, 4:], rtol=3e-6, atol=3e-5, err_msg=f"step {k}: process state columns")
        clipped = oracle.last_clipped
        err = np.abs(rew - o_rew)
        assert np.all(err[~clipped] <= 1e-5 + 1e-6 * np.abs(o_rew[~clipped])), f"step {k}: reward {err[~clipped].max()}"
        assert np.all(err[clipped] <= 2e-4)
    assert dones[0]
    fused.set_action_host(action)
    assert fused.step_repeat_device(steps) == (steps, True)
    np.testing.assert_array_equal(fused.state, env.state)
    assert fused.episode_return_sums()[0] == pytest.approx(env.episode_return_sums()[0], rel=1e-6)
    env.close(), fused.close()


@pytest.mark.gpu
def test_user_plugin_refusals_and_cache():
    import time

    from mbt_gym_amd._native import NativeError

    t0 = time.perf_counter()
    first = make_env(_cfg(1024))
    t1 = time.perf_counter()
    second = make_env(_cfg(2048))  # same plugins, same kernels: served from the process-wide module cache
    t2 = time.perf_counter()
    assert (t2 - t1) < 0.5 * (t1 - t0) + 0.2
    first.close(), second.close()
    # device expressions are compiled around the ORDER-BOOK kernels: with trading-with-speed dynamics the library refuses them (a
    # NumPy-only class takes the host-callback route there: tests/test_gpu_host_callbacks.py, fixture user_reward_speed)
    from mbt_gym_amd.gym.ModelDynamics import TradinghWithSpeedModelDynamics
    from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
    from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel
    from mbt_gym_amd.stochastic_processes.price_impact_models import TemporaryAndPermanentPriceImpact
    from tests.user_plugins import ExponentialInventoryCost

    with pytest.raises(NativeError, match="speed|order-book"):
        TradingEnvironment(n_steps=20, num_trajectories=64, reward_function=ExponentialInventoryCost(0.1, 0.5, 0.05), model_dynamics=TradinghWithSpeedModelDynamics(
            midprice_model=BrownianMotionMidpriceModel(step_size=1 / 20, num_trajectories=64),
            price_impact_model=TemporaryAndPermanentPriceImpact(n_steps=20, num_trajectories=64), num_trajectories=64), normalise_action_space=False,
            normalise_observation_space=False)


def _state_reading_env(g, **kw):
    from mbt_gym_amd.gym.ModelDynamics import LimitOrderModelDynamics
    from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment
    from mbt_gym_amd.rewards.RewardFunctions import RunningInventoryPenalty
    from mbt_gym_amd.stochastic_processes.fill_probability_models import ExponentialFillFunction
    from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel
    from tests.user_plugins import StateReadingArrivals

    n, ns = 32, 90  # tools/refgen/make_golden.py: case "user_state_reading_arrivals", constructor call for constructor call
    md = LimitOrderModelDynamics(
        midprice_model=BrownianMotionMidpriceModel(drift=0.5, volatility=2.5, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
        arrival_model=StateReadingArrivals([40.0, 30.0], 20.0, 0.6, 3.0, 0.8, 100.0, step_size=1 / ns, terminal_time=1.0, num_trajectories=n),
        fill_probability_model=ExponentialFillFunction(fill_exponent=1.5, step_size=1 / ns, num_trajectories=n), num_trajectories=n)
    kw = dict(dict(noise="injected", initial_inventory=(-2, 3)), **kw)
    return TradingEnvironment(terminal_time=1.0, n_steps=ns, seed=75, max_inventory=6, num_trajectories=n, reward_function=RunningInventoryPenalty(0.01, 0.05),
                              model_dynamics=md, normalise_action_space=False, normalise_observation_space=False, **kw)


def test_state_update_expressions_name_the_matrix_the_reference_hands_update(no_device):
    """`S_next`, `t_next`, `q_next`, `cash_next` in a state-update expression (include/mbt_env.h): compiles without a GPU."""
    from tests.golden_io import load_case

    env = _state_reading_env(load_case("user_state_reading_arrivals")[1])
    code = env._user_code()
    assert code.state_columns == 2 and b"t_next" in code.state_update[0] and b"S_next" in code.state_update[1]
    env.check_device_expressions()


@pytest.mark.gpu
@pytest.mark.parametrize("precise", [False, True])
def test_a_device_expression_that_reads_the_advanced_state_matches_the_reference(precise):
    """The reference hands update() the state matrix with cash / inventory / time and the midprice already advanced (TE:206-211); the
    NumPy class of the fixture reads them off it, the device expression names them `S_next`, `t_next`, `q_next`."""
    from tests.golden_io import load_case

    cfg, g = load_case("user_state_reading_arrivals")
    env = _state_reading_env(g, precise_state=precise)
    env.record_events(True)
    env.reset()
    for k in range(g["actions"].shape[0]):
        env.set_noise(g["u_arr"][k], g["u_fill"][k], g["z"][k])
        obs, rew, dones, _ = env.step(g["actions"][k])
        np.testing.assert_array_equal(env.last_arrivals.astype(np.uint8), g["arrivals"][k], err_msg=f"step {k}: arrivals")
        np.testing.assert_array_equal(env.last_fills.astype(np.uint8), g["fills"][k], err_msg=f"step {k}: fills")
        np.testing.assert_array_equal(obs[:, 1], g["obs"][k][:, 1], err_msg=f"step {k}: inventory")
        if precise:
            np.testing.assert_allclose(env.state64[:, 4:6], g["obs"][k][:, 4:6], rtol=1e-12, atol=1e-12, err_msg=f"step {k}: intensities")
            np.testing.assert_array_equal(rew, g["rewards"][k].astype(np.float32), err_msg=f"step {k}: rewards")
        else:
            np.testing.assert_allclose(obs[:, 4:6], g["obs"][k][:, 4:6], rtol=2e-6, atol=2e-4, err_msg=f"step {k}: intensities (float32 state)")
    env.close()


@pytest.mark.gpu
def test_the_fused_rollout_hands_state_update_expressions_the_same_advanced_state_as_the_step_kernel():
    """`t_next` / `S_next` / `q_next` inside the fused rollout kernel (its own clock, its own registers) and in the step kernel: a
    rollout under a fixed action is the step loop, bit for bit (production noise)."""
    from mbt_gym_amd.agents.BaselineAgents import FixedActionAgent
    from tests.golden_io import load_case

    g = load_case("user_state_reading_arrivals")[1]
    env_a, env_b = _state_reading_env(g, noise="philox", initial_inventory=1), _state_reading_env(g, noise="philox", initial_inventory=1)
    agent = FixedActionAgent(np.array([0.5, 0.7], np.float32), env_a)
    env_a.reset(), env_b.reset()
    obs_r, act_r, rew_r, steps, done = env_a.rollout(agent)
    assert steps == env_a.n_steps and done
    action = np.tile(np.array([[0.5, 0.7]], np.float32), (env_b.num_trajectories, 1))
    for k in range(steps):
        obs, rew, dones, _ = env_b.step(action)
        np.testing.assert_array_equal(obs_r[k + 1], obs, err_msg=f"step {k}: observation (incl. the two intensities)")  # (time-major: tests/test_gpu_rollout.py)
        np.testing.assert_array_equal(rew_r[k], rew, err_msg=f"step {k}: reward")
    np.testing.assert_array_equal(env_a.state, env_b.state)
    env_a.close(), env_b.close()


@pytest.mark.gpu
def test_normalise_rewards_calibrates_an_environment_with_a_user_defined_reward():
    """TE:329-343 rolls a copy of the environment out under the fixed action 1 / kappa; with a user-defined reward (or
    midprice) that copy is a run-time compiled environment too.  The scale it finds is 1 / (mean episode return) of exactly
    that rollout, and a fill model without a `fill_exponent` is refused with a message instead of an AttributeError."""
    from mbt_gym_amd.gym.TradingEnvironment import UnsupportedOnDevice

    cfg = _cfg(4096, fill="exponential", fill_exponent=1.5)
    plain = make_env(cfg)
    scaled = make_env(cfg, normalise_rewards=True)
    assert np.isfinite(scaled.reward_scaling) and scaled.reward_scaling > 0
    action = np.tile(np.array([[1 / 1.5, 1 / 1.5]], np.float32), (4096, 1))
    plain.reset(), scaled.reset()
    total_plain = total_scaled = 0.0
    for _ in range(cfg.n_steps):
        total_plain += plain.step(action)[1].astype(np.float64).mean()
        total_scaled += scaled.step(action)[1].astype(np.float64).mean()
    assert total_scaled == pytest.approx(total_plain * scaled.reward_scaling, rel=1e-4)
    assert total_scaled == pytest.approx(1.0, abs=0.1)  # the calibration's own rollout (100 000 lanes, another key) has mean return 1 / scale
    plain.close(), scaled.close()
    with pytest.raises((UnsupportedOnDevice, AssertionError)):
        make_env(_cfg(64), normalise_rewards=True)  # the power-law fill model has no fill_exponent (and is not the exponential model TE:90-93 asserts)
