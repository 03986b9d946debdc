"""The oracle against the reference ITSELF, over random draws from the plugin space - where the reference is importable.

tests/golden/*.npz pin oracle/mbt_oracle.py to the reference on 26 hand-picked configurations (the fixtures travel to the GPU
box, the reference cannot).  In the build container /root/reference is present, so here the pin is spread over the whole
space the GPU fuzz tests draw from (tests/random_configs.py: every built-in midprice / arrival / dynamics / reward /
price-impact kind, normalised or not, late start times, foreign step sizes): the REAL reference environment - built by the
same construction code as ours (tests/env_factory.py, package="mbt_gym") - and the oracle are fed identical
float32-representable draws and actions, and every observation, reward and done flag must agree.  Together with the GPU
fuzz (HIP vs oracle on the same space) this closes the chain reference -> oracle -> HIP for every combination drawn.

Skipped where /root/reference does not exist (the GPU box; nothing here is marked gpu).  Never reads the reference's sources:
it imports the package, read-only, with bytecode writing off, and gym satisfied by the numerics-free stand-in of
tools/refgen/gym_standin (gym is not installed here and there is no network)."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

from oracle.mbt_oracle import InjectedNoise, OracleEnv
from tests.env_factory import make_env
from tests.random_configs import (NUMPY_ONLY_KINDS, random_actions, random_config, random_numpy_only_actions, random_numpy_only_config, random_speed_actions,
                                  random_speed_config)

REFERENCE = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "mbt_gym")), reason="the reference is only present in the build container")

CASES = int(os.environ.get("MBT_LIVE_CASES", "40"))
SEED = int(os.environ.get("MBT_FUZZ_SEED", "0"))


@pytest.fixture(scope="module", autouse=True)
def reference_on_path():
    sys.dont_write_bytecode = True
    added = [os.path.join(ROOT, "tools", "refgen", "gym_standin"), REFERENCE]
    for p in added:
        sys.path.insert(0, p)
    yield
    for p in added:
        sys.path.remove(p)


class Replay:
    """Stands in for the numpy Generator of one reference process (they only call uniform(size=) / normal(size=):
    ARR:55, ARR:122, FILL:33, MID:64, MID:143)."""

    def __init__(self, draws):
        self.draws, self.k = draws, 0

    def _next(self, size):
        out = np.asarray(self.draws[self.k], dtype=np.float64).reshape(size)
        self.k += 1
        return out

    def uniform(self, size=None):
        return self._next(size)

    def normal(self, size=None):
        return self._next(size)


def _noise(rng, steps, n):
    u_arr = (rng.integers(0, 1 << 24, size=(steps, n, 2)) / float(1 << 24)).astype(np.float32)
    u_fill = (rng.integers(0, 1 << 24, size=(steps, n, 2)) / float(1 << 24)).astype(np.float32)
    z = rng.normal(size=(steps, n)).astype(np.float32)
    return u_arr, u_fill, z


def _compare(cfg, actions, u_arr, u_fill, z, tag):
    with contextlib.redirect_stdout(io.StringIO()):  # TE:291-297 prints whole arrays whenever a clip fires
        ref = make_env(cfg, package="mbt_gym")
    md = ref.model_dynamics
    md.midprice_model.rng = Replay(z)
    if md.arrival_model is not None:
        md.arrival_model.rng = Replay(u_arr)
    if md.fill_probability_model is not None:
        md.fill_probability_model.rng = Replay(u_fill)
    oracle = OracleEnv(cfg, InjectedNoise(u_arr, u_fill, z))
    n = cfg.num_trajectories
    with contextlib.redirect_stdout(io.StringIO()):
        r_obs = ref.reset()
    o_obs = oracle.reset()
    np.testing.assert_array_equal(np.asarray(r_obs, dtype=np.float64), o_obs, err_msg=f"{tag}: reset")
    for k in range(actions.shape[0]):
        with contextlib.redirect_stdout(io.StringIO()):
            r_obs, r_rew, r_done, _ = ref.step(actions[k].astype(np.float64))
        o_obs, o_rew, o_done = oracle.step(actions[k].astype(np.float64))
        r_rew = np.broadcast_to(np.asarray(r_rew, dtype=np.float64), (n,))  # ExponentialUtility returns the scalar 0 (RW:156-163)
        o_rew = np.broadcast_to(np.asarray(o_rew, dtype=np.float64), (n,))
        np.testing.assert_array_equal(np.asarray(r_obs, dtype=np.float64), o_obs, err_msg=f"{tag} step {k}: observation")
        np.testing.assert_array_equal(r_rew, o_rew, err_msg=f"{tag} step {k}: reward")
        assert bool(r_done[0]) == bool(o_done[0]), f"{tag} step {k}: done"
    assert bool(r_done[0])


@pytest.mark.parametrize("case", range(CASES))
def test_oracle_equals_the_reference_on_a_random_order_book_configuration(case):
    rng = np.random.default_rng(SEED + 21000 + case)
    n = int(rng.choice([1, 5, 64]))
    cfg = random_config(rng, n)
    steps = cfg.n_steps - int(round(cfg.start_time / cfg.step_size))
    actions = random_actions(rng, cfg, steps)
    u_arr, u_fill, z = _noise(rng, steps, n)
    tag = f"live case {case}: {cfg.midprice}/{cfg.arrival}/{cfg.dynamics}/{cfg.reward} norm={cfg.normalise_observation_space} N={n}"
    _compare(cfg, actions, u_arr, u_fill, z, tag)


@pytest.mark.parametrize("case", range(CASES // 2))
def test_oracle_equals_the_reference_on_a_random_optimal_execution_configuration(case):
    rng = np.random.default_rng(SEED + 23000 + case)
    n = int(rng.choice([1, 5, 64]))
    cfg = random_speed_config(rng, n)
    actions = random_speed_actions(rng, cfg, cfg.n_steps)
    u_arr, u_fill, z = _noise(rng, cfg.n_steps, n)
    tag = f"live speed case {case}: {cfg.midprice}/{cfg.impact}/{cfg.reward} norm={cfg.normalise_observation_space} N={n}"
    _compare(cfg, actions, u_arr, u_fill, z, tag)


@pytest.mark.parametrize("kind", NUMPY_ONLY_KINDS)
@pytest.mark.parametrize("case", range(max(3, CASES // 8)))
def test_oracle_restates_the_numpy_only_user_classes_on_random_markets(case, kind):
    """tests/numpy_only_plugins.py bound to the REFERENCE's base classes and run by the reference, against the oracle's restatement of
    those classes (oracle/mbt_oracle.py: "user_exp_inventory_cost", "user_cev", "user_sqrt", "user_adaptive") - the pin the GPU tests of
    the host-callback route lean on, spread from one fixture per class over random markets."""
    rng = np.random.default_rng(SEED + 27000 + 100 * NUMPY_ONLY_KINDS.index(kind) + case)
    n = int(rng.choice([1, 5, 64]))
    cfg = random_numpy_only_config(rng, n, kind)
    steps = cfg.n_steps - int(round(cfg.start_time / cfg.step_size))
    actions = random_numpy_only_actions(rng, cfg, steps)
    u_arr, u_fill, z = _noise(rng, steps, n)
    _compare(cfg, actions, u_arr, u_fill, z, f"live {kind} case {case}: {cfg.midprice}/{cfg.arrival}/{cfg.dynamics}/{cfg.impact}/{cfg.reward} norm={cfg.normalise_observation_space} N={n}")


# ---- the closed-form agents (callers of the path; host code on both sides) -------------------------------------------

def _agents(package):
    import importlib

    return importlib.import_module(package + ".agents.BaselineAgents")


@pytest.mark.parametrize("case", range(max(4, CASES // 4)))
def test_closed_form_agents_equal_the_references_on_random_parameters(case, no_device):
    """AvellanedaStoikov (AG:52-83), CarteaJaimungalMm (AG:86-170; asymmetric intensities, the matrix exponential) and
    CarteaJaimungalOe (AG:173-210) of this package against the reference's own classes, each constructed on its own
    package's environment of the same random market, on a batch of random states with a common time stamp."""
    from oracle.mbt_oracle import OracleConfig

    rng = np.random.default_rng(SEED + 25000 + case)
    n, ns, T = 16, int(rng.integers(20, 200)), float(rng.choice([0.5, 1.0, 2.0]))
    q_max = int(rng.integers(3, 25))
    market = dict(num_trajectories=n, n_steps=ns, terminal_time=T, midprice="bm", volatility=float(rng.uniform(0.5, 3.0)), initial_price=100.0,
                  arrival="poisson", intensity=(float(rng.uniform(20, 160)), float(rng.uniform(20, 160))), fill_exponent=float(rng.uniform(0.5, 3.0)),
                  initial_inventory=0, max_inventory=q_max, seed=11, normalise_action_space=False, normalise_observation_space=False)
    state = np.zeros((n, 4))
    state[:, 1] = rng.integers(-q_max - 2, q_max + 3, size=n)  # including inventories beyond the agent's table (AG:133-135 clamps)
    state[:, 2] = T * int(rng.integers(0, ns)) / ns
    state[:, 3] = 100.0 + rng.normal(size=n)

    def both(cfg):
        with contextlib.redirect_stdout(io.StringIO()):
            return make_env(cfg), make_env(cfg, package="mbt_gym")

    # Avellaneda-Stoikov on plain PnL
    ours, ref = both(OracleConfig(dynamics="limit", reward="pnl", **market))
    gamma = float(rng.choice([0.0, 0.01, 0.1, 1.0]))
    a, b = _agents("mbt_gym_amd").AvellanedaStoikovAgent(gamma, ours), _agents("mbt_gym").AvellanedaStoikovAgent(gamma, ref)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "quoting a negative spread" for large inventories, on both sides
        np.testing.assert_allclose(a.get_action(state), b.get_action(state), rtol=1e-6, atol=1e-6, err_msg=f"AS case {case}")  # float32 actions
    # Cartea-Jaimungal market making on its criterion
    ours, ref = both(OracleConfig(dynamics="limit", reward="cjmm", phi=float(rng.uniform(0.0, 0.05)), alpha=float(rng.uniform(0.0, 0.01)), **market))
    a, b = _agents("mbt_gym_amd").CarteaJaimungalMmAgent(ours), _agents("mbt_gym").CarteaJaimungalMmAgent(ref)
    np.testing.assert_allclose(a.get_action(state), b.get_action(state), rtol=2e-6, atol=2e-6, err_msg=f"CJ-MM case {case}")
    # (the reference's calculate_true_value_function adds (n,1) to (n,): an (n,n) broadcast - compared through h instead)
    np.testing.assert_allclose(a.h_table()[int(round(state[0, 2] / T * ns))], b._calculate_ht(state[0, 2]).reshape(-1), rtol=1e-8, atol=1e-10, err_msg=f"CJ-MM h case {case}")
    # Cartea-Jaimungal optimal execution on temporary + permanent impact
    cfg = OracleConfig(num_trajectories=n, n_steps=ns, terminal_time=T, midprice="bm", volatility=market["volatility"], initial_price=100.0, arrival="none",
                       dynamics="speed", impact="temp_perm", temporary_impact=float(rng.uniform(0.005, 0.05)), permanent_impact=float(rng.uniform(0.0, 0.01)),
                       impact_step_size=T / ns, reward="cjoe", phi=0.01, alpha=0.05, initial_inventory=int(rng.integers(1, 30)), max_inventory=1000, seed=11,
                       normalise_action_space=False, normalise_observation_space=False)
    ours, ref = both(cfg)
    phi, alpha = float(rng.uniform(1e-4, 1e-2)), float(rng.uniform(1e-3, 1e-1))
    a, b = _agents("mbt_gym_amd").CarteaJaimungalOeAgent(phi, alpha, ours), _agents("mbt_gym").CarteaJaimungalOeAgent(phi, alpha, ref)
    np.testing.assert_allclose(a.get_action(state), b.get_action(state), rtol=2e-6, atol=1e-6, err_msg=f"CJ-OE case {case}")


# ---- the API surface: every mirrored class takes the reference's constructor arguments and offers its public methods ----

MIRRORED = {
    "gym.TradingEnvironment": ["TradingEnvironment"],
    "gym.ModelDynamics": ["ModelDynamics", "LimitOrderModelDynamics", "LimitAndMarketOrderModelDynamics", "AtTheTouchModelDynamics", "TradinghWithSpeedModelDynamics"],
    "stochastic_processes.StochasticProcessModel": ["StochasticProcessModel"],
    "stochastic_processes.midprice_models": ["ConstantMidpriceModel", "BrownianMotionMidpriceModel", "GeometricBrownianMotionMidpriceModel", "OuMidpriceModel",
                                             "BrownianMotionJumpMidpriceModel", "OuJumpMidpriceModel"],
    "stochastic_processes.arrival_models": ["ArrivalModel", "PoissonArrivalModel", "PoissonArrivalNonLinearModel", "HawkesArrivalModel"],
    "stochastic_processes.fill_probability_models": ["FillProbabilityModel", "ExponentialFillFunction", "ExogenousMmFillProbabilityModel"],
    "stochastic_processes.price_impact_models": ["PriceImpactModel", "TemporaryPowerPriceImpact", "TemporaryAndPermanentPriceImpact",
                                                 "TemporaryAndTransientPriceImpact", "TransientPriceImpact"],
    "rewards.RewardFunctions": ["RewardFunction", "PnL", "CjOeCriterion", "CjMmCriterion", "RunningInventoryPenalty", "ExponentialUtility"],
    "agents.BaselineAgents": ["RandomAgent", "FixedActionAgent", "FixedSpreadAgent", "AvellanedaStoikovAgent", "CarteaJaimungalMmAgent", "CarteaJaimungalOeAgent"],
    "gym.wrappers": ["ReduceStateSizeWrapper", "NormaliseASObservation", "RemoveTerminalRewards"],
    "agents.PolicyGradientAgent": ["PolicyGradientAgent"],
}


@pytest.mark.parametrize("module", sorted(MIRRORED))
def test_mirrored_classes_take_the_references_arguments_and_offer_its_methods(module):
    """Drop-in at the source level: for every class on the path, the reference's constructor parameters exist here under the
    same names, in the same order, with the same defaults (ours may ADD keyword-only ones: device, trajectory_offset, ...), and
    every public method / property of the reference's class exists here."""
    import importlib
    import inspect

    ref_mod, our_mod = importlib.import_module("mbt_gym." + module), importlib.import_module("mbt_gym_amd." + module)
    gaps = []
    for name in MIRRORED[module]:
        ref_cls, our_cls = getattr(ref_mod, name), getattr(our_mod, name)
        ref_params = [p for p in inspect.signature(ref_cls.__init__).parameters.values() if p.name != "self"]
        our_params = {p.name: p for p in inspect.signature(our_cls.__init__).parameters.values()}
        our_positional = [p.name for p in our_params.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) and p.name != "self"]
        ref_positional = [p.name for p in ref_params if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        assert our_positional[: len(ref_positional)] == ref_positional, f"{module}.{name}: constructor parameters {our_positional} vs the reference's {ref_positional}"
        for p in ref_params:
            if p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD):
                continue
            assert p.name in our_params, f"{module}.{name}: constructor parameter {p.name} is missing"
            ours = our_params[p.name].default
            same = (ours is p.default) or (isinstance(p.default, np.ndarray) and np.array_equal(ours, p.default)) or \
                   (not isinstance(p.default, np.ndarray) and not isinstance(ours, np.ndarray) and ours == p.default)
            assert same, f"{module}.{name}: default of {p.name} is {ours!r}, the reference's is {p.default!r}"
        public = {m for m, _ in inspect.getmembers(ref_cls) if not m.startswith("_")}
        missing = sorted(m for m in public if not hasattr(our_cls, m))
        if missing:
            gaps.append(f"{name}: {missing}")
    assert not gaps, f"{module}: public members of the reference's classes missing here: {gaps}"


def test_generate_trajectory_and_index_names_match_the_references():
    import importlib
    import inspect

    ref = importlib.import_module("mbt_gym.gym.helpers.generate_trajectory").generate_trajectory
    ours = importlib.import_module("mbt_gym_amd.gym.helpers.generate_trajectory").generate_trajectory
    ref_params, our_params = list(inspect.signature(ref).parameters.values()), list(inspect.signature(ours).parameters.values())
    assert [p.name for p in our_params][: len(ref_params)] == [p.name for p in ref_params]
    assert [p.default for p in our_params][: len(ref_params)] == [p.default for p in ref_params]
    ref_idx, our_idx = importlib.import_module("mbt_gym.gym.index_names"), importlib.import_module("mbt_gym_amd.gym.index_names")
    names = [n for n in dir(ref_idx) if n.isupper()]
    assert names and all(getattr(our_idx, n) == getattr(ref_idx, n) for n in names)


@pytest.mark.parametrize("case", range(12))
def test_instance_attributes_of_environment_processes_and_rewards(case, no_device):
    """What user code READS: every public instance attribute the reference's environment, dynamics, processes and reward
    function carry after construction exists here, and the plain-data ones (numbers, strings, arrays: bounds, step sizes,
    model parameters, initial states) hold the same values."""
    rng = np.random.default_rng(SEED + 27000 + case)
    cfg = random_speed_config(rng, 8) if case % 3 == 2 else random_config(rng, 8)
    with contextlib.redirect_stdout(io.StringIO()):
        ours, ref = make_env(cfg), make_env(cfg, package="mbt_gym")
    gaps, differs = [], []

    def compare(label, a, b):
        for name, value in vars(b).items():
            if name.startswith("_") or name in ("rng", "np_random", "spec"):
                continue
            if not (hasattr(type(a), name) or name in vars(a)):
                gaps.append(f"{label}.{name}")
                continue
            try:
                mine = getattr(a, name)
            except Exception:  # noqa: BLE001 - device-resident data (state matrices): present, not readable without a GPU
                continue
            if isinstance(value, (bool, int, float, str, np.floating, np.integer)) and not callable(mine):
                if not (mine == value or (isinstance(value, float) and np.isclose(mine, value, rtol=1e-12, atol=0))):
                    differs.append(f"{label}.{name}: {mine!r} vs {value!r}")
            elif isinstance(value, np.ndarray) and value.dtype != object and value.size and isinstance(mine, np.ndarray):
                if mine.shape != value.shape or not np.allclose(mine, value, rtol=1e-12, atol=0, equal_nan=True):
                    differs.append(f"{label}.{name}: {mine!r} vs {value!r}")

    compare("env", ours, ref)
    compare("model_dynamics", ours.model_dynamics, ref.model_dynamics)
    compare("reward_function", ours.reward_function, ref.reward_function)
    for proc in ("midprice_model", "arrival_model", "fill_probability_model", "price_impact_model"):
        a, b = getattr(ours.model_dynamics, proc), getattr(ref.model_dynamics, proc)
        assert (a is None) == (b is None), proc
        if b is not None:
            compare(proc, a, b)
    for space in ("observation_space", "action_space"):
        a, b = getattr(ours, space), getattr(ref, space)
        if hasattr(b, "low"):
            np.testing.assert_array_equal(a.low, b.low, err_msg=space)
            np.testing.assert_array_equal(a.high, b.high, err_msg=space)
            assert a.shape == b.shape and a.dtype == b.dtype
    assert not gaps, f"attributes of the reference's objects missing here: {gaps}"
    assert not differs, f"attributes that differ: {differs}"


def test_backtesting_statistics_equal_the_references_on_the_same_trajectory(monkeypatch):
    """gym/backtesting.py: both packages' functions are fed the SAME recorded trajectory (their generate_trajectory is
    replaced by a stub returning it), so what is compared is the three formulas - per lane here, one lane there."""
    import importlib
    import warnings

    ours, ref = importlib.import_module("mbt_gym_amd.gym.backtesting"), importlib.import_module("mbt_gym.gym.backtesting")
    rng = np.random.default_rng(SEED + 29000)
    steps, lanes = 60, 7
    obs = np.zeros((lanes, 4, steps + 1))
    obs[:, 3, :] = 100.0 + np.cumsum(rng.normal(0, 0.5, size=(lanes, steps + 1)), axis=-1)
    obs[:, 1, :] = np.cumsum(rng.integers(-1, 2, size=(lanes, steps + 1)), axis=-1)
    obs[:, 0, :] = 1000.0 + np.cumsum(rng.normal(0, 20.0, size=(lanes, steps + 1)), axis=-1)

    class Env:
        n_steps = steps

        def __init__(self, n):
            self.num_trajectories = n

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name in ("get_sharpe_ratio", "get_sortino_ratio", "get_maximum_drawdown"):
            monkeypatch.setattr(ours, "generate_trajectory", lambda env, agent: (obs, None, None))
            batch = getattr(ours, name)(Env(lanes), None)
            assert batch.shape == (lanes,)
            for lane in range(lanes):
                one = obs[lane:lane + 1]
                monkeypatch.setattr(ref, "generate_trajectory", lambda env, agent, one=one: (one, None, None))
                monkeypatch.setattr(ours, "generate_trajectory", lambda env, agent, one=one: (one, None, None))
                want = getattr(ref, name)(Env(1), None)
                assert getattr(ours, name)(Env(1), None) == pytest.approx(want, rel=1e-12), (name, lane)
                assert batch[lane] == pytest.approx(want, rel=1e-12), (name, lane)


def test_wrappers_equal_the_references_on_a_scripted_environment():
    """gym/wrappers.py: each package's three wrappers around the same scripted environment (its Box built from that package's
    own Box class) return the same observations, rewards, dones and bounds, step by step - including the reference's
    reset / step asymmetry in NormaliseASObservation and its terminal-reward rescaling."""
    import importlib

    ref_w, our_w = importlib.import_module("mbt_gym.gym.wrappers"), importlib.import_module("mbt_gym_amd.gym.wrappers")
    ref_box = importlib.import_module("gym").spaces.box.Box
    our_box = importlib.import_module("mbt_gym_amd.spaces").Box

    class Reward:
        per_step_inventory_aversion, terminal_inventory_aversion = 0.01, 0.5

    def scripted(box):
        class Env:
            metadata, spec = {}, None

            def __init__(self):
                self.observation_space = box(low=np.float32([-10, -4, 0, 90]), high=np.float32([10, 4, 1, 110]))
                self.action_space = box(low=np.float32([0, 0]), high=np.float32([3, 3]))
                self.reward_function, self.num_trajectories, self.k = Reward(), 1, 0

            def _obs(self):
                return np.array([[1.5 + self.k, -2.0, 0.25 * self.k, 101.0 - self.k]])

            def reset(self):
                self.k = 0
                return self._obs()

            def step(self, action):
                self.k += 1
                return self._obs(), np.array([2.0 + self.k]), np.array([self.k == 3]), [{}]

        return Env()

    for name, kwargs in (("ReduceStateSizeWrapper", {}), ("ReduceStateSizeWrapper", {"list_of_state_indices": [3, 0]}), ("NormaliseASObservation", {}),
                         ("RemoveTerminalRewards", {})):
        a, b = getattr(our_w, name)(scripted(our_box), **kwargs), getattr(ref_w, name)(scripted(ref_box), **kwargs)
        np.testing.assert_array_equal(a.observation_space.low, b.observation_space.low, err_msg=name)
        np.testing.assert_array_equal(a.observation_space.high, b.observation_space.high, err_msg=name)
        assert a.observation_space.dtype == b.observation_space.dtype, name
        np.testing.assert_array_equal(a.reset(), b.reset(), err_msg=name)
        for _ in range(3):
            out_a, out_b = a.step(None), b.step(None)
            for x, y in zip(out_a[:3], out_b[:3]):
                np.testing.assert_array_equal(np.asarray(x), np.asarray(y), err_msg=name)


@pytest.mark.parametrize("seed", [1, 7, 50, 12345])
def test_host_side_draws_at_reset_equal_the_references(seed, no_device):
    """What the HOST decides at reset - random initial inventories from the environment's generator (TE:72, TE:270-281; one
    draw is consumed by the constructor, TE:74), the start time from a callable, quantised to the step grid (TE:257-268), the
    seeds handed to the processes (TE:345-348) - against the reference's own environment, reset after reset."""
    from oracle.mbt_oracle import OracleConfig

    n = 37
    cfg = OracleConfig(num_trajectories=n, n_steps=40, terminal_time=2.0, midprice="bm", volatility=1.0, initial_price=100.0, arrival="hawkes",
                       intensity=(10.0, 12.0), hawkes_jump=5.0, hawkes_speed=10.0, fill_exponent=1.5, dynamics="limit", reward="pnl",
                       initial_inventory=(-5, 9), max_inventory=20, seed=seed, normalise_action_space=False, normalise_observation_space=False)
    import itertools

    starts, ref_starts = itertools.cycle([0.0, 0.33, 1.02, 1.94, 0.71]), itertools.cycle([0.0, 0.33, 1.02, 1.94, 0.71])
    with contextlib.redirect_stdout(io.StringIO()):
        ours = make_env(cfg, start_time=lambda: next(starts))
        ref = make_env(cfg, package="mbt_gym", start_time=lambda: next(ref_starts))
    assert [p.seed_ for p in ours.stochastic_processes.values()] == [p.seed_ for p in ref.stochastic_processes.values()] == [seed + 1, seed + 2, seed + 3]
    del no_device[:]  # (the constructor's own reset)
    for _ in range(3):
        ours.reset()
        with contextlib.redirect_stdout(io.StringIO()):
            ref.reset()
        start, q0 = no_device[-1]
        np.testing.assert_array_equal(q0.astype(np.float64), ref.model_dynamics.state[:, 1])
        assert start == ref.model_dynamics.state[0, 2]
    from mbt_gym_amd._native import NativeError

    ref.seed(seed + 100)
    with pytest.raises(NativeError):  # the host side of seed() is done by the time the (absent) device is told its new key
        ours.seed(seed + 100)
    ours.reset()
    with contextlib.redirect_stdout(io.StringIO()):
        ref.reset()
    np.testing.assert_array_equal(no_device[-1][1].astype(np.float64), ref.model_dynamics.state[:, 1])
    assert [p.seed_ for p in ours.stochastic_processes.values()] == [p.seed_ for p in ref.stochastic_processes.values()]


@pytest.mark.parametrize("case", range(6))
def test_host_side_normalisation_maps_equal_the_references(case, no_device):
    """normalise_action / normalise_observation and their inverses (TE:112-126) on random arrays, both packages' environments
    of the same random market: same float32 Box bounds, same arithmetic, equal results."""
    rng = np.random.default_rng(SEED + 31000 + case)
    cfg = random_speed_config(rng, 16) if case % 3 == 2 else random_config(rng, 16)
    cfg.normalise_action_space = cfg.normalise_observation_space = cfg.dynamics != "touch"
    with contextlib.redirect_stdout(io.StringIO()):
        ours, ref = make_env(cfg), make_env(cfg, package="mbt_gym")
    act = rng.uniform(-1, 1, size=(16, cfg.action_dim))
    obs = rng.uniform(-1, 1, size=(16, ours.observation_space.shape[0]))
    for inverse in (False, True):
        np.testing.assert_array_equal(ours.normalise_action(act, inverse=inverse), ref.normalise_action(act, inverse=inverse))
        with np.errstate(divide="ignore", invalid="ignore"):  # a constant midprice has a zero-width Box column on both sides
            np.testing.assert_array_equal(ours.normalise_observation(obs, inverse=inverse), ref.normalise_observation(obs, inverse=inverse))
    np.testing.assert_array_equal(ours.normalise_rewards(act[:, 0]), ref.normalise_rewards(act[:, 0]))
