"""Round 6: graph-capturable stepping - the clock on the device (include/mbt_env.h: mbt_env_device_clock_begin ...).

What is held here:
  * a loop of mbt_env_step_device_captured is the mbt_env_step_many_device(auto_reset) loop TO THE BIT, over two episode ends, for
    every kernel family that steps without the host (float32 tiers, precise_state, exogenous fills, speed dynamics, run-time
    compiled plugins): state, remainders, observation, rewards, clock, episode log, clip count;
  * the same against the float64 oracle on the kernel's own draws, across an episode end (the terminal observation included);
  * a HIP graph captured ONCE and replayed steps the environment on: raw HIP (hipStreamBeginCapture through ctypes) and
    torch.cuda.graph with a torch policy writing the action buffer in place;
  * what the mode refuses.
Reference for the contract: gym/StableBaselinesTradingEnvironment.py:25-37 (SB3's VecEnv step: auto-reset + terminal_observation)."""
import ctypes as C

import numpy as np
import pytest

from mbt_gym_amd import _native
from oracle.mbt_oracle import InjectedNoise, OracleConfig, OracleEnv
from tests.env_factory import make_env

pytestmark = pytest.mark.gpu


def _cfg(n, **kw):
    base = dict(num_trajectories=n, n_steps=8, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0, arrival="poisson",
                intensity=(140.0, 140.0), fill_exponent=1.5, dynamics="limit", reward="pnl", initial_inventory=0, max_inventory=50, seed=50,
                normalise_action_space=False, normalise_observation_space=False)
    base.update(kw)
    return OracleConfig(**base)


HAWKES = dict(arrival="hawkes", intensity=(10.0, 10.0), hawkes_jump=40.0, hawkes_speed=6.0, midprice="ou", ou_level=100.0, ou_speed=0.01)
SPEED = dict(dynamics="speed", arrival="none", reward="cjoe", phi=0.01, alpha=0.001, initial_inventory=1, max_inventory=10)

# name -> (OracleConfig overrides, TradingEnvironment overrides): one member of every kernel family that has a captured form
FAMILIES = {
    "as_f32": (dict(), dict()),
    "cjp_quadratic": (dict(reward="cjmm", phi=0.01, alpha=0.001), dict()),
    "cjmm_random_initial_inventory": (dict(reward="cjmm", phi=0.01, alpha=0.001, initial_inventory=(-3, 4)), dict()),
    "running_general_exponent": (dict(reward="running", phi=0.01, alpha=0.001, inventory_exponent=4.0), dict()),
    "hawkes_exact_ou": (HAWKES, dict()),
    "hawkes_f32_intensities": (HAWKES, dict(hawkes_float32_intensities=True)),
    "limit_and_market": (dict(dynamics="limit_and_market", market_half_spread=0.5), dict()),
    "touch": (dict(dynamics="touch", market_half_spread=0.5), dict()),
    "normalised": (dict(normalise_action_space=True, normalise_observation_space=True), dict()),
    "exogenous_fill": (dict(fill="exogenous", base_fill_probability=0.8, exo_depth=(0.3, 0.4), exo_depth_lo=(0.1, 0.2), exo_depth_hi=(0.6, 0.7), normalise_observation_space=True), dict()),
    "precise_as": (dict(), dict(precise_state=True)),
    "precise_hawkes_normalised": (dict(normalise_observation_space=True, **HAWKES), dict(precise_state=True)),
    "clip_cash": (dict(max_cash=3.0, initial_cash=0.0), dict()),
    "speed_temp_power": (dict(impact="temp_power", **SPEED), dict()),
    "speed_cubic_impact_and_quartic_penalty": (dict(impact="temp_power", impact_exponent=3.0, inventory_exponent=4.0, **SPEED), dict()),  # the instantiation with powers (power_f32, wave priorities)
    "speed_transient_state": (dict(impact="temp_transient", **SPEED), dict()),
    "speed_precise_state": (dict(impact="temp_perm", **SPEED), dict(precise_state=True)),
    "user_power_law_fill": (dict(fill="user_power_law", fill_scale=1.25, fill_power=1.5), dict()),
    "user_cev_midprice": (dict(midprice="user_cev", drift=0.05, volatility=0.6, cev_gamma=0.75, initial_price=50.0, midprice_lo=20.0, midprice_hi=80.0), dict()),
}


def _actions(cfg, n, seed=3):
    rng = np.random.default_rng(seed)
    a = cfg.action_dim
    if cfg.dynamics == "speed":
        return rng.uniform(-1.0, 1.0, size=(n, a)).astype(np.float32)
    if cfg.dynamics == "touch":
        return rng.integers(0, 2, size=(n, a)).astype(np.float32)
    act = rng.uniform(0.05, 1.2, size=(n, a)).astype(np.float32)
    if cfg.dynamics == "limit_and_market":
        act[:, 2:] = (rng.uniform(size=(n, 2)) < 0.1).astype(np.float32)
    if cfg.normalise_action_space:
        act = rng.uniform(-1.0, 1.0, size=(n, a)).astype(np.float32)
    return act


def _snapshot(env):
    """Everything a step leaves behind, as host arrays."""
    import torch

    env.synchronize()
    out = {"state": env.state.copy(), "obs": env.observation_host().copy(), "reward": torch.as_tensor(env.reward_device, device="cuda").cpu().numpy().copy(),
           "clip_count": env.clip_count, "clock": env.clock}
    return out


def _pop_all(env):
    out = []
    while True:
        sums = env.episode_log_pop(wait=True)
        if sums is None:
            return out
        out.append(sums)


@pytest.mark.parametrize("name", sorted(FAMILIES))
@pytest.mark.parametrize("n", [1000, 5000])
def test_captured_steps_are_the_step_many_device_loop_to_the_bit(name, n):
    """k = 2 episodes + 3 steps through both routes from the same seed and reset: the device's clock, its in-kernel reset and its
    episode log against the host's (launch_step, log_push, do_reset).  n = 1000: two (one, for speed dynamics) workgroups and pad
    lanes; n = 5000: ten workgroups, pad lanes."""
    overrides, env_overrides = FAMILIES[name]
    cfg = _cfg(n, **overrides)
    steps = 2 * cfg.n_steps + 3
    action = _actions(cfg, n)
    snapshots, logs, states64 = [], [], []
    for route in ("host clock", "device clock"):
        env = make_env(cfg, noise="philox", **env_overrides)
        env.track_lane_returns(True)
        env.reset_device()
        env.set_action_host(action)
        if route == "host clock":
            assert env.step_many_device(steps, auto_reset=True) == (steps, 2)
        else:
            env.device_clock_begin(auto_reset=True)
            for _ in range(steps):
                env.step_device_captured()
            now = env.device_clock_read()
            assert (now["steps"], now["episodes"], now["log_count"], now["episode_step"], now["done"]) == (steps, 2, 2, 3, 0)
            assert env.clock == (now["time"], now["episode_step"], now["philox_step"])  # (mbt_env_get_clock reads the device's clock in the mode)
            env.device_clock_end()
        snapshots.append(_snapshot(env))
        states64.append(env.state64.copy())
        logs.append(_pop_all(env))
        # ... and the host's clock carries on from where the device left it: three more steps through the ordinary route
        env.step_many_device(3, auto_reset=True)
        snapshots.append(_snapshot(env))
        env.close()
    for a, b in ((snapshots[0], snapshots[2]), (snapshots[1], snapshots[3])):
        for key in ("state", "obs", "reward"):
            np.testing.assert_array_equal(a[key], b[key], err_msg=f"{name}: {key}")
        assert a["clip_count"] == b["clip_count"] and a["clock"] == b["clock"], name
    np.testing.assert_array_equal(states64[0], states64[1], err_msg=f"{name}: float64 state (rows + remainders)")
    assert len(logs[0]) == len(logs[1]) == 2
    for x, y in zip(logs[0], logs[1]):
        np.testing.assert_array_equal(x, y, err_msg=f"{name}: episode log [sum R, sum R^2, lanes]")
        assert np.isfinite(x).all() and x[2] == n


def test_without_auto_reset_the_clock_runs_on_like_the_hosts():
    """No reset at the episode's end: `done` is raised and the clock carries on past the terminal time, as launch_step's does."""
    n = 1000
    cfg = _cfg(n, n_steps=5)
    action = _actions(cfg, n)
    out = []
    for route in ("host", "device"):
        env = make_env(cfg, noise="philox")
        env.reset_device()
        env.set_action_host(action)
        if route == "host":
            dones = [env.step_device() for _ in range(7)]
            assert dones == [False] * 4 + [True] * 3
        else:
            env.device_clock_begin(auto_reset=False)
            dones = []
            for _ in range(7):
                env.step_device_captured()
                dones.append(bool(env.device_clock_read()["done"]))
            assert dones == [False] * 4 + [True] * 3
            env.device_clock_end()
        out.append(_snapshot(env))
        env.close()
    for key in ("state", "obs", "reward"):
        np.testing.assert_array_equal(out[0][key], out[1][key])
    assert out[0]["clock"] == out[1]["clock"]


@pytest.mark.parametrize("family", ["as_f32", "hawkes_exact_ou", "normalised"])
def test_captured_episode_against_the_oracle_on_the_kernels_own_draws(family):
    """One episode + two steps of the next in device-clock mode against the float64 oracle fed with the draws the kernel used (the
    Philox step counts on across the reset, TE:96-101 does not reseed): rewards within 1e-5 and inventory exact at every step, the
    terminal observation is the oracle's last state, the observation after the episode's end is a reset's."""
    import torch

    n = 2048
    overrides, env_overrides = FAMILIES[family]
    cfg = _cfg(n, n_steps=7, **overrides)
    total = cfg.n_steps + 2
    env = make_env(cfg, noise="philox", **env_overrides)
    draws = [_native.rng_fill(cfg.seed, 0, k, n) for k in range(total)]
    oracle = OracleEnv(cfg, InjectedNoise(*[np.stack(x) for x in zip(*draws)]))
    first = env.reset().copy()
    oracle.reset()
    action = _actions(cfg, n)
    env.set_action_host(action)
    env.device_clock_begin(auto_reset=True, keep_terminal_observation=True)
    reward = torch.as_tensor(env.reward_device, device="cuda")
    for k in range(total):
        env.step_device_captured()
        env.synchronize()
        o_obs, o_rew, o_done = oracle.step(action.astype(np.float64))
        obs = env.observation_host()
        np.testing.assert_allclose(reward.cpu().numpy(), o_rew, rtol=0, atol=1e-5, err_msg=f"step {k}: reward")
        if k == cfg.n_steps - 1:  # the episode's end: the rows were reset inside the launch, the last observation was kept aside
            assert o_done.all()
            terminal = torch.as_tensor(env.terminal_obs_device, device="cuda").cpu().numpy()
            np.testing.assert_allclose(terminal, o_obs, rtol=1e-6, atol=3e-4, err_msg="terminal observation")
            if not cfg.normalise_observation_space:
                np.testing.assert_array_equal(terminal[:, 1].astype(np.float64), o_obs[:, 1], err_msg="terminal inventory")
            np.testing.assert_array_equal(obs, first, err_msg="the observation after the episode's end is the reset's")
            oracle.reset()  # (the oracle's noise cursor counts on, like the Philox step)
        else:
            np.testing.assert_allclose(obs, o_obs, rtol=1e-6, atol=3e-4, err_msg=f"step {k}: observation")
            if not cfg.normalise_observation_space:
                np.testing.assert_array_equal(obs[:, 1].astype(np.float64), o_obs[:, 1], err_msg=f"step {k}: inventory")
    env.device_clock_end()
    env.close()


class _Hip:
    """The few HIP runtime entry points a raw stream capture needs, through the runtime libmbtenv.so itself is bound to."""

    def __init__(self):
        self.rt = C.CDLL(_native.LIB_PATH)  # (dlsym on the handle also searches its dependencies: libamdhip64)
        for name, args in (("hipStreamCreate", [C.POINTER(C.c_void_p)]), ("hipStreamDestroy", [C.c_void_p]), ("hipStreamBeginCapture", [C.c_void_p, C.c_int]),
                           ("hipStreamEndCapture", [C.c_void_p, C.POINTER(C.c_void_p)]), ("hipGraphInstantiate", [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
                           ("hipGraphLaunch", [C.c_void_p, C.c_void_p]), ("hipGraphExecDestroy", [C.c_void_p]), ("hipGraphDestroy", [C.c_void_p]), ("hipStreamSynchronize", [C.c_void_p])):
            fn = getattr(self.rt, name)
            fn.argtypes, fn.restype = args, C.c_int

    def call(self, name, *args):
        code = getattr(self.rt, name)(*args)
        assert code == 0, f"{name} -> hipError {code}"


@pytest.mark.parametrize("family,n", [("as_f32", 1000), ("as_f32", 1 << 16), ("hawkes_exact_ou", 5000), ("speed_transient_state", 3000)])
def test_a_hip_graph_captured_once_steps_the_environment_on_every_replay(family, n):
    """hipStreamBeginCapture, k launches of mbt_env_step_device_captured, hipStreamEndCapture; the graph replayed m times = k m steps
    of the ordinary loop, episode ends (k does not divide the episode) and their log included.  No torch."""
    hip = _Hip()
    overrides, env_overrides = FAMILIES[family]
    cfg = _cfg(n, n_steps=11, **overrides)
    k, replays = 4, 9  # 36 steps: three episode ends, none at a graph boundary
    action = _actions(cfg, n)
    reference = make_env(cfg, noise="philox", **env_overrides)
    reference.reset_device()
    reference.set_action_host(action)
    assert reference.step_many_device(k * replays, auto_reset=True) == (k * replays, 3)
    want, want_log = _snapshot(reference), _pop_all(reference)
    reference.close()

    env = make_env(cfg, noise="philox", **env_overrides)
    env.reset_device()
    env.set_action_host(action)
    stream, graph, executable = C.c_void_p(), C.c_void_p(), C.c_void_p()
    hip.call("hipStreamCreate", C.byref(stream))
    env.set_stream(stream.value)
    env.device_clock_begin(auto_reset=True)
    hip.call("hipStreamBeginCapture", stream, 0)  # hipStreamCaptureModeGlobal
    for _ in range(k):
        env.step_device_captured()
    hip.call("hipStreamEndCapture", stream, C.byref(graph))
    assert env.device_clock_read()["steps"] == 0, "a capture records launches, it does not run them"
    hip.call("hipGraphInstantiate", C.byref(executable), graph, None, None, 0)
    for _ in range(replays):
        hip.call("hipGraphLaunch", executable, stream)
    now = env.device_clock_read()
    assert (now["steps"], now["episodes"]) == (k * replays, 3)
    env.device_clock_end()
    got, got_log = _snapshot(env), _pop_all(env)
    for key in ("state", "obs", "reward"):
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    assert got["clock"] == want["clock"]
    assert len(got_log) == len(want_log) == 3
    for x, y in zip(got_log, want_log):
        np.testing.assert_array_equal(x, y)
    hip.call("hipGraphExecDestroy", executable)
    hip.call("hipGraphDestroy", graph)
    env.close()
    hip.call("hipStreamDestroy", stream)


def test_torch_cuda_graph_of_policy_and_step_replays_the_eager_loop():
    """The consumer the mode exists for (SBE:25-37): a torch policy on the observation view writes the action view in place, the step
    follows on the same stream; torch.cuda.graph captures k of those pairs and every replay continues the episode - bit-identical to
    the eager loop of the same torch ops and mbt_env_step_device."""
    import torch

    n, k, replays = 4096, 5, 7
    cfg = _cfg(n, n_steps=16, normalise_action_space=True, normalise_observation_space=True)
    torch.manual_seed(0)
    weight = (torch.randn(4, 2, device="cuda") * 0.3).contiguous()
    bias = torch.tensor([0.1, -0.2], device="cuda")

    def act(obs_t, action_t):
        action_t.copy_(torch.tanh(obs_t @ weight + bias))

    eager = make_env(cfg, noise="philox")
    eager.reset_device()
    obs_t, action_t = torch.as_tensor(eager.obs_device, device="cuda"), torch.as_tensor(eager.action_device, device="cuda")
    eager.set_stream(torch.cuda.current_stream().cuda_stream)
    for step in range(k * replays):
        act(obs_t, action_t)
        if eager.step_device():
            eager.reset_device()
    want = _snapshot(eager)
    eager.close()

    env = make_env(cfg, noise="philox")
    env.reset_device()
    side = torch.cuda.Stream()
    env.set_stream(side.cuda_stream)
    env.device_clock_begin(auto_reset=True)
    obs_t, action_t = torch.as_tensor(env.obs_device, device="cuda"), torch.as_tensor(env.action_device, device="cuda")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(k):
            act(obs_t, action_t)
            env.step_device_captured()
    for _ in range(replays):
        graph.replay()
    torch.cuda.synchronize()
    now = env.device_clock_read()
    assert (now["steps"], now["episodes"]) == (k * replays, 2)
    # the clock block itself, for device code that wants `done` or the step count without a host round trip (mbt_env_device_clock_ptr):
    # eight int32 words laid out as struct mbt_device_clock
    words = torch.as_tensor(env.device_clock_view, device="cuda").cpu().numpy()
    assert (int(words[2]), int(words[3]), int(words[4]), int(words[5]), int(words[6])) == (now["episode_step"], now["philox_step"], now["steps"], now["episodes"], now["done"])
    assert words[:2].view(np.float64)[0] == now["time"]
    env.device_clock_end()
    got = _snapshot(env)
    for key in ("state", "obs", "reward"):
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    assert got["clock"] == want["clock"]
    del graph
    env.close()


def test_episodes_that_end_in_the_mode_go_through_the_communicator():
    """one 24-byte all-reduce per episode.  In device-clock mode the launch that ends an episode files the sums on the device;
    mbt_env_device_clock_end sends each of them through the environment's communicator (every rank replays the same graph, so every rank
    files the same number) before it enters the episode log.  One RCCL rank here (RCCL refuses two ranks on one device): the collective is
    enqueued on the environment's stream for every episode, and the log equals the host-clock loop's with the same communicator, to the bit."""
    from mbt_gym_amd.distributed import RcclCommunicator

    n = 4096
    cfg = _cfg(n, n_steps=9)
    action = _actions(cfg, n)
    comm = RcclCommunicator(rank=0, world_size=1, device=0)
    logs, clocks = [], []
    for route in ("host clock", "device clock"):
        env = make_env(cfg, noise="philox")
        env.set_communicator(comm)
        env.reset_device()
        env.set_action_host(action)
        if route == "host clock":
            assert env.step_many_device(30, auto_reset=True) == (30, 3)
        else:
            env.device_clock_begin(auto_reset=True)
            for _ in range(30):
                env.step_device_captured()
            env.device_clock_end()
        logs.append(_pop_all(env))
        clocks.append(env.clock)
        env.set_communicator(None)
        env.close()
    assert clocks[0] == clocks[1]
    assert len(logs[0]) == len(logs[1]) == 3
    for x, y in zip(logs[0], logs[1]):
        np.testing.assert_array_equal(x, y)
        assert x[2] == n and np.isfinite(x[0])
    comm.close()


def test_what_the_mode_refuses():
    n = 1000
    cfg = _cfg(n)
    env = make_env(cfg, noise="philox")
    with pytest.raises(_native.NativeError, match="outside mbt_env_device_clock_begin"):
        env.step_device_captured()
    env.reset_device()
    env.device_clock_begin(auto_reset=True)
    for call in (env.step_device, env.reset_device, lambda: env.step_many_device(2), lambda: env.seed(3), lambda: env.step(_actions(cfg, n)),
                 lambda: env.record_events(True), lambda: env.device_clock_begin()):
        with pytest.raises(_native.NativeError, match="mbt_env_device_clock_end first"):
            call()
    env.step_device_captured()
    assert env.clock[1:] == (1, 1)
    env.device_clock_end()
    env.device_clock_end()  # a second end is a no-op
    assert env.step_device() is False and env.clock[1:] == (2, 2)
    with pytest.raises(_native.NativeError, match="needs MBT_CLOCK_AUTO_RESET"):
        env.device_clock_begin(auto_reset=False, keep_terminal_observation=True)
    with pytest.raises(RuntimeError, match="keep_terminal_observation"):
        env.terminal_obs_device
    env.close()
    injected = make_env(cfg, noise="injected")
    injected.reset_device()
    with pytest.raises(_native.NativeError, match="injected noise"):
        injected.device_clock_begin()
    injected.close()


def test_the_resident_kernel_lives_across_the_wrap_of_its_sequence_numbers(monkeypatch):
    """the host's sequence counter skips the value that means "leave" (0xFFFFFFFF) when it wraps; the resident kernel's own
    count has to skip it too, or a kernel alive across the wrap answers a number the host is not waiting for.  The test hook starts the
    counter twelve steps before the wrap; thirty steps through the resident kernel equal the one-launch-per-step path."""
    n = 1000
    cfg = _cfg(n, n_steps=40)
    action = _actions(cfg, n)
    reference = make_env(cfg, noise="philox")
    reference.reset()
    want = [reference.step(action) for _ in range(30)]
    reference.close()
    monkeypatch.setenv("MBT_TEST_FLAG_SEQ", "0xFFFFFFF2")
    monkeypatch.setenv("MBT_RESIDENT_IDLE_US", "2000000")  # (the kernel must not leave by itself between two steps of the test)
    env = make_env(cfg, noise="philox", resident_step=True)
    env.reset()
    import time

    t0 = time.perf_counter()
    for k in range(30):
        obs, rew, done, _ = env.step(action)
        np.testing.assert_array_equal(obs, want[k][0], err_msg=f"step {k}")
        np.testing.assert_array_equal(rew, want[k][1], err_msg=f"step {k}")
    assert time.perf_counter() - t0 < 0.15, "a step waited for its 200 ms answer time-out: the kernel answered another sequence number"
    env.close()


def test_graphs_of_any_length_replayed_in_any_order_with_single_calls_between_them():
    """The clock has two slots and each captured launch carries the slot it reads (step_kernel.hpp: CapturedParams::parity); an align
    kernel in front of every capture's first step - and of every call outside a capture - brings the current slot to 0.  So: a graph of
    THREE steps (odd: it ends on the other slot), a graph of TWO, a graph of ONE and single calls, replayed in a scrambled order, take
    exactly the steps the ordinary loop takes - episode ends (n_steps = 7 divides none of it) and their log included."""
    hip = _Hip()
    n = 3000
    cfg = _cfg(n, n_steps=7, reward="cjmm", phi=0.01, alpha=0.001, initial_inventory=(-3, 4))
    action = _actions(cfg, n)
    env = make_env(cfg, noise="philox")
    env.reset_device()
    env.set_action_host(action)
    stream = C.c_void_p()
    hip.call("hipStreamCreate", C.byref(stream))
    env.set_stream(stream.value)
    env.device_clock_begin(auto_reset=True)
    graphs = {}
    for k in (3, 2, 1):
        graph, executable = C.c_void_p(), C.c_void_p()
        hip.call("hipStreamBeginCapture", stream, 0)
        for _ in range(k):
            env.step_device_captured()
        hip.call("hipStreamEndCapture", stream, C.byref(graph))
        hip.call("hipGraphInstantiate", C.byref(executable), graph, None, None, 0)
        graphs[k] = (graph, executable)
    order = [3, 3, 1, 0, 2, 3, 0, 0, 1, 1, 3, 2, 2, 0, 3, 1, 3, 3, 2, 0, 1, 3]  # 0 = a single call outside any capture
    total = 0
    for k in order:
        if k == 0:
            env.step_device_captured()
            total += 1
        else:
            hip.call("hipGraphLaunch", graphs[k][1], stream)
            total += k
        assert env.device_clock_read()["steps"] == total  # (reading waits for the stream: every piece has run before the next is issued ...
    for k in order:  # ... and once more without waiting in between: the pieces queue up behind each other on the stream)
        if k == 0:
            env.step_device_captured()
            total += 1
        else:
            hip.call("hipGraphLaunch", graphs[k][1], stream)
            total += k
    now = env.device_clock_read()
    assert (now["steps"], now["episodes"], now["episode_step"]) == (total, total // 7, total % 7)
    env.device_clock_end()
    got, got_log = _snapshot(env), _pop_all(env)
    for graph, executable in graphs.values():
        hip.call("hipGraphExecDestroy", executable)
        hip.call("hipGraphDestroy", graph)
    env.close()
    hip.call("hipStreamDestroy", stream)

    reference = make_env(cfg, noise="philox")
    reference.reset_device()
    reference.set_action_host(action)
    assert reference.step_many_device(total, auto_reset=True) == (total, total // 7)
    want, want_log = _snapshot(reference), _pop_all(reference)
    reference.close()
    for key in ("state", "obs", "reward"):
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    assert got["clock"] == want["clock"]
    assert len(got_log) == len(want_log) == min(16, total // 7)
    for x, y in zip(got_log, want_log):
        np.testing.assert_array_equal(x, y)


FUZZ_SCALE = int(__import__("os").environ.get("MBT_FUZZ_SCALE", "1"))
FUZZ_SEED = int(__import__("os").environ.get("MBT_FUZZ_SEED", "0"))


@pytest.mark.parametrize("case", range(40 * FUZZ_SCALE))
def test_random_configurations_step_the_same_from_the_devices_clock(case):
    """Over the plugin space nobody picked by hand (tests/random_configs.py: random midprice / arrival / fill / dynamics / reward kinds and
    parameters, late start times, tight limits, random initial inventories; order-book and trading-speed families; every third case in the
    precise_state tier): a captured loop over two episode ends and three steps leaves the state, the float64 state, the rewards, the clock, the
    clip count and the episode log of the ordinary loop.  MBT_FUZZ_SCALE / MBT_FUZZ_SEED turn it into a soak."""
    from tests.random_configs import random_actions, random_config, random_speed_actions, random_speed_config

    rng = np.random.default_rng(FUZZ_SEED + 61000 + case)
    n = int(rng.choice([7, 600, 1500]))
    speed = case % 4 == 3
    cfg = random_speed_config(rng, n) if speed else random_config(rng, n)
    cfg.n_steps = int(rng.choice([5, 9, 16]))
    if cfg.arrival == "hawkes" and cfg.hawkes_speed * (cfg.arrival_step_size or cfg.terminal_time / cfg.n_steps) >= 1.0:
        cfg.hawkes_speed = 0.5 * cfg.n_steps / cfg.terminal_time
    precise = case % 3 == 2
    action = (random_speed_actions if speed else random_actions)(rng, cfg, 1)[0]
    steps = 2 * cfg.n_steps + 3
    results = []
    for route in ("host clock", "device clock"):
        env = make_env(cfg, noise="philox", precise_state=precise)
        env.track_lane_returns(True)
        env.reset_device()
        env.set_action_host(action)
        first_done = None
        if route == "host clock":
            taken, ended = env.step_many_device(steps, auto_reset=True)
        else:
            env.device_clock_begin(auto_reset=True)
            for _ in range(steps):
                env.step_device_captured()
            now = env.device_clock_read()
            taken, ended = now["steps"], now["episodes"]
            env.device_clock_end()
        snap = _snapshot(env)
        snap["state64"], snap["log"], snap["counts"] = env.state64.copy(), _pop_all(env), (taken, ended)
        results.append(snap)
        env.close()
    a, b = results
    assert a["counts"] == b["counts"] and a["clock"] == b["clock"] and a["clip_count"] == b["clip_count"], (case, a["counts"], b["counts"], a["clock"], b["clock"])
    for key in ("state", "obs", "reward", "state64"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=f"case {case}: {key}")
    assert len(a["log"]) == len(b["log"]) == a["counts"][1]
    for x, y in zip(a["log"], b["log"]):
        np.testing.assert_array_equal(x, y, err_msg=f"case {case}: episode log")
