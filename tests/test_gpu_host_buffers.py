"""The host buffers of `env.step(np.ndarray)`: outputs live in pinned memory that the step's DMA copies write directly and
that is re-used ACROSS steps - yet, like the reference's fresh arrays (TE:101, TE:110), an array the caller keeps keeps its
values (mbt_gym_amd/_native.py: OutputPool; include/mbt_env.h: mbt_host_alloc)."""
import ctypes as C

import numpy as np
import pytest

from oracle.mbt_oracle import OracleConfig
from tests.env_factory import make_env

pytestmark = pytest.mark.gpu


def _cfg(n, **kw):
    base = dict(num_trajectories=n, n_steps=40, terminal_time=1.0, volatility=2.0, initial_price=100.0, intensity=(140.0, 140.0), fill_exponent=1.5,
                initial_inventory=0, max_inventory=50, seed=11, normalise_action_space=False, normalise_observation_space=False)
    base.update(kw)
    return OracleConfig(**base)


@pytest.mark.parametrize("n", [600, 70000])  # the mapped-memory path of small batches and the DMA path
def test_outputs_are_reused_but_never_while_the_caller_holds_them(n):
    env, ref = make_env(_cfg(n)), make_env(_cfg(n))
    action = np.tile(np.array([[0.6, 0.8]], np.float32), (n, 1))
    obs0 = env.reset()
    ref.reset()
    kept = []  # a caller that keeps every third step's outputs
    expected = []
    for k in range(12):
        obs, rew, done, _ = env.step(action)
        r_obs, r_rew, r_done, _ = ref.step(action)
        np.testing.assert_array_equal(obs, r_obs)
        np.testing.assert_array_equal(rew, r_rew)
        if k % 3 == 0:
            kept.append((obs, rew, done))
            expected.append((r_obs.copy(), r_rew.copy(), r_done.copy()))
    for (obs, rew, done), (e_obs, e_rew, e_done) in zip(kept, expected):  # nothing the caller kept was written over
        np.testing.assert_array_equal(obs, e_obs)
        np.testing.assert_array_equal(rew, e_rew)
        np.testing.assert_array_equal(done, e_done)
    np.testing.assert_array_equal(obs0[:, 2], 0.0)
    held = len(kept) + 1
    assert len(env._host_buffers()["obs"].buffers) <= held + 2
    # the plain loop (names rebound every step) alternates between two buffers: nothing is allocated any more
    del kept, obs, rew, done, obs0
    before = len(env._host_buffers()["obs"].buffers)
    seen = set()
    for _ in range(10):
        obs, rew, done, _ = env.step(action)
        ref.step(action)
        seen.add(obs.ctypes.data)
    assert len(env._host_buffers()["obs"].buffers) == before and len(seen) <= 2
    # a view (or a torch tensor made from one) keeps its buffer out of circulation just the same
    view = obs[5:9, 1]
    snapshot = view.copy()
    del obs
    for _ in range(4):
        o2, _, _, _ = env.step(action)
        ref.step(action)
    np.testing.assert_array_equal(view, snapshot)
    env.close(), ref.close()


def test_action_buffer_pageable_and_float64_actions_give_the_same_step():
    n = 70000
    envs = [make_env(_cfg(n)) for _ in range(3)]
    rng = np.random.default_rng(0)
    for e in envs:
        e.reset()
    for k in range(5):
        a32 = rng.uniform(0.2, 1.2, size=(n, 2)).astype(np.float32)
        envs[0].action_buffer[:] = a32  # written in place: DMA-copied as it is
        out0 = envs[0].step(envs[0].action_buffer)
        out1 = envs[1].step(a32)                      # pageable float32
        out2 = envs[2].step(a32.astype(np.float64))   # the reference's dtype
        for o in (out1, out2):
            np.testing.assert_array_equal(out0[0], o[0])
            np.testing.assert_array_equal(out0[1], o[1])
    with pytest.raises(ValueError):
        envs[0].step(np.zeros((n, 3), np.float32))
    for e in envs:
        e.close()


def test_the_c_abi_takes_pinned_and_pageable_host_pointers_alike():
    """mbt_env_step_host with buffers from mbt_host_alloc (direct DMA) and with ordinary memory (bounced): same results."""
    from mbt_gym_amd import _native

    lib = _native.load_library()
    n = 50000
    env_a, env_b = make_env(_cfg(n)), make_env(_cfg(n))
    env_a.reset(), env_b.reset()
    action = np.tile(np.array([[0.5, 0.9]], np.float32), (n, 1))
    pinned = {name: _native.PinnedBuffer(shape) for name, shape in (("act", (n, 2)), ("obs", (n, 4)), ("rew", (n,)))}
    p_act, p_obs, p_rew = (pinned[k].array() for k in ("act", "obs", "rew"))
    p_act[:] = action
    obs, rew = np.empty((n, 4), np.float32), np.empty((n,), np.float32)
    done = C.c_int32(0)
    for _ in range(3):
        _native.check(lib.mbt_env_step_host(env_a._handle, _native.fptr(p_act), _native.fptr(p_obs), _native.fptr(p_rew), C.byref(done)))
        _native.check(lib.mbt_env_step_host(env_b._handle, _native.fptr(action), _native.fptr(obs), _native.fptr(rew), C.byref(done)))
        np.testing.assert_array_equal(p_obs, obs)
        np.testing.assert_array_equal(p_rew, rew)
    # mixed: pinned outputs, pageable action - and NULL outputs
    _native.check(lib.mbt_env_step_host(env_a._handle, _native.fptr(action), _native.fptr(p_obs), None, C.byref(done)))
    _native.check(lib.mbt_env_step_host(env_b._handle, _native.fptr(p_act), None, _native.fptr(rew), C.byref(done)))
    np.testing.assert_array_equal(p_obs, env_b.observation_host())
    env_a.close(), env_b.close()


def test_sb3_adapter_auto_reset_hands_out_terminal_observations_that_stay_valid():
    from mbt_gym_amd.gym.StableBaselinesTradingEnvironment import StableBaselinesTradingEnvironment

    n = 3000
    cfg = _cfg(n, n_steps=6)
    venv, ref = StableBaselinesTradingEnvironment(make_env(cfg)), make_env(cfg)
    action = np.tile(np.array([[0.6, 0.8]], np.float32), (n, 1))
    venv.reset(), ref.reset()
    terminal = None
    for k in range(14):
        obs, rew, dones, infos = venv.step(action)
        r_obs, r_rew, r_dones, _ = ref.step(action)
        np.testing.assert_array_equal(rew, r_rew)
        if r_dones[0]:
            assert dones.all()
            terminal = (infos, r_obs.copy())
            np.testing.assert_array_equal(obs, ref.reset())  # the observation of the automatic reset (SBE:36)
        else:
            np.testing.assert_array_equal(obs, r_obs)
    infos, want = terminal  # kept across later steps: still the terminal observation of that episode
    np.testing.assert_array_equal(np.stack([infos[i]["terminal_observation"] for i in (0, 7, n - 1)]), want[[0, 7, n - 1]])
    venv.close(), ref.close()


def test_recorded_rollouts_reuse_pinned_arrays_only_after_the_caller_dropped_them():
    """env.rollout(record=True) hands out arrays over pooled pinned memory: a recording the caller keeps keeps its values, one that
    was dropped lends its memory to the next (no fresh gigabytes per episode)."""
    from mbt_gym_amd.agents.BaselineAgents import FixedSpreadAgent

    n = 3000
    env, ref = make_env(_cfg(n, n_steps=25), noise="philox"), make_env(_cfg(n, n_steps=25), noise="philox")
    agent = FixedSpreadAgent(env, half_spread=0.7)
    env.reset()
    kept = env.rollout(agent, record=True)[:3]          # episode 1, kept
    kept_copy = [a.copy() for a in kept]
    env.reset()
    second = env.rollout(agent, record=True)[:3]        # episode 2 (Philox counters moved on): other memory than episode 1
    assert all(a.ctypes.data != b.ctypes.data for a, b in zip(kept, second))
    for a, b in zip(kept, kept_copy):
        np.testing.assert_array_equal(a, b)
    address = [a.ctypes.data for a in second]
    second_copy = [a.copy() for a in second]
    del second
    env.reset()
    third = env.rollout(agent, record=True)[:3]         # the dropped recording's memory comes back
    assert [a.ctypes.data for a in third] == address
    for a, b in zip(kept, kept_copy):
        np.testing.assert_array_equal(a, b)
    # and the values are those of the step loop on a second environment with the same seed (which pools nothing of this kind)
    action = np.tile(np.array([[0.7, 0.7]], np.float32), (n, 1))
    for recorded in (kept, second_copy, third):
        obs = ref.reset()
        np.testing.assert_array_equal(recorded[0][0], obs)
        for k in range(25):
            obs, rew, done, _ = ref.step(action)
            np.testing.assert_array_equal(recorded[0][k + 1], obs)
            np.testing.assert_array_equal(recorded[2][k], rew)
            np.testing.assert_array_equal(recorded[1][k], action)
    env.close(), ref.close()
