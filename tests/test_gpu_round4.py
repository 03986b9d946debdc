"""Round 4: the small-batch host path (one launch, completion flag), the launch gate, the specialised and streaming
instantiations of the contract tier, the per-configuration block of the bench line, BASELINE configs[4] sharded, and the tests
that become hardware evidence for the N > 1 path the moment a box shows two GPUs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from mbt_gym_amd import _native
from oracle.mbt_oracle import OracleConfig
from tests.env_factory import make_env

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(n, **kw):
    base = dict(num_trajectories=n, n_steps=40, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0, arrival="poisson",
                intensity=(140.0, 140.0), fill_exponent=1.5, dynamics="limit", reward="pnl", initial_inventory=0, max_inventory=50, seed=50,
                normalise_action_space=False, normalise_observation_space=False)
    base.update(kw)
    return OracleConfig(**base)


VARIANTS = [dict(), dict(normalise_action_space=True, normalise_observation_space=True),
            dict(arrival="hawkes", intensity=(20.0, 15.0), hawkes_jump=20.0, hawkes_speed=15.0, midprice="ou", ou_level=100.0, ou_speed=0.02, reward="running", phi=0.01, alpha=0.02),
            dict(dynamics="limit_and_market", market_half_spread=0.4, reward="cjmm", phi=0.01, alpha=0.05, initial_inventory=(-2, 3)),
            dict(dynamics="speed", arrival="none", impact="temp_perm", temporary_impact=0.02, permanent_impact=0.01, reward="cjoe", phi=0.01, alpha=0.05,
                 initial_inventory=12, max_inventory=1000, volatility=0.3),
            dict(precise_state=True), dict(precise_state=True, normalise_action_space=True, normalise_observation_space=True)]


def _action(rng, cfg, n):
    if cfg.dynamics == "speed":
        return rng.uniform(-0.2, 0.4, size=(n, 1)).astype(np.float32)
    if cfg.normalise_action_space:
        return rng.uniform(-1, 1, size=(n, 4 if cfg.dynamics == "limit_and_market" else 2)).astype(np.float32)
    depths = rng.uniform(0, 1.5, size=(n, 2))
    if cfg.dynamics == "limit_and_market":
        return np.concatenate([depths, rng.choice([0.0, 1.0], p=[0.9, 0.1], size=(n, 2))], axis=1).astype(np.float32)
    return depths.astype(np.float32)


@pytest.mark.parametrize("kw", VARIANTS)
@pytest.mark.parametrize("n", [1000, 37, 40000])
def test_small_batch_step_mirrors_exactly_what_the_device_holds(n, kw):
    """env.step() of a small batch is ONE launch: the kernel itself writes observation rows and rewards into host memory and
    raises a flag.  What comes back is what the device buffers hold (the DMA path's answer), in every tier and layout - and it is
    what an environment that takes the large-batch path (three DMA copies) returns for the same lanes."""
    kw = dict(kw)
    precise = kw.pop("precise_state", False)
    cfg = _cfg(n, **kw)
    fast = make_env(cfg, precise_state=precise)
    os.environ["MBT_HOST_FAST_PATH_LANES"] = "0"  # the same environment through the DMA path
    try:
        slow = make_env(cfg, precise_state=precise)
    finally:
        del os.environ["MBT_HOST_FAST_PATH_LANES"]
    rng = np.random.default_rng(3)
    np.testing.assert_array_equal(fast.reset(), slow.reset())
    for k in range(cfg.n_steps):
        action = _action(rng, cfg, n)
        o_f, r_f, d_f, _ = fast.step(action)
        o_s, r_s, d_s, _ = slow.step(action)
        np.testing.assert_array_equal(o_f, o_s, err_msg=f"step {k}: observation")
        np.testing.assert_array_equal(r_f, r_s, err_msg=f"step {k}: rewards")
        np.testing.assert_array_equal(d_f, d_s)
        np.testing.assert_array_equal(fast.observation_host(), o_f)  # the device-side buffers hold the same rows
    assert d_f.all()
    assert fast.episode_return_sums()[0] == slow.episode_return_sums()[0]
    fast.close(), slow.close()


def test_actions_handed_over_by_a_host_step_are_what_a_later_device_step_reads():
    """The small-batch path hands the actions over in the stage; a following step_device() (no pointer) / action repeat must find
    them in the library's action buffer (they are filed there on demand)."""
    cfg = _cfg(1000)
    a, b = make_env(cfg), make_env(cfg)
    a.reset(), b.reset()
    first, second = np.full((1000, 2), 0.4, np.float32), np.full((1000, 2), 0.9, np.float32)
    a.step(first)
    a.step(second)
    a.step_device()  # re-uses `second`
    a.step_repeat_device(3)
    b.set_action_host(first)
    b.step_device()
    b.set_action_host(second)
    for _ in range(5):
        b.step_device()
    np.testing.assert_array_equal(a.state, b.state)
    a.close(), b.close()


@pytest.mark.parametrize("burst", [1, 64, 4096])
def test_the_launch_gate_changes_the_schedule_and_nothing_else(burst):
    cfg = _cfg(1 << 16, n_steps=50)
    plain, gated = make_env(cfg), make_env(cfg)
    for env in (plain, gated):
        env.set_action_host(np.full((1 << 16, 2), 0.7, np.float32))
        env.reset_device()
    gated.set_launch_gate(burst)
    assert plain.step_many_device(330) == gated.step_many_device(330) == (330, 6)
    np.testing.assert_array_equal(plain.state, gated.state)
    logs = [[env.episode_log_pop() for _ in range(6)] for env in (plain, gated)]
    np.testing.assert_array_equal(np.array(logs[0]), np.array(logs[1]))
    gated.set_launch_gate(0)
    assert gated.step_many_device(7) == plain.step_many_device(7)
    np.testing.assert_array_equal(plain.state, gated.state)
    with pytest.raises(_native.NativeError):
        gated.set_launch_gate(1 << 20)
    plain.close(), gated.close()


@pytest.mark.parametrize("kw", [dict(), dict(reward="cjmm", phi=0.01, alpha=0.02, initial_inventory=(-2, 3)), dict(reward="running", phi=0.01, alpha=0.05),
                                dict(dynamics="limit_and_market", market_half_spread=0.3, initial_inventory=4),
                                dict(arrival="hawkes", intensity=(20.0, 15.0), hawkes_jump=20.0, hawkes_speed=15.0, midprice="ou", ou_level=100.0, ou_speed=0.02),
                                dict(dynamics="touch", market_half_spread=0.2, reward="running", phi=0.02, alpha=0.01)])
def test_specialised_and_general_instantiations_of_the_contract_tier_agree_to_the_bit(kw):
    """precise_state: raw spaces + PnL / exponent-2 penalties run specialised kernels (no pow / exp / normalisation code), injected
    noise runs the general one.  Fed the specialised kernel's own draws, the general kernel must reproduce its float64 state and
    its rewards bit for bit - the operations and their order are the same by construction."""
    n = 2048
    cfg = _cfg(n, **kw)
    special, general = make_env(cfg, precise_state=True), make_env(cfg, noise="injected", precise_state=True)
    rng = np.random.default_rng(9)
    np.testing.assert_array_equal(special.reset(), general.reset())
    for k in range(cfg.n_steps):
        if cfg.dynamics == "touch":
            action = rng.integers(0, 2, size=(n, 2)).astype(np.float32)
        else:
            action = _action(rng, cfg, n)
        general.set_noise(*_native.rng_fill(cfg.seed, 0, k, n))
        o_s, r_s, _, _ = special.step(action)
        o_g, r_g, _, _ = general.step(action)
        np.testing.assert_array_equal(r_s, r_g, err_msg=f"step {k}: rewards")
        np.testing.assert_array_equal(special.state64, general.state64, err_msg=f"step {k}: float64 state")
    special.close(), general.close()


def test_streaming_instantiation_of_the_contract_tier_is_bit_identical():
    """Beyond the Infinity Cache the contract tier now has non-temporal-load instantiations too (MBT_STREAM_LOADS forces the choice)."""
    cfg = _cfg(1 << 14, arrival="hawkes", intensity=(20.0, 15.0), hawkes_jump=20.0, hawkes_speed=15.0, midprice="ou", ou_level=100.0, ou_speed=0.02)
    envs = []
    for stream in ("0", "1"):
        os.environ["MBT_STREAM_LOADS"] = stream
        try:
            envs.append(make_env(cfg, precise_state=True))
        finally:
            del os.environ["MBT_STREAM_LOADS"]
    for env in envs:
        env.set_action_host(np.full((1 << 14, 2), 0.6, np.float32))
        env.reset_device()
        env.step_many_device(cfg.n_steps, auto_reset=False)
    np.testing.assert_array_equal(envs[0].state64, envs[1].state64)
    for env in envs:
        env.close()


def _bench(*args, timeout=900):
    env = dict(os.environ)
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(key, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-hbm-resident", "--no-rollout", "--no-device-loop", *args], capture_output=True, text=True,
                         timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_the_bench_line_measures_every_baseline_config_and_the_contract_tier():
    line = _bench("--steps", "200", "--warmup", "20")
    roof = line["roofline"]
    assert roof["frac"] == pytest.approx(min(roof["frac_events"], roof["frac_rocprof"] if roof["frac_rocprof"] is not None else 1.0))
    assert roof["achieved"] == pytest.approx(roof["frac"] * roof["peak"])
    rows = roof["configs"]
    assert len(rows) == 9 and not any("error" in row for row in rows), rows  # (round 5: cfg3 twice in the float32 tier - exact and float32 intensities)
    assert [row["credited_bytes_per_env_step"] for row in rows] == [44, 44, 60, 60, 52, 44, 44, 60, 52]
    assert [row["moved_bytes_per_env_step"] for row in rows] == [44, 44, 76, 60, 52, 60, 60, 92, 68]
    assert [row["tier"].split()[0].rstrip(",") for row in rows] == ["float32"] * 5 + ["precise_state"] * 4
    for row in rows:
        assert 0.05 < row["frac"] <= row["frac_events"] < 1.0 and row["avg_launch_us"] > 0.0
        assert row["frac_events"] == pytest.approx(row["credited_bytes_per_env_step"] * row["lanes"] / (row["avg_launch_us"] * 1e-6) / 8e12)
        assert row["kernel"].startswith("mbt::step_kernel<mbt::Variant<")


@pytest.mark.timeout(900)
def test_cfg4_sharded_block_of_the_multi_rank_line():
    """BASELINE.json configs[4] through the N > 1 code path: two gloo ranks on GPU 0, and RCCL with a world of one."""
    two = _bench("--gpus", "2", "--backend", "gloo", "--single-device", "--lanes", str(1 << 16), "--steps", "30", "--warmup", "5", "--prewarm-steps", "64",
                 "--cfg4-total-lanes", str(1 << 18), "--cfg4-steps", "60")
    one = _bench("--gpus", "1", "--force-distributed", "--lanes", str(1 << 16), "--steps", "30", "--warmup", "5", "--prewarm-steps", "64",
                 "--cfg4-total-lanes", str(1 << 18), "--cfg4-steps", "60")
    for line, ranks in ((two, 2), (one, 1)):
        block = line["cfg4_sharded"]
        assert block["scaling"] == "strong" and block["num_trajectories_total"] == 1 << 18 and block["num_trajectories_per_gpu"] == (1 << 18) // ranks
        assert block["credited_bytes_per_env_step"] == 52 and block["value"] > 0 and 0 < block["frac_per_gpu"] < 1
        assert block["avg_launch_us_fastest_rank"] <= block["avg_launch_us_slowest_rank"]
        assert line["roofline"]["avg_launch_us_per_rank"]["min"] <= line["roofline"]["avg_launch_us_per_rank"]["max"]


# ---- hardware evidence for N > 1: these run by themselves the moment a box shows two GPUs ---------------------------------
def _two_gpus():
    if _native.device_count() < 2:
        pytest.skip("needs two visible gfx950 devices")


@pytest.mark.timeout(1200)
def test_two_rccl_ranks_on_two_devices_reproduce_one_rank():
    """`python bench.py --gpus 2` as the driver launches it: two ranks, two devices, the C-ABI RCCL communicator over xGMI.  RCCL
    itself must report two ranks, the known-answer all-reduce must hold, and the mean episode return must equal that of ONE rank
    stepping all 2^21 lanes to 1e-12 (Philox keyed on global lane ids; the 24-byte all-reduce is the only exchange)."""
    _two_gpus()
    common = ("--steps", "1100", "--warmup", "0", "--prewarm-steps", "0", "--cfg4-total-lanes", str(1 << 20), "--cfg4-steps", "50")
    two = _bench("--gpus", "2", *common)
    one = _bench("--gpus", "1", "--lanes", str(1 << 21), "--no-configs", *common)
    assert two["n_gpus"] == 2 and two["config"]["rccl_ranks_seen"] == 2
    assert two["config"]["return_allreduce"].startswith("RCCL via mbt_env_set_communicator")
    assert two["collective"]["known_answer_ok"] is True
    assert two["mean_episode_return"] == pytest.approx(one["mean_episode_return"], rel=1e-12)
    assert two["cfg4_sharded"]["num_trajectories_per_gpu"] == 1 << 19


def test_one_process_two_devices_is_bit_equal_to_one_device():
    _two_gpus()
    from tests.test_gpu_multi_device import _cfg as md_cfg, _sharded

    cfg = md_cfg(4096)
    single, multi = make_env(cfg), _sharded(cfg, [0, 1])
    assert len(multi.shards) == 2
    rng = np.random.default_rng(2)
    np.testing.assert_array_equal(multi.reset(), single.reset())
    for k in range(cfg.n_steps):
        action = np.concatenate([rng.uniform(0, 1.5, size=(4096, 2)), rng.choice([0.0, 1.0], p=[0.9, 0.1], size=(4096, 2))], axis=1).astype(np.float32)
        o_s, r_s, d_s, _ = single.step(action)
        o_m, r_m, d_m, _ = multi.step(action)
        np.testing.assert_array_equal(o_m, o_s, err_msg=f"step {k}")
        np.testing.assert_array_equal(r_m, r_s)
    np.testing.assert_array_equal(multi.state, single.state)
    single.close(), multi.close()


def test_released_buffers_come_back_on_demand():
    """pinned output pools and the library's HBM staging of recorded rollouts are returned by close() - and by
    release_host_buffers() on a live environment, after which everything is re-created on demand and nothing changes in what
    the environment computes."""
    from mbt_gym_amd.agents.BaselineAgents import FixedActionAgent

    cfg = _cfg(3000, n_steps=30)
    a, b = make_env(cfg), make_env(cfg)
    agent_a, agent_b = FixedActionAgent(np.array([0.5, 0.6], np.float32), a), FixedActionAgent(np.array([0.5, 0.6], np.float32), b)
    a.reset(), b.reset()
    first_a, first_b = a.rollout(agent_a), b.rollout(agent_b)
    kept = first_a[0]  # an array the caller still holds must survive the release with its values
    a.release_host_buffers()
    assert "_pools" not in a.__dict__ and "_trajectory_pools" not in a.__dict__
    np.testing.assert_array_equal(kept, first_b[0])
    a.reset(), b.reset()
    for x, y in zip(a.rollout(agent_a)[:3], b.rollout(agent_b)[:3]):
        np.testing.assert_array_equal(x, y)
    obs_a = a.reset()
    np.testing.assert_array_equal(obs_a, b.reset())
    a.close(), b.close()
    assert "_pools" not in a.__dict__
