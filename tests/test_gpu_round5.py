"""Round 5: Hawkes intensities held exactly in the default tier (arrivals = the float64 reference's, 76 B per env-step), the
stale-stage fix of the small-batch host path, the kernel table split over translation units."""
import json
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest

from mbt_gym_amd import _native
from oracle.mbt_oracle import InjectedNoise, OracleConfig, OracleEnv
from tests.env_factory import make_env

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOAK = int(os.environ.get("MBT_HAWKES_SOAK", "0"))  # MBT_HAWKES_SOAK=k: k seeds x 2^17 lanes x 800 steps = k x 1.05e8 lane-steps (15 s of NumPy each)


def _cfg(n, **kw):
    base = dict(num_trajectories=n, n_steps=40, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0, arrival="poisson",
                intensity=(140.0, 140.0), fill_exponent=1.5, dynamics="limit", reward="pnl", initial_inventory=0, max_inventory=50, seed=50,
                normalise_action_space=False, normalise_observation_space=False)
    base.update(kw)
    return OracleConfig(**base)


HAWKES = dict(arrival="hawkes", intensity=(10.0, 10.0), hawkes_jump=40.0, hawkes_speed=60.0, midprice="ou", ou_level=100.0, ou_speed=0.01)  # BASELINE configs[3]


def test_hawkes_decisions_of_the_default_tier_are_the_float64_references_over_a_soak():
    """BASELINE configs[3] (Hawkes arrivals + OU midprice, ARR:89-93 defaults) in the DEFAULT tier, production noise, against the
    float64 oracle fed with the kernel's own draws: arrivals, fills and inventory array-equal on EVERY lane-step, the float64
    intensities (state64) equal to the oracle's - no lane retired, no window.  (Round 4: the float32 intensities of this tier
    decided ~7e-7 of lane-steps differently.)  2^15 lanes x 200 steps in the suite; MBT_HAWKES_SOAK=k: k seeds of 2^17 x 800 = 1.05e8 each."""
    for seed in range(50, 50 + max(SOAK, 1)):
        _hawkes_soak_of_one_seed(seed)


def _hawkes_soak_of_one_seed(seed):
    n, n_steps = (1 << 17, 800) if SOAK else (1 << 15, 200)
    cfg = _cfg(n, n_steps=n_steps, seed=seed, **HAWKES)
    env = make_env(cfg, noise="philox")
    env.record_events(True)
    draws = [_native.rng_fill(cfg.seed, 0, k, n) for k in range(n_steps)]
    oracle = OracleEnv(cfg, InjectedNoise(*[np.stack(x) for x in zip(*draws)]))
    del draws
    env.reset(), oracle.reset()
    rng = np.random.default_rng(seed - 45)
    arrivals_seen = 0
    for k in range(n_steps):
        action = rng.uniform(0.05, 1.2, size=(n, 2)).astype(np.float32)
        obs, rew, dones, _ = env.step(action)
        o_obs, o_rew, _ = oracle.step(action.astype(np.float64))
        np.testing.assert_array_equal(env.last_arrivals.astype(bool), oracle.last_arrivals.astype(bool), err_msg=f"step {k}: arrivals")
        np.testing.assert_array_equal(env.last_fills.astype(bool), oracle.last_fills.astype(bool), err_msg=f"step {k}: fills")
        np.testing.assert_array_equal(obs[:, 1].astype(np.float64), o_obs[:, 1], err_msg=f"step {k}: inventory")
        np.testing.assert_array_equal(obs[:, 4:6], o_obs[:, 4:6].astype(np.float32), err_msg=f"step {k}: observed intensities")
        if k % 20 == 0 or k == n_steps - 1:
            np.testing.assert_array_equal(env.state64[:, 4:6], oracle.state[:, 4:6], err_msg=f"step {k}: float64 intensities")
        assert np.max(np.abs(rew - o_rew)) <= 1e-5 + 1e-6 * np.max(np.abs(o_rew)) + 60.0 * cfg.ou_speed * 2e-4, f"step {k}: rewards"  # (OU coupling: |q| theta S_err)
        arrivals_seen += int(env.last_arrivals.sum())
    assert dones.all() and arrivals_seen > n  # (the process was alive: more than one arrival per lane)
    env.close()


@pytest.mark.parametrize("kw", [dict(), dict(dynamics="limit_and_market", market_half_spread=0.3, reward="cjmm", phi=0.01, alpha=0.02),
                                dict(dynamics="touch", market_half_spread=0.2, reward="running", phi=0.02, alpha=0.01, midprice="bm"),
                                dict(normalise_action_space=True, normalise_observation_space=True), dict(fill="exogenous", exo_depth=(0.25, 0.5), base_fill_probability=0.7,
                                                                                                       exo_depth_lo=(0.0, 0.1), exo_depth_hi=(0.6, 0.9))])
def test_exact_and_float32_intensity_tiers_agree_wherever_float32_can_tell(kw):
    """hawkes_float32_intensities=True is the same model with float32 intensity state: over a short episode at a few thousand lanes no
    draw falls into its ~1e-6 window, so inventories are equal and cash / midprice - which both tiers hold in float32 and advance with
    the same instructions - are bit-identical; the intensities differ by float32 rounding only.  Step loop and fused rollout."""
    from mbt_gym_amd.agents.BaselineAgents import FixedActionAgent

    n = 3000
    cfg = _cfg(n, **{**HAWKES, "hawkes_jump": 20.0, "hawkes_speed": 25.0, **kw})
    exact, f32 = make_env(cfg), make_env(cfg, hawkes_float32_intensities=True)
    rng = np.random.default_rng(11)
    o_e, o_f = exact.reset(), f32.reset()
    np.testing.assert_array_equal(o_e, o_f)
    lo = -1.0 if cfg.normalise_action_space else 0.0
    for k in range(cfg.n_steps):
        a = rng.uniform(lo, 1.0, size=(n, cfg.action_dim)).astype(np.float32)
        if cfg.dynamics == "touch":
            a = np.rint(np.abs(a)).astype(np.float32)
        (o_e, r_e, d_e, _), (o_f, r_f, d_f, _) = exact.step(a), f32.step(a)
        np.testing.assert_array_equal(o_e[:, :4], o_f[:, :4], err_msg=f"step {k}: cash / inventory / time / midprice")
        np.testing.assert_allclose(o_e[:, 4:6], o_f[:, 4:6], rtol=1e-6, atol=1e-4, err_msg=f"step {k}: intensities")
        np.testing.assert_array_equal(r_e, r_f, err_msg=f"step {k}: rewards")
    assert d_e.all() and d_f.all()
    # the fused rollout of the exact tier is its own step loop, bit for bit (remainders stay in registers there)
    fused, twin = make_env(cfg), make_env(cfg)  # (fresh: a reset does not rewind the Philox counters)
    fixed = np.full(cfg.action_dim, 1.0 if cfg.dynamics == "touch" else 0.3, dtype=np.float32)
    fused.reset(), twin.reset()
    obs_r, act_r, rew_r, steps, done = fused.rollout(FixedActionAgent(fixed, fused))
    for k in range(steps):
        obs, rew, dones, _ = twin.step(np.tile(fixed, (n, 1)))
        np.testing.assert_array_equal(obs, obs_r[k + 1], err_msg=f"rollout step {k}: observation")
        np.testing.assert_array_equal(rew, rew_r[k], err_msg=f"rollout step {k}: rewards")
    np.testing.assert_array_equal(fused.state64, twin.state64)
    for env in (exact, f32, fused, twin):
        env.close()


def test_exact_intensities_survive_reset_set_state_and_sharding():
    """The remainder buffer follows the life cycle: reset() writes what float32 left of the baselines (10.1 is not a float32), set_state()
    (float32 rows: no remainder) zeroes it, and a shard at a trajectory offset equals the same lanes of the whole environment."""
    n = 2048
    market = {**HAWKES, "intensity": (10.1, 14.3), "n_steps": 100}  # (60 x 0.01: inside the recursion's contraction domain)
    cfg = _cfg(n, **market)
    whole = make_env(cfg)
    whole.reset()
    np.testing.assert_array_equal(whole.state64[:, 4:6], np.tile([10.1, 14.3], (n, 1)))
    part = make_env(_cfg(1024, **market), trajectory_offset=1024)
    part.reset()
    a = np.full((n, 2), 0.35, np.float32)
    for _ in range(25):
        whole.step(a), part.step(a[:1024])
    np.testing.assert_array_equal(whole.state64[1024:], part.state64)
    rows = whole.state
    whole.set_state(rows)
    np.testing.assert_array_equal(whole.state64[:, 4:6], rows[:, 4:6].astype(np.float64))
    whole.close(), part.close()


def test_a_device_step_with_its_own_action_pointer_is_not_overwritten_by_an_earlier_host_step():
    """after a small-batch host step the newest actions sit in the host stage; a following step_device(ptr) /
    step_many_device(k, ptr) at N != n_pad copies the CALLER's actions into the library's buffer - and the stale stage must not be
    filed over them."""
    n = 1000
    cfg = _cfg(n)
    first, second = np.full((n, 2), 0.4, np.float32), np.full((n, 2), 0.9, np.float32)
    for many in (False, True):
        a, b, donor = make_env(cfg), make_env(cfg), make_env(cfg)
        a.reset(), b.reset(), donor.reset()
        donor.set_action_host(second)  # a device buffer holding `second` (n rows, no pad rows)
        ptr = donor.action_device.ptr
        a.step(first)  # host step: `first` sits in a's stage
        if many:
            assert a.step_many_device(3, ptr, auto_reset=False)[0] == 3
        else:
            for _ in range(3):
                a.step_device(ptr)
        b.set_action_host(first)
        b.step_device()
        b.set_action_host(second)
        for _ in range(3):
            b.step_device()
        np.testing.assert_array_equal(a.state, b.state, err_msg="step_many_device" if many else "step_device")
        for env in (a, b, donor):
            env.close()


def test_every_kernel_family_of_the_table_launches():
    """The step / rollout instantiations live in seven translation units (csrc/kernels_*.hip) and are launched through pointers the
    table hands out: one environment per unit, a step and - where the family has one - a fused rollout."""
    from mbt_gym_amd.agents.BaselineAgents import FixedActionAgent

    n = 1500
    hawkes = dict(HAWKES, hawkes_speed=25.0)
    families = [dict(), dict(hawkes), dict(hawkes, _float32=True), dict(fill="exogenous", exo_depth=(0.25, 0.5), base_fill_probability=0.7, exo_depth_lo=(0.0, 0.1), exo_depth_hi=(0.6, 0.9)),
                dict(_precise=True), dict(hawkes, _precise=True),
                dict(dynamics="speed", arrival="none", impact="temp_perm", temporary_impact=0.02, permanent_impact=0.01, reward="cjoe", phi=0.01, alpha=0.05,
                     initial_inventory=12, max_inventory=1000, volatility=0.3)]
    for kw in families:
        kw = dict(kw)
        env = make_env(_cfg(n, **{k: v for k, v in kw.items() if not k.startswith("_")}), precise_state=kw.get("_precise", False),
                       hawkes_float32_intensities=kw.get("_float32", False))
        obs = env.reset()
        a = np.full((n, env.action_dim), 0.3, np.float32)
        obs2, rew, dones, _ = env.step(a)
        assert np.isfinite(obs2).all() and np.isfinite(rew).all() and not np.array_equal(obs, obs2)
        env.reset()
        out = env.rollout(FixedActionAgent(a[0], env))
        assert out[3] == env.n_steps and out[4]
        env.close()


def _bench(*args, timeout=900):
    env = dict(os.environ)
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(key, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_the_bench_line_carries_the_fused_rollout():
    """the driver-run line itself says what the fused rollout does - returns only (env-steps/s; no HBM fraction, the
    vector-issue fraction from the committed counters) and recorded (written GB/s at 2^18 and 2^20 lanes against the write-only floor)."""
    line = _bench("--steps", "50", "--warmup", "5", "--no-cpu-baseline", "--no-hbm-resident", "--no-configs")
    block = line["rollout"]
    assert "error" not in block, block
    for name in ("returns_only_avellaneda_stoikov_policy", "returns_only_fixed_policy"):
        row = block[name]
        assert row["lanes"] == 1 << 20 and row["env_steps_per_s"] > 1e11 and row["hbm_bytes_per_env_step"] == 0 and "frac" not in row
    for log2n in (18, 20):
        row = block[f"recorded_avellaneda_stoikov_2^{log2n}"]
        assert "error" not in row, row
        assert row["written_bytes_per_env_step"] == 28 and row["lanes"] == 1 << log2n
        assert row["write_GBps"] == pytest.approx(28.0 * row["lanes"] / row["us_per_env_step_of_all_lanes"] * 1e-3)
        assert 0.3 < row["frac_of_8TBps"] < 1.0
        assert 30.0 < row["mean_return_of_the_recording"] < 100.0  # (the AS policy on this market earns ~64 per episode: the recording is a real trajectory)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_ranks_rehearsed_on_one_device(ranks):
    """the 8-GPU box will run `bench.py --gpus N --steps 20 --warmup 5` for N = 2, 4, 8 once, unattended.
    Rehearsed here with what one device allows - N self-spawned ranks sharing GPU 0, gloo as the transport (RCCL refuses two ranks on one
    device), the driver's arguments, BASELINE configs[4] sharded 2^24 lanes over the ranks: rendezvous, port, prewarm drift (every rank
    must finish the same number of episodes), memory of N cfg1 + N cfg4 shards, ONE JSON line - which carries a block PER RANK (device,
    PCI bus id, NUMA node, what the communicator spans, where the set-up seconds went, its own launch-to-launch time, the step kernel's
    rate at a size beyond the Infinity Cache on its device), so that a scaling curve that bends explains itself without a second run -
    and, for N = 8, the mean episode return equal to one rank stepping the same 2^23 global lanes to 1e-12 (Philox is keyed on global lane
    ids; the 24-byte all-reduce is the only exchange)."""
    import time

    common = ("--steps", "20", "--warmup", "5", "--prewarm-steps", "2048", "--no-cpu-baseline")
    t0 = time.time()
    line = _bench("--gpus", str(ranks), "--backend", "gloo", "--single-device", "--per-rank-hbm-lanes", str(1 << 22), *common)
    wall = time.time() - t0
    assert line["n_gpus"] == ranks and line["steps"] == 20 and line["warmup"] == 5 and line["scaling"] == "weak"
    assert line["config"]["num_trajectories_total"] == ranks << 20 and line["config"]["rccl_ranks_seen"] == ranks
    assert line["value"] == pytest.approx((ranks << 20) * 20 / (line["ms_per_step"] * 1e-3 * 20))
    assert line["collective"]["known_answer_ok"] is True
    span = line["collective"]["episodes_in_the_log_per_rank"]
    assert span["min"] == span["max"] == 2, span  # 2048 + 5 + 20 steps of a 1000-step episode, on every rank
    block = line["cfg4_sharded"]
    assert "error" not in block, block
    assert block["num_trajectories_total"] == 1 << 24 and block["num_trajectories_per_gpu"] == (1 << 24) // ranks and block["scaling"] == "strong"
    assert wall < 150.0, f"the {ranks}-rank line took {wall:.0f} s"
    per_rank = line["ranks"]
    assert [r["rank"] for r in per_rank] == list(range(ranks))
    for r in per_rank:
        assert r["device_ordinal"] == 0 and r["device_name"].startswith("gfx950") and r["pci_bus_id"].count(":") == 2 and isinstance(r["numa_node"], int)
        assert r["seconds"]["rendezvous_and_first_barrier"] > 0.0 and r["seconds"]["warm_up_steps_and_their_collectives"] > 0.0 and r["seconds"]["known_answer_collective"] > 0.0
        assert r["seconds"]["comm_init_rank"] is None and r["rccl_comm_count"] is None and "gloo" in r["return_allreduce"]  # (no RCCL communicator between ranks of one device)
        assert 1.0 < r["avg_launch_us"] < 500.0
        assert r["hbm_resident"]["lanes"] == 1 << 22 and 0.0 < r["hbm_resident"]["frac"] < 1.0, r["hbm_resident"]
    slowest = max(r["avg_launch_us"] for r in per_rank)
    assert slowest == pytest.approx(line["roofline"]["avg_launch_us_per_rank"]["max"], rel=1e-6)
    if ranks == 8:
        one = _bench("--gpus", "1", "--lanes", str(8 << 20), "--no-hbm-resident", "--no-configs", "--no-rollout", "--no-device-loop", *common)
        assert one["mean_episode_return"] == pytest.approx(line["mean_episode_return"], rel=1e-12)


@pytest.mark.parametrize("kw", [dict(), dict(normalise_action_space=True, normalise_observation_space=True), dict(hawkes=True, midprice="ou", ou_level=100.0, ou_speed=0.02, reward="running", phi=0.01, alpha=0.02),
                                dict(dynamics="limit_and_market", market_half_spread=0.4, reward="cjmm", phi=0.01, alpha=0.05, initial_inventory=(-2, 3))])
@pytest.mark.parametrize("n", [1000, 37, 4000])
def test_resident_small_batch_stepping_is_the_one_launch_path_bit_for_bit(n, kw, monkeypatch):
    """MBT_RESIDENT_STEP=1 (opt-in): env.step() of a small batch rings the doorbell of a kernel that stays on the device instead of
    launching one.  Same step_tile, same Philox counters, the clock advanced with the host's arithmetic: observations, rewards, dones,
    episode sums and the state afterwards equal the one-launch path's to the bit - over two episodes, with calls in between that make
    the kernel leave (reset, a device step, a state read, an idle pause longer than its time-out)."""
    import time

    kw = dict(kw)
    if kw.pop("hawkes", False):
        kw.update(arrival="hawkes", intensity=(20.0, 15.0), hawkes_jump=20.0, hawkes_speed=15.0)
    cfg = _cfg(n, **kw)
    plain = make_env(cfg)
    monkeypatch.setenv("MBT_RESIDENT_IDLE_US", "300")
    if n == 37:  # one size with mailbox and actions in pinned HOST memory (what a platform without a host-writable BAR gets)
        monkeypatch.setenv("MBT_RESIDENT_VRAM", "0")
    resident = make_env(cfg, resident_step=True)
    rng = np.random.default_rng(3)
    lo = -1.0 if cfg.normalise_action_space else 0.0
    for episode in range(2):
        np.testing.assert_array_equal(plain.reset(), resident.reset())
        for k in range(cfg.n_steps):
            a = rng.uniform(lo, 1.0, size=(n, cfg.action_dim)).astype(np.float32)
            if cfg.action_dim == 4:
                a[:, 2:] = rng.choice([lo, 1.0], p=[0.9, 0.1], size=(n, 2))
            (o_p, r_p, d_p, _), (o_r, r_r, d_r, _) = plain.step(a), resident.step(a)
            np.testing.assert_array_equal(o_r, o_p, err_msg=f"episode {episode} step {k}: observation")
            np.testing.assert_array_equal(r_r, r_p, err_msg=f"episode {episode} step {k}: rewards")
            np.testing.assert_array_equal(d_r, d_p)
            if d_p[0]:
                break
            if k == 5:
                np.testing.assert_array_equal(resident.state, plain.state)  # a state read: the kernel leaves, the next step starts another
            if k == 9:
                time.sleep(0.002)  # longer than the kernel's idle time-out: it leaves by itself, the next step notices and relaunches
            if k == 13:  # a device step in between (the actions of the last host step are filed from the resident stage)
                plain.step_device(), resident.step_device()
                np.testing.assert_array_equal(resident.state, plain.state)
        assert plain.clock == resident.clock
        np.testing.assert_array_equal(resident.state, plain.state)
        assert resident.episode_return_sums()[0] == plain.episode_return_sums()[0]
    plain.close(), resident.close()


def test_callback_inputs_are_not_overwritten_under_a_queued_device_step():
    """C ABI, small batch, a host-callback fill model: the probabilities live in one mapped block the step kernel reads in place.  After
    mbt_env_step_host the kernel has finished (its flag was waited for); after mbt_env_step_DEVICE nothing was waited for, so the next
    mbt_env_set_host_fill_probabilities must settle the stream before it overwrites the block.  Both routes, the same per-step
    probabilities: identical states."""
    import ctypes as C

    import torch

    from mbt_gym_amd.gym.ModelDynamics import LimitOrderModelDynamics
    from mbt_gym_amd.gym.TradingEnvironment import HostCallbackWarning, TradingEnvironment
    from mbt_gym_amd.stochastic_processes.arrival_models import PoissonArrivalModel
    from mbt_gym_amd.stochastic_processes.fill_probability_models import FillProbabilityModel
    from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel

    n, ns = 2048, 60

    class HostFill(FillProbabilityModel):
        def __init__(self):
            super().__init__(np.array([[]]), np.array([[]]), 1 / ns, 0.0, np.array([[]]), n, None)

        def _get_fill_probabilities(self, depths):
            return np.exp(-depths)

        max_depth = 4.0

        def update(self, arrivals, fills, actions, state=None):
            pass

    def build():
        md = LimitOrderModelDynamics(midprice_model=BrownianMotionMidpriceModel(volatility=2.0, initial_price=100, terminal_time=1.0, step_size=1 / ns, num_trajectories=n),
                                     arrival_model=PoissonArrivalModel(intensity=np.array([50.0, 50.0]), step_size=1 / ns, num_trajectories=n),
                                     fill_probability_model=HostFill(), num_trajectories=n)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", HostCallbackWarning)
            return TradingEnvironment(terminal_time=1.0, n_steps=ns, seed=7, max_inventory=20, num_trajectories=n, model_dynamics=md,
                                      normalise_action_space=False, normalise_observation_space=False)

    lib = _native.load_library()
    rng = np.random.default_rng(3)
    action = rng.uniform(0.1, 1.0, size=(n, 2)).astype(np.float32)
    probabilities = [np.ascontiguousarray(np.where(rng.uniform(size=(n, 2)) < 0.5, 1.0, 0.0)) for _ in range(ns - 1)]  # (0 / 1: every difference shows as a fill)
    dptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    done = C.c_int32(0)

    by_host, by_device = build(), build()
    by_host.reset(), by_device.reset()
    obs, rew = np.empty((n, 4), np.float32), np.empty((n,), np.float32)
    for p in probabilities:
        _native.check(lib.mbt_env_set_host_fill_probabilities(by_host._handle, dptr(p)))
        _native.check(lib.mbt_env_step_host(by_host._handle, action.ctypes.data, obs.ctypes.data, rew.ctypes.data, C.byref(done)))
    action_device = torch.as_tensor(action, device="cuda")
    torch.cuda.synchronize()
    for p in probabilities:
        _native.check(lib.mbt_env_set_host_fill_probabilities(by_device._handle, dptr(p)))
        _native.check(lib.mbt_env_step_device(by_device._handle, action_device.data_ptr(), C.byref(done)))
    np.testing.assert_array_equal(by_device.state64, by_host.state64)
    assert np.abs(by_host.state64[:, 1]).max() > 0  # (fills happened)
    by_host.close(), by_device.close()


def test_a_resident_kernel_that_does_not_answer_hands_the_environment_back_to_the_one_launch_path(monkeypatch, capfd):
    """The safety net of the resident mode: a step the kernel does not answer within MBT_RESIDENT_ANSWER_MS (200) is taken by a launch
    instead - the kernel takes a step whole or not at all - and the environment stays on that route (a latency optimisation, not a
    contract).  MBT_RESIDENT_ANSWER_MS=0 forces it at the first step: results equal the plain environment's to the bit, and the
    library says once what happened."""
    cfg = _cfg(1000, n_steps=30)
    plain = make_env(cfg)
    monkeypatch.setenv("MBT_RESIDENT_ANSWER_MS", "0")
    impatient = make_env(cfg, resident_step=True)
    rng = np.random.default_rng(9)
    np.testing.assert_array_equal(plain.reset(), impatient.reset())
    for k in range(cfg.n_steps):
        a = rng.uniform(0.0, 1.0, size=(1000, 2)).astype(np.float32)
        (o_p, r_p, d_p, _), (o_i, r_i, d_i, _) = plain.step(a), impatient.step(a)
        np.testing.assert_array_equal(o_i, o_p, err_msg=f"step {k}")
        np.testing.assert_array_equal(r_i, r_p, err_msg=f"step {k}")
        assert d_i[0] == d_p[0]
    np.testing.assert_array_equal(impatient.state, plain.state)
    assert impatient.clock == plain.clock and impatient.episode_return_sums()[0] == plain.episode_return_sums()[0]
    said = capfd.readouterr().err
    assert said.count("steps through one launch per step from here on") <= 1  # (once at most: the kernel may have answered the first doorbell within the first poll)
    plain.close(), impatient.close()


@pytest.mark.parametrize("resident", [False, True])
def test_environments_stepped_from_several_threads_at_once_equal_their_sequential_selves(resident):
    """The C ABI's threading contract: one host thread per environment handle, any number of handles at once (what
    MultiDeviceTradingEnvironment does with its shards).  Six threads step six small environments concurrently through env.step() - the
    one-launch path and the resident kernel, whose flag spins, mapped stages and signal-handler probe are per environment or behind a
    mutex - and each must reproduce, to the bit, what the same environment does alone."""
    from concurrent.futures import ThreadPoolExecutor

    threads, n, n_steps = 6, 700, 60

    def run(seed):
        cfg = _cfg(n, n_steps=n_steps, seed=seed)
        env = make_env(cfg, resident_step=resident)
        rng = np.random.default_rng(seed)
        out = [env.reset().copy()]
        for _ in range(n_steps):
            obs, rew, dones, _ = env.step(rng.uniform(0.0, 1.0, size=(n, 2)).astype(np.float32))
            out.append(obs.copy()), out.append(rew.copy())
        assert dones.all()
        total = env.episode_return_sums()[0]
        env.close()
        return out, total

    alone = [run(100 + k) for k in range(threads)]
    with ThreadPoolExecutor(max_workers=threads) as pool:
        together = list(pool.map(run, [100 + k for k in range(threads)]))
    for (a, ta), (b, tb) in zip(alone, together):
        assert ta == tb
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
