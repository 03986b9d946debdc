"""Round-2 entry points of the C ABI on the GPU: k steps per host call with the episode log, the RCCL all-reduce
(world size 1 here - one GPU box; the N-rank path is the same code with N > 1), bench.py's self-spawned ranks, the
calibration of normalise_rewards against the reference, and the refusals that replaced silent misbehaviour."""
import json
import os
import subprocess
import time
import sys

import numpy as np
import pytest

from oracle.mbt_oracle import OracleConfig
from tests.env_factory import make_env

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(n, **kw):
    base = dict(num_trajectories=n, n_steps=40, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0, arrival="poisson",
                intensity=(140.0, 140.0), fill_exponent=1.5, dynamics="limit", reward="pnl", initial_inventory=0, max_inventory=40, seed=9,
                normalise_action_space=False, normalise_observation_space=False)
    base.update(kw)
    return OracleConfig(**base)


@pytest.mark.parametrize("kw", [dict(), dict(reward="cjmm", phi=0.01, alpha=0.02, initial_inventory=(-3, 4), max_inventory=6),
                                dict(arrival="hawkes", intensity=(10.0, 10.0), hawkes_speed=20.0, midprice="ou", ou_level=100.0, ou_speed=0.02)])
def test_step_many_equals_the_step_loop_and_logs_every_episode(kw):
    """mbt_env_step_many_device(k, auto_reset) == k x mbt_env_step_device with the consumer's own episode handling:
    bit-identical state, and the episode log holds each finished episode's sums in order."""
    n, k = 5000, 40 * 2 + 17
    cfg = _cfg(n, **kw)
    many, loop = make_env(cfg), make_env(cfg)
    action = np.tile(np.array([[0.6, 0.8]], np.float32), (n, 1))
    want = []
    for e in (many, loop):
        e.track_lane_returns(True)
        e.reset()
        e.set_action_host(action)
    for _ in range(k):
        if loop.step_device():
            want.append(loop.episode_return_sums())
            explicit_reset_like_auto_reset(loop)
    steps, episodes = many.step_many_device(k)
    assert (steps, episodes) == (k, 2)
    got = [many.episode_log_pop() for _ in range(2)]
    assert many.episode_log_pop() is None and many.episode_log_pop(wait=False) is None
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    np.testing.assert_array_equal(many.state, loop.state)
    np.testing.assert_array_equal(many.episode_return_sums(), loop.episode_return_sums())
    assert many.clock == loop.clock
    many.close(), loop.close()


def explicit_reset_like_auto_reset(env):
    """What auto_reset does, spelled out through the public ABI: the start time and the per-lane initial inventories of the
    last explicit reset (a fresh draw of random inventories is a HOST decision, TE:270-281, so auto_reset cannot make it)."""
    from mbt_gym_amd import _native

    _native.check(_native.load_library().mbt_env_reset(env._handle, env._get_start_time(), _native.fptr(env._q0_first)))


@pytest.fixture(autouse=True)
def _remember_first_q0(monkeypatch):
    """Record the per-lane initial inventories of every environment's explicit resets (host decision, TE:270-281)."""
    from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment

    original = TradingEnvironment._get_initial_inventories

    def remembering(self):
        q0 = original(self)
        self._q0_first = q0
        return q0

    monkeypatch.setattr(TradingEnvironment, "_get_initial_inventories", remembering)


def test_step_many_without_auto_reset_stops_at_the_end_of_the_episode():
    env = make_env(_cfg(2000))
    env.reset()
    env.set_action_host(np.tile(np.array([[0.7, 0.7]], np.float32), (2000, 1)))
    assert env.step_many_device(25, auto_reset=False) == (25, 0)
    assert env.step_many_device(100, auto_reset=False) == (15, 1)
    assert env.episode_log_pop() is None  # nothing is logged without auto_reset
    env.close()


def test_episode_log_ring_keeps_the_newest_sixteen():
    n = 1024
    env = make_env(_cfg(n, n_steps=5))
    env.reset()
    env.set_action_host(np.tile(np.array([[0.7, 0.7]], np.float32), (n, 1)))
    assert env.step_many_device(5 * 20) == (100, 20)
    popped = 0
    while env.episode_log_pop() is not None:
        popped += 1
    assert popped == 16
    env.close()


def test_rccl_allreduce_through_the_c_abi_world_size_one():
    """mbt_comm_* + mbt_env_allreduce_returns + mbt_env_set_communicator on a one-rank RCCL communicator: the symbols
    resolve (dlopen of the process's RCCL), the collective runs on the environment's stream, sums come back unchanged -
    including the NaN of an untracked second moment."""
    from mbt_gym_amd.distributed import RcclCommunicator

    n = 4096
    env = make_env(_cfg(n))
    comm = RcclCommunicator(rank=0, world_size=1, device=0)
    env.reset()
    env.set_action_host(np.tile(np.array([[0.7, 0.7]], np.float32), (n, 1)))
    env.step_many_device(10)
    local = env.episode_return_sums()
    assert np.isnan(local[1]) and local[2] == n
    np.testing.assert_array_equal(env.allreduce_return_sums(comm, local), local)
    np.testing.assert_array_equal(env.allreduce_return_sums(comm, [1.5, 2.5, 3.0]), [1.5, 2.5, 3.0])
    env.set_communicator(comm)
    twin = make_env(_cfg(n))
    twin.reset()
    twin.set_action_host(np.tile(np.array([[0.7, 0.7]], np.float32), (n, 1)))
    twin.step_many_device(10)
    assert env.step_many_device(70) == twin.step_many_device(70) == (70, 2)
    for _ in range(2):
        np.testing.assert_array_equal(env.episode_log_pop(), twin.episode_log_pop())
    env.set_communicator(None)
    env.close(), twin.close()
    comm.close()


def test_a_process_that_made_a_communicator_without_torch_exits_cleanly():
    """Regression: librccl used to be opened RTLD_GLOBAL; its librocm_smi64 then lent its amd::smi globals to the
    libamd_smi.so RCCL opens at initialisation, and the interpreter died in a double free AT EXIT (return code 134 after
    every test had passed) - unless torch had been imported first.  A fresh interpreter, no torch, must return 0."""
    code = ("from mbt_gym_amd.distributed import RcclCommunicator\n"
            "import sys\n"
            "c = RcclCommunicator(0, 1, 0)\n"
            "c.close()\n"
            "assert 'torch' not in sys.modules\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env=dict(os.environ, PYTHONPATH=ROOT))
    assert out.returncode == 0, (out.returncode, out.stderr[-2000:])


def test_the_rccl_probe_diagnostic_and_the_ranks_rccl_reports(tmp_path):
    """`bench.py --probe rank world gpu id-file` (a diagnostic, not part of the measured job): a world of one makes the C-ABI
    communicator, all-reduces a known answer on an environment's stream and reads ncclCommCount back."""
    from mbt_gym_amd.distributed import RcclCommunicator

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--probe", "0", "1", "0", str(tmp_path / "id")], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode == 0 and "ranks seen 1" in out.stderr, out.stderr[-2000:]
    comm = RcclCommunicator(rank=0, world_size=1, device=0)
    assert comm.count() == 1
    comm.close()


def test_the_watchdog_ends_a_process_whose_native_call_never_returns():
    """A collective (or communicator creation) that hangs cannot be cancelled from inside its process; bench.py's deadline
    says which step hung and exits with code 17 - the launcher then stops the other ranks - instead of hanging the run."""
    code = ("import sys, time\n"
            f"sys.path.insert(0, {ROOT!r})\n"
            "import bench\n"
            "with bench.Watchdog(0.3, 'a step that hangs', 3):\n"
            "    time.sleep(30)\n")
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 17 and "a step that hangs" in out.stderr and "rank 3" in out.stderr, (out.returncode, out.stderr[-500:])
    assert time.time() - t0 < 25
    with __import__("bench").Watchdog(5.0, "a step that returns", 0):
        pass  # cancelled on exit: nothing fires later


def _bench(*args):
    env = dict(os.environ)
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(key, None)
    # (the per-configuration block and the sharded cfg4 block have tests of their own: tests/test_gpu_round4.py)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-hbm-resident", "--no-configs", "--no-rollout", "--no-device-loop", "--cfg4-total-lanes", "0", *args],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_spawns_its_own_ranks_and_sharding_is_invisible_in_the_result():
    """`python bench.py --gpus 2` with no launcher: the script starts two ranks (here both on GPU 0, gloo transport - RCCL
    refuses two ranks on one device), each steps its shard of the trajectory axis, the 24-byte all-reduce merges the
    episode sums; the mean episode return equals that of ONE rank stepping all 2^21 lanes (Philox keyed on global ids)."""
    common = ("--steps", "1100", "--warmup", "0", "--prewarm-steps", "0")
    two = _bench("--gpus", "2", "--backend", "gloo", "--single-device", "--lanes", str(1 << 20), *common)
    one = _bench("--gpus", "1", "--lanes", str(1 << 21), *common)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["num_trajectories_total"] == one["config"]["num_trajectories_total"] == 1 << 21
    assert two["config"]["episodes_finished_in_timed_region"] == one["config"]["episodes_finished_in_timed_region"] == 1
    assert two["mean_episode_return"] == pytest.approx(one["mean_episode_return"], rel=1e-12)
    assert 60.0 < one["mean_episode_return"] < 75.0  # Avellaneda-Stoikov market, constant quote 0.7: about 67
    for line in (one, two):
        assert line["value"] == pytest.approx(line["config"]["num_trajectories_total"] * 1100 / (line["ms_per_step"] * 1e-3 * 1100), rel=1e-9)


@pytest.mark.timeout(600)
def test_bench_at_the_drivers_arguments_is_not_dominated_by_fixed_costs():
    """--steps 20 --warmup 5 (what the driver runs): the wall clock per step stays within 2x of the kernel's own
    launch-to-launch time (it was 9x when the timed region held event creation, two device syncs and cold clocks)."""
    line = _bench("--gpus", "1", "--steps", "20", "--warmup", "5")
    assert line["steps"] == 20 and line["warmup"] == 5
    assert line["ms_per_step"] * 1e3 <= 2.0 * line["roofline"]["avg_launch_us"], line
    assert line["roofline"]["avg_launch_us"] < 12.0
    assert line["config"]["prewarm_steps"] == 8192  # a count, not a time budget: the same on every rank


@pytest.mark.timeout(600)
def test_two_ranks_at_the_drivers_arguments_finish_the_same_number_of_episodes():
    """The driver's N > 1 line with everything at its default: both ranks take the same 8192 + 5 + 20 steps, so both end
    8 episodes and the 8 per-episode all-reduces pair up (with a time-based warm-up the ranks could drift apart by an
    episode, and the odd collective out would wait for ever)."""
    line = _bench("--gpus", "2", "--backend", "gloo", "--single-device", "--steps", "20", "--warmup", "5")
    assert line["n_gpus"] == 2 and line["config"]["prewarm_steps"] == 8192
    assert line["config"]["num_trajectories_total"] == 1 << 21
    assert line["config"]["episodes_finished_in_timed_region"] == 0
    # no episode ends inside the 20 timed steps: the line reports the last episode that finished during the warm-up (the
    # eighth), over both shards - the same number one rank stepping all 2^21 lanes for as long reports
    one = _bench("--gpus", "1", "--lanes", str(1 << 21), "--steps", "20", "--warmup", "5", "--prewarm-steps", "8192")
    assert line["mean_episode_return"] == pytest.approx(one["mean_episode_return"], rel=1e-12)
    assert 60.0 < line["mean_episode_return"] < 75.0  # a whole episode of the AS market at a constant quote of 0.7: about 67
    assert line["collective"]["episodes_all_reduced_before_the_timed_region"] == 8


@pytest.mark.timeout(600)
def test_the_multi_rank_code_path_of_bench_runs_on_rccl_with_a_world_of_one():
    """Everything `bench.py --gpus 8` does that a one-rank run does not - the nccl process group, the NUMA pinning, the C-ABI
    RCCL communicator made from an id broadcast over that group, the every-rank-or-none agreement, the per-episode
    all-reduce enqueued in-stream, ncclCommCount in the line, the known-answer all-reduce and its latency - forced onto the
    one GPU a test box has (`--force-distributed`: a world of one).  Nothing in the 8-rank path differs but the world size."""
    line = _bench("--gpus", "1", "--force-distributed", "--steps", "1100", "--warmup", "5", "--prewarm-steps", "64")
    assert line["config"]["return_allreduce"].startswith("RCCL via mbt_env_set_communicator"), line["config"]
    assert line["config"]["rccl_ranks_seen"] == 1
    assert line["config"]["episodes_finished_in_timed_region"] == 1  # one in-stream reduce + all-reduce + reset inside the timed region
    c = line["collective"]
    assert c["known_answer_ok"] is True and 0.0 < c["us"] < 5000.0 and c["per_episode_share_of_stepping"] < 0.2
    assert 60.0 < line["mean_episode_return"] < 75.0
    # the per-rank block of the multi-rank line (round 6), through the RCCL route: what RCCL says the communicator spans, the seconds
    # of its creation and of its FIRST collective, this rank's own launch-to-launch time, the kernel's HBM-resident rate on its device
    (only,) = line["ranks"]
    assert only["rank"] == 0 and only["rccl_comm_count"] == 1 and only["device_name"].startswith("gfx950")
    assert only["seconds"]["comm_init_rank"] > 0.0 and only["seconds"]["first_collective"] > 0.0 and only["seconds"]["rendezvous_and_first_barrier"] > 0.0
    assert only["avg_launch_us"] == pytest.approx(line["roofline"]["avg_launch_us_per_rank"]["max"], rel=1e-6)
    assert only["hbm_resident"]["lanes"] == 1 << 24 and 0.5 < only["hbm_resident"]["frac"] < 1.0, only["hbm_resident"]
    plain = _bench("--gpus", "1", "--steps", "1100", "--warmup", "5", "--prewarm-steps", "64")
    assert "collective" not in plain and plain["mean_episode_return"] == pytest.approx(line["mean_episode_return"], rel=1e-12)


@pytest.mark.timeout(600)
def test_two_gloo_ranks_report_the_collective_and_the_world_size():
    line = _bench("--gpus", "2", "--backend", "gloo", "--single-device", "--lanes", str(1 << 16), "--steps", "50", "--warmup", "5", "--prewarm-steps", "64")
    assert line["config"]["rccl_ranks_seen"] == 2 and line["config"]["return_allreduce"] == "torch.distributed/gloo"
    assert line["collective"]["known_answer_ok"] is True


def test_reward_scaling_matches_the_reference_calibration(repo_root):
    """normalise_rewards=True: 1 / (mean episode return of the fixed action 1/kappa), TE:329-343 - including the reference's
    behaviour that with normalise_action_space=True (the default) that fixed action is a NORMALISED action."""
    g = np.load(os.path.join(repo_root, "tests", "golden", "agents_reward_scaling.npz"))
    ns = int(g["n_steps"])
    for tag, norm in (("default", True), ("raw_actions", False)):
        cfg = OracleConfig(num_trajectories=64, n_steps=ns, terminal_time=1.0, midprice="bm", volatility=float(g["sigma"]), initial_price=100.0,
                           arrival="poisson", intensity=tuple(g["intensity"]), fill_exponent=float(g["kappa"]), dynamics="limit", reward="pnl",
                           initial_inventory=0, max_inventory=int(g["max_inventory"]), seed=7, normalise_action_space=norm,
                           normalise_observation_space=norm)
        env = make_env(cfg, normalise_rewards=True)
        # Monte-Carlo over 100 000 lanes on both sides (std of an episode return ~ 4): relative standard error ~ 3e-3
        assert env.reward_scaling == pytest.approx(float(g[tag]), rel=0.015), tag
        env.close()


def test_row_width_disagreements_and_stiff_hawkes_are_refused():
    from mbt_gym_amd._native import NativeError
    from mbt_gym_amd.gym.ModelDynamics import AtTheTouchModelDynamics
    from mbt_gym_amd.gym.TradingEnvironment import TradingEnvironment, UnsupportedOnDevice
    from mbt_gym_amd.stochastic_processes.arrival_models import PoissonArrivalModel
    from mbt_gym_amd.stochastic_processes.fill_probability_models import ExogenousMmFillProbabilityModel
    from mbt_gym_amd.stochastic_processes.midprice_models import BrownianMotionMidpriceModel
    from tests.env_factory import _FixedBoundsProcess

    n, dt = 64, 0.01
    # the host would count 6 columns (the fill model's two depths), the device lays at-the-touch out with 4: refused
    best = tuple(_FixedBoundsProcess(0.2, 0.0, 1.0, dt, n) for _ in range(2))
    md = AtTheTouchModelDynamics(
        midprice_model=BrownianMotionMidpriceModel(step_size=dt, num_trajectories=n),
        arrival_model=PoissonArrivalModel(intensity=np.array([50.0, 50.0]), step_size=dt, num_trajectories=n), num_trajectories=n)
    md.fill_probability_model = ExogenousMmFillProbabilityModel(best, step_size=dt, num_trajectories=n)
    with pytest.raises((UnsupportedOnDevice, NativeError)):
        TradingEnvironment(n_steps=100, model_dynamics=md, num_trajectories=n, normalise_action_space=False, normalise_observation_space=False)
    # Hawkes with mean_reversion_speed * step_size >= 1: refused unless explicitly allowed
    stiff = _cfg(n, arrival="hawkes", intensity=(10.0, 10.0), hawkes_speed=60.0)  # 60 * (1/40) = 1.5
    with pytest.raises(NativeError, match="Hawkes"):
        make_env(stiff)
    make_env(stiff, allow_stiff_hawkes=True).close()
    # ... and the step_size setter (TE:158-167) re-checks the same domain: 20 * 0.025 is fine, 20 * 0.06 is not
    soft = make_env(_cfg(n, arrival="hawkes", intensity=(10.0, 10.0), hawkes_speed=20.0))
    soft.step_size = 0.03
    with pytest.raises(NativeError, match="Hawkes"):
        soft.step_size = 0.06
    soft.close()
