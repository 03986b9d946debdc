"""BASELINE.json configs 3 and 4 at their full per-GPU sizes (2^22 lanes Hawkes + OU; 2^21 lanes limit + market
orders), checked through exact expectations of the discrete-time model - sharp at these sizes (standard errors of
1e-3 .. 1e-2) and independent of the oracle's speed:

  Hawkes (ARR:110-123):  E lambda_{k+1} = E lambda_k (1 - beta dt + eta dt) + beta lambda_0 dt   (arrival prob. lambda_k dt)
  OU     (MID:140-143):  E S_k - L = (1 - theta)^k (S_0 - L),   Var S_{k+1} = (1 - theta)^2 Var S_k + sigma^2 dt
  PnL with a fixed quote delta and no binding inventory limit:
      E sum R = sum_k (E lambda^b_k + E lambda^a_k) dt e^{-kappa delta} delta  +  E[q_{k+1}] E[dS_k]  (independent; zero here)
  market orders (MD:208-222): each costs the half spread h against the midprice: E sum R -= K (p_buy + p_sell) h,
      E q_K = q_0 + K (p_buy - p_sell)
"""
import numpy as np
import pytest

from oracle.mbt_oracle import OracleConfig
from tests.env_factory import make_env

pytestmark = pytest.mark.gpu


def _stats(total, total_sq, count):
    mean = total / count
    return mean, np.sqrt(max(total_sq / count - mean * mean, 0.0)) / np.sqrt(count)


def test_config3_hawkes_ou_at_2_to_22_lanes():
    n, steps, dt = 1 << 22, 200, 1e-3
    beta, eta, lam0, theta, level, s0, sigma, kappa, delta = 60.0, 40.0, (10.0, 14.0), 0.01, 100.0, 101.0, 2.0, 1.5, 0.7
    cfg = OracleConfig(num_trajectories=n, n_steps=1000, terminal_time=1.0, midprice="ou", ou_level=level, ou_speed=theta, volatility=sigma,
                       initial_price=s0, arrival="hawkes", intensity=lam0, hawkes_jump=eta, hawkes_speed=beta, fill_exponent=kappa,
                       dynamics="limit", reward="pnl", initial_inventory=0, max_inventory=10_000, seed=2024,
                       normalise_action_space=False, normalise_observation_space=False)
    env = make_env(cfg)
    env.track_lane_returns(True)
    env.reset()
    env.set_action_host(np.tile(np.array([[delta, delta]], np.float32), (n, 1)))
    lam = np.array(lam0, np.float64)
    mean_s, var_s, expected_return = s0 - level, 0.0, 0.0
    p_fill = np.exp(-kappa * np.float64(np.float32(delta)))
    e_q = 0.0
    for _ in range(steps):
        env.step_device()
        trades = lam * dt * p_fill  # per side, this step
        d_mid = -theta * mean_s  # E[S' - S]; the inventory after the step is independent of this step's noise
        e_q += trades[0] - trades[1]
        expected_return += trades.sum() * delta + e_q * d_mid
        lam = lam * (1 - beta * dt + eta * dt) + beta * np.array(lam0) * dt
        mean_s, var_s = (1 - theta) * mean_s, (1 - theta) ** 2 * var_s + sigma**2 * dt
    env.synchronize()
    st = env.state.astype(np.float64)
    assert st.shape == (n, 6) and env.clock[0] == pytest.approx(steps * dt, abs=1e-9) and env.clock[1] == steps
    se = np.sqrt(n)
    for side in range(2):  # Hawkes intensities: mean against the exact linear recursion
        col = st[:, 4 + side]
        assert col.mean() == pytest.approx(lam[side], abs=5 * col.std() / se), (side, col.mean(), lam[side])
        assert col.min() >= min(lam0) * 0.0  # never negative: baseline pull + positive jumps
    mid = st[:, 3]
    assert mid.mean() == pytest.approx(level + mean_s, abs=5 * np.sqrt(var_s) / se + 1e-4)
    assert mid.var() == pytest.approx(var_s, rel=5 * np.sqrt(2.0 / n) + 1e-4)
    # inventory: integer valued, mean = expected net flow (the asymmetric baselines make it non-zero)
    q = st[:, 1]
    assert np.all(q == np.rint(q))
    assert q.mean() == pytest.approx(e_q, abs=5 * q.std() / se)
    # the correlation of inventory with the OU pull is what the E[q] E[dS] term assumes away exactly: q_{k+1} depends
    # on arrivals/fills only, dS_k on (S_k, Z_k) only
    mean, stderr = _stats(*env.episode_return_sums())
    assert mean == pytest.approx(expected_return, abs=5 * stderr), (mean, expected_return, stderr)
    assert env.clip_count == 0
    env.close()


def test_config4_limit_and_market_orders_at_2_to_21_lanes():
    torch = pytest.importorskip("torch")
    n, steps, dt = 1 << 21, 100, 1e-3
    lam, kappa, delta, h, q0, p_buy, p_sell = 140.0, 1.5, 0.7, 0.5, 10, 0.02, 0.01
    cfg = OracleConfig(num_trajectories=n, n_steps=1000, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0,
                       arrival="poisson", intensity=(lam, lam), fill_exponent=kappa, dynamics="limit_and_market", market_half_spread=h,
                       reward="pnl", initial_inventory=q0, max_inventory=10_000, seed=77, normalise_action_space=False,
                       normalise_observation_space=False)
    env = make_env(cfg)
    env.track_lane_returns(True)
    env.set_stream(torch.cuda.current_stream().cuda_stream)
    env.reset()
    act = torch.as_tensor(env.action_device, device="cuda")  # the library's own action buffer, written in place
    assert act.shape == (n, 4)
    gen = torch.Generator(device="cuda").manual_seed(5)
    n_buy = n_sell = 0
    for _ in range(steps):
        u = torch.rand((n, 2), generator=gen, device="cuda")
        act[:, 0] = delta
        act[:, 1] = delta
        act[:, 2] = (u[:, 0] < p_buy).float()  # m_buy, m_sell ~ Bernoulli, thresholded at 0.5 by the dynamics (MD:202-206)
        act[:, 3] = (u[:, 1] < p_sell).float()
        n_buy += int(act[:, 2].sum().item())
        n_sell += int(act[:, 3].sum().item())
        env.step_device()
    torch.cuda.synchronize()
    st = env.state.astype(np.float64)
    q = st[:, 1]
    assert np.all(q == np.rint(q))
    # inventory: every market order moved it by exactly one unit; limit fills are symmetric
    se = np.sqrt(n)
    assert q.mean() == pytest.approx(q0 + (n_buy - n_sell) / n, abs=5 * q.std() / se)
    p_trade = np.float64(np.float32(lam * dt)) * np.exp(-kappa * np.float64(np.float32(delta)))
    expected = steps * 2 * p_trade * delta - (n_buy + n_sell) / n * h  # zero drift: inventory carries no expected PnL
    mean, stderr = _stats(*env.episode_return_sums())
    assert mean == pytest.approx(expected, abs=5 * stderr), (mean, expected, stderr)
    # cash + inventory * midprice telescopes to the summed rewards plus the initial mark-to-market value
    lane_total = st[:, 0] + q * st[:, 3] - q0 * 100.0
    assert lane_total.mean() == pytest.approx(mean, abs=2e-3)
    env.close()


def test_two_to_the_28_lanes_addressing():
    """The largest batch tried: 2^28 trajectories (state rows beyond the 4 GB mark of their buffer - 64-bit addressing in
    every kernel).  Seven back-to-back step launches equal the fused rollout of seven steps, row for row, on the device."""
    torch = pytest.importorskip("torch")
    from mbt_gym_amd.agents.BaselineAgents import FixedActionAgent

    n = 1 << 28
    cfg = OracleConfig(num_trajectories=n, n_steps=1000, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0, arrival="poisson",
                       intensity=(140.0, 140.0), fill_exponent=1.5, dynamics="limit", reward="pnl", initial_inventory=0, max_inventory=1000, seed=50,
                       normalise_action_space=False, normalise_observation_space=False)
    env_a, env_b = make_env(cfg), make_env(cfg)
    agent = FixedActionAgent(np.array([0.7, 0.7], np.float32), env_a)
    act = torch.as_tensor(env_b.action_device, device="cuda")
    act[:, 0] = 0.7
    act[:, 1] = 0.7
    torch.cuda.synchronize()
    env_a.reset_device()
    env_b.reset_device()
    steps, done = env_a.rollout_device(agent, max_steps=7)
    assert steps == 7 and not done
    for _ in range(7):
        env_b.step_device()
    env_a.synchronize()
    env_b.synchronize()
    state_a, state_b = torch.as_tensor(env_a.obs_device, device="cuda"), torch.as_tensor(env_b.obs_device, device="cuda")
    assert torch.equal(state_a, state_b)
    q = state_a[:, 1]
    assert float(q.abs().max()) <= 7 and bool(torch.all(q == torch.round(q)))
    assert float(state_a[-1, 2]) == pytest.approx(7e-3, abs=1e-6)  # the very last row was stepped too
    assert env_a.episode_return_sums()[0] == pytest.approx(env_b.episode_return_sums()[0], rel=1e-9)
    env_a.close()
    env_b.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("log2_lanes,kw", [
    (28, dict()),  # 16-byte rows: the state buffers are 4 GiB each - every byte offset past 2^32 is exercised
    (27, dict(arrival="hawkes", intensity=(10.0, 14.0), hawkes_jump=40.0, hawkes_speed=60.0, midprice="ou", ou_level=100.0, ou_speed=0.01)),  # 24-byte rows via LDS
])
def test_top_end_sizes_address_every_lane(log2_lanes, kw):
    """Maximum sizes: 2^28 lanes (Avellaneda-Stoikov; ~14 GB of device buffers) and 2^27 lanes with 24-byte rows.  The last
    tile of the big environment must be EXACTLY what a 1024-lane environment placed at that global offset computes (noise is a
    function of the global lane id; any 32-bit overflow in an index or byte offset would land somewhere else), the first
    tile what one at offset 0 computes, and the episode-return sums must count every lane."""
    torch = pytest.importorskip("torch")
    n, steps, tail = 1 << log2_lanes, 12, 1024
    base = dict(n_steps=1000, terminal_time=1.0, midprice="bm", volatility=2.0, initial_price=100.0, arrival="poisson", intensity=(140.0, 140.0),
                fill_exponent=1.5, dynamics="limit", reward="pnl", initial_inventory=0, max_inventory=1000, seed=91, normalise_action_space=False,
                normalise_observation_space=False)
    base.update(kw)
    big = make_env(OracleConfig(num_trajectories=n, **base))
    big.reset_device()
    torch.as_tensor(big.action_device, device="cuda").fill_(0.7)  # no 2 GB host array: the quote is written where it lives
    for _ in range(steps):
        big.step_device()
    big.synchronize()
    obs = torch.as_tensor(big.obs_device, device="cuda")
    rew = torch.as_tensor(big.reward_device, device="cuda")
    assert obs.shape == (n, big.observation_dim) and rew.shape == (n,)
    for offset in (0, n - tail, n // 2 + 3 * tail):
        small = make_env(OracleConfig(num_trajectories=tail, **base), trajectory_offset=offset)
        small.reset_device()
        small.set_action_host(np.full((tail, 2), 0.7, np.float32))
        for _ in range(steps):
            small.step_device()
        small.synchronize()
        np.testing.assert_array_equal(obs[offset:offset + tail].cpu().numpy(), small.observation_host(), err_msg=f"lanes from {offset}")
        np.testing.assert_array_equal(rew[offset:offset + tail].cpu().numpy(), torch.as_tensor(small.reward_device, device="cuda").cpu().numpy())
        small.close()
    total, _, count = big.episode_return_sums()
    assert count == n
    # lanes traded everywhere, not just below the first 2^32 bytes: the same share of non-zero inventories in both halves
    lower, upper = (float(torch.count_nonzero(half[:, 1]).item()) / (n // 2) for half in (obs[: n // 2], obs[n // 2:]))
    assert lower > 0.05 and upper == pytest.approx(lower, rel=5e-3)
    per_lane = total / n
    expected = steps * (140.0 + 140.0) * 1e-3 * np.exp(-1.5 * np.float64(np.float32(0.7))) * 0.7 if not kw else None
    if expected is not None:
        assert per_lane == pytest.approx(expected, rel=2e-3)  # E sum R of a fixed quote (test above); s.e. ~ 1e-5 at this size
    big.close()
