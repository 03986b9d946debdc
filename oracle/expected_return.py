"""Exact expectation of the episode return of the DISCRETE-TIME environment under an inventory-time policy.
TEST INFRASTRUCTURE - NOT PRODUCT CODE.

For limit-order dynamics with Poisson arrivals, exponential fills and a Brownian midprice (the reference's
Avellaneda-Stoikov / Cartea-Jaimungal configurations) the inventory is a Markov chain on -Q..Q that does not depend
on the midprice, and every reward term has a closed-form conditional expectation given (step, inventory):
    P(bid trade) = min(1, lambda_b dt) * min(1, exp(-kappa d_b)) * [q <  Q]     (ARR:56, FILL:58, TE:323-327)
    P(ask trade) = min(1, lambda_a dt) * min(1, exp(-kappa d_a)) * [q > -Q]
    E[PnL | k, q]   = P_b d_b + P_a d_a + mu dt E[q']                           (RW:27-33 with MD:108-116, MID:60-65)
    running penalty = dt phi E[q'^2] (+ alpha E[q'^2] at the terminal step)     (RW:131-138)
    CjMm            = dt phi E[q'^2] + alpha (E[q'^2] - q^2 + dt/L q0^2)        (RW:99-109)
Propagating the inventory distribution forward gives the exact mean of the total reward - a known answer for the
whole step path (arrivals, fills, mask, dynamics, reward, policy plumbing) that a 2^20-lane run must hit within a
few standard errors.  The continuous-time closed form of CJP-2015 differs from it by the discretisation bias only.
"""
import numpy as np


def expected_episode_return(cfg, depth_fn, q0: int = 0):
    """cfg: oracle.mbt_oracle.OracleConfig (limit dynamics, Poisson arrivals, BM midprice).
    depth_fn(k, q_grid) -> (bid depths, ask depths) for observation time step k and inventories q_grid.
    Returns (expected total reward, inventory distribution at the end as a dict q -> probability)."""
    assert cfg.dynamics == "limit" and cfg.arrival == "poisson" and cfg.midprice == "bm"
    q_max = int(cfg.max_inventory)
    grid = np.arange(-q_max, q_max + 1)
    dt = cfg.step_size
    lam = np.minimum(np.asarray(cfg.intensity, dtype=np.float64) * dt, 1.0)
    k0 = int(round(cfg.start_time / dt))
    prob = np.zeros(grid.size)
    prob[q0 + q_max] = 1.0
    episode_length = cfg.terminal_time - k0 * dt
    total = 0.0
    for k in range(k0, cfg.n_steps):
        d_b, d_a = depth_fn(k, grid)
        p_b = lam[0] * np.minimum(np.exp(-cfg.fill_exponent * np.asarray(d_b, dtype=np.float64)), 1.0) * (grid < q_max)
        p_a = lam[1] * np.minimum(np.exp(-cfg.fill_exponent * np.asarray(d_a, dtype=np.float64)), 1.0) * (grid > -q_max)
        up, down, stay = p_b * (1 - p_a), p_a * (1 - p_b), 1 - p_b * (1 - p_a) - p_a * (1 - p_b)
        e_q = grid + p_b - p_a
        e_q2 = up * (grid + 1.0) ** 2 + down * (grid - 1.0) ** 2 + stay * grid.astype(np.float64) ** 2
        reward = p_b * d_b + p_a * d_a + cfg.drift * dt * e_q
        if cfg.reward == "running":
            reward = reward - dt * cfg.phi * e_q2 - (cfg.alpha * e_q2 if k == cfg.n_steps - 1 else 0.0)
        elif cfg.reward == "cjmm":
            reward = reward - dt * cfg.phi * e_q2 - cfg.alpha * (e_q2 - grid.astype(np.float64) ** 2 + dt / episode_length * q0**2)
        total += float(np.dot(prob, np.where(prob > 0, reward, 0.0)))
        nxt = prob * stay
        nxt[1:] += (prob * up)[:-1]
        nxt[:-1] += (prob * down)[1:]
        prob = nxt
    return total, dict(zip(grid.tolist(), prob.tolist()))
