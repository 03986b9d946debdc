"""The reference's only automated tests (mbt_gym/rewards/tests/testRewardFunctions.py:33-135),
re-expressed against the oracle's reward restatement.  CPU only."""
import copy

import numpy as np

from oracle.mbt_oracle import cj_mm_criterion, pnl_reward, running_inventory_penalty

DT = 0.2
CUR = np.array([[120, 2, 0.5, 100]], dtype=float)
NXT = np.array([[20, 3, 0.5 + DT, 100.05]], dtype=float)  # a bid fill
PHI, ALPHA, T_END = 0.01, 1.0, 1.0
# columns: cash, inventory, time, midprice
EPISODE = [
    np.array([[100.0, 0, 0.0, 100]]),
    np.array([[0.5, 1, DT, 101]]),
    np.array([[102.0, 0, 2 * DT, 102]]),
    np.array([[103.0, 0, 3 * DT, 103]]),
    np.array([[206.5, -1, 4 * DT, 104]]),
    np.array([[103.0, 0, 5 * DT, 103]]),
]


def test_pnl_is_mark_to_market_difference():
    want = (NXT[:, 0] + NXT[:, 1] * NXT[:, 3]) - (CUR[:, 0] + CUR[:, 1] * CUR[:, 3])
    assert pnl_reward(CUR, NXT) == want


def test_running_inventory_penalty_per_step_value():
    want = pnl_reward(CUR, NXT) - PHI * DT * abs(NXT[:, 1]) ** 2
    got = running_inventory_penalty(CUR, NXT, False, PHI, ALPHA, 2.0)
    assert abs(want.item() - got.item()) < 5e-6


def _telescopes(states, start=0):
    q_init = states[start][:, 1]
    length = T_END - states[start][:, 2]
    cj, target = 0.0, 0.0
    for i in range(start, len(states) - 1):
        terminal = bool(states[i + 1][:, 2] == 1)
        cj += cj_mm_criterion(states[i], states[i + 1], PHI, ALPHA, 2.0, q_init, length).item()
        target += running_inventory_penalty(states[i], states[i + 1], terminal, PHI, ALPHA, 2.0).item()
    assert abs(cj - target) < 5e-6


def test_cjmm_sums_to_running_penalty_over_an_episode():
    _telescopes(EPISODE)


def test_cjmm_with_nonzero_initial_inventory():
    states = copy.deepcopy(EPISODE)
    states[0][:, 1] = 2
    states[0][:, 0] = -100
    states[-1] = copy.deepcopy(states[-2])
    states[-1][:, 2] = 1.0
    _telescopes(states)


def test_cjmm_from_a_partial_trajectory():
    _telescopes(EPISODE, start=2)
