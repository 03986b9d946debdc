"""The DEFAULT (float32-state) tier's reward guarantee as a formula - include/mbt_env.h states it, DESIGN.md section 4 derives it,
tests/test_gpu_random_configs.py asserts it.  No constant here is fitted to a soak: each is a count of float32 roundings or a
bound on a trade size, and the only measured quantities are the errors of the float32 STATE the step started from.

Notation: U = 2^-24, the relative error bound of one float32 rounding (half an ulp).  dS = midprice' - midprice, q' the
inventory after the step, r the reward; dS_err, c_err, q_err = |float32 state - reference state| BEFORE the step (what the
test measures on both environments; the tier's drift model bounds them, see `drift_bound`).

    |r_hip - r_ref|  <=  CONTRACT + COUPLING + CLIP

  CONTRACT  = 1e-5 + 1e-6 max(|r|, |q' dS|)   north_star's 1e-5, plus float32 itself: the rounding of an output >> 1 and of the
                                           product q' dS it is made of (16 U = 1e-6: a GBM move of 9 at q = 40 is a term of 360)
  COUPLING  = |q'| L (dS_err + 2 U |S|)  the increment's dependence on the float32 midprice: L = dS / dS_state =
                                           0       Brownian motion, jumps, constant (the increment does not read the state)
                                           theta   OU: the pull -theta (S - level) (MID:140-143; 2 U |S| = the rounding of `level`)
                                           |dS/S|  GBM: dS = S (mu dt + sigma sqrt(dt) z) (MID:95-103)
              (+ real-valued inventory, speed dynamics: q_err |dS| - the reward holds q' dS)
  CLIP      = only on lane-steps where the clip of TE:283-289 changed a value - there the reward carries the LEVEL of float32
              state, not just the step's increments:
                inventory clip:  |dq_clip| (dS_err + 2 U |S|)  [+ q_err |S| when the inventory is real-valued]
                cash clip:       c_err + |dq| (dS_err + 2 U |S|) + 4 U |c|
              with |dq|, |dq_clip| <= MAX_TRADE: one limit fill and one market order per side and step (MD:208-222): 2
              (speed dynamics: the step's volume |v dt|).
"""
import numpy as np

U = 2.0 ** -24
MAX_TRADE = 2.0  # units of inventory one order-book step can move (one limit fill + one market order on a side, MD:208-222)


def contract(r_ref, q_ds=0.0):
    return 1e-5 + 1e-6 * np.maximum(np.abs(r_ref), np.abs(q_ds))


def increment_sensitivity(cfg, s_prev_ref, s_next_ref):
    """L: how much of the float32 midprice error before the step shows in the step's increment dS."""
    if cfg.midprice in ("ou", "ou_jump"):
        return np.full_like(s_prev_ref, cfg.ou_speed)
    if cfg.midprice == "gbm":
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.nan_to_num(np.abs(s_next_ref - s_prev_ref) / np.abs(s_prev_ref), nan=0.0, posinf=0.0)
    return np.zeros_like(s_prev_ref)


def order_book_reward_bound(cfg, prev_hip, prev_ref, next_ref, r_ref, clipped):
    """Per-lane bound on |r_hip - r_ref| for one order-book step.  prev_*: RAW (N, D) states before the step of the device
    (float32) and of the reference (float64); next_ref: the reference's state after it; clipped: lanes where TE:283-289 fired."""
    s_err = np.abs(prev_hip[:, 3].astype(np.float64) - prev_ref[:, 3]) + 2 * U * np.abs(prev_ref[:, 3])
    c_err = np.abs(prev_hip[:, 0].astype(np.float64) - prev_ref[:, 0])
    q_next = np.maximum(np.abs(next_ref[:, 1]), np.abs(prev_ref[:, 1]))
    coupling = q_next * increment_sensitivity(cfg, prev_ref[:, 3], next_ref[:, 3]) * s_err
    clip = MAX_TRADE * s_err + (c_err + MAX_TRADE * s_err + 4 * U * np.maximum(np.abs(prev_ref[:, 0]), np.abs(next_ref[:, 0])))
    q_ds = next_ref[:, 1] * (next_ref[:, 3] - prev_ref[:, 3])
    return contract(r_ref, q_ds) + coupling + np.where(clipped, clip, 0.0)


def speed_clip_term(prev_hip, prev_ref, next_hip, next_ref, volume, price_if_undefined=0.0):
    """CLIP for trading-with-speed dynamics (MD:262-267), where the inventory is real-valued float32 state too and a step trades
    `volume` = |v dt| units: on a lane-step the clip touched, the inventory's own error is marked to market (q_err |S|) next to
    the levels of cash and midprice.  RAW (N, D) states before / after the step, device and reference.  (A constant midprice
    observed through a normalised, zero-width Box column reads NaN on both sides: its error is zero and its level is
    `price_if_undefined`.)"""
    fix = lambda state: np.where(np.isnan(state[:, 3]), price_if_undefined, state[:, 3])  # noqa: E731
    prev_hip, prev_ref, next_hip, next_ref = (np.column_stack([x[:, :3], fix(x), x[:, 4:]]) for x in (prev_hip, prev_ref, next_hip, next_ref))
    s_err = np.abs(prev_hip[:, 3] - prev_ref[:, 3]) + 2 * U * np.abs(prev_ref[:, 3])
    c_err = np.abs(prev_hip[:, 0] - prev_ref[:, 0])
    q_err = np.maximum(np.abs(prev_hip[:, 1] - prev_ref[:, 1]), np.abs(next_hip[:, 1] - next_ref[:, 1])) + 2 * U * np.abs(next_ref[:, 1])
    s_abs = np.maximum(np.abs(prev_ref[:, 3]), np.abs(next_ref[:, 3]))
    c_abs = np.maximum(np.abs(prev_ref[:, 0]), np.abs(next_ref[:, 0]))
    # the execution price is S + impact (MD:263-266) and a stateful impact model's y is float32 state as well (IMP:87-91, :130-135)
    y_err = (np.abs(prev_hip[:, 4] - prev_ref[:, 4]) + 2 * U * np.abs(prev_ref[:, 4])) if prev_ref.shape[1] > 4 else 0.0
    return (volume * (s_err + y_err) + q_err * s_abs) + (c_err + volume * (s_err + y_err) + 4 * U * c_abs)


def drift_bound(k, magnitude, roundings_per_step=2):
    """The tier's drift model for one float32 state column after k steps: every step rounds the column `roundings_per_step`
    times at a magnitude <= `magnitude`, independently and uniformly within half an ulp (variance ulp^2 / 12 each).  Six standard
    deviations of that sum: 6 sqrt(k roundings / 12) ulp32(magnitude) - exceeded by one lane in ~10^9 (a soak sees ~10^8
    lane-ends).  What the column additionally inherits from OTHER columns (cash from the midprice through -dq S) is added by the
    caller from the measured error of those columns."""
    ulp = np.spacing(np.float32(np.abs(magnitude))).astype(np.float64)
    return 6.0 * np.sqrt((k + 1) * roundings_per_step / 12.0) * ulp
