// precise_state: the general tier (every midprice model, every reward, runtime normalisation flags) on the reference's
// float64 state: {Poisson-type, Hawkes} x {limit, limit + market, touch} + the exogenous-depth fill model on {limit, limit +
// market}, x noise = 20 step + 10 rollout kernels.  Round 4: like the float32 tier, the contract tier has SPECIALISED
// instantiations for what the BASELINE configurations run - {Brownian, other built-in midprice} x {plain PnL, the penalised
// rewards with exponent 2} x raw spaces (no pow / exp / normalisation code in the instruction stream: reward_exact<TIER>) -
// and STREAM / MIRROR instantiations of every production-noise kernel; same operations in the same order, so which one runs
// changes no bit.
#define MBT_KERNEL_TU 1
#include "kernel_table.hpp"

namespace mbt_table {
namespace {
template <int ARR, int DYN, bool EXO>
StepKernel pick_precise(bool inject, int mode) {
  if (inject) return pick_injected<OrderBookShape<ARR, false, DYN, false, mbt::kRewardGeneral, true, true, EXO, true>>(mode);
  return pick_mode<OrderBookShape<ARR, false, DYN, false, mbt::kRewardGeneral, true, false, EXO, true>>(mode);
}
template <int ARR, int DYN, int REW>
StepKernel pick_precise_special(bool brownian, int mode) {
  using B = OrderBookShape<ARR, false, DYN, true, REW, false, false, false, true>;   // Brownian midprice (BASELINE configs 1, 2, 4)
  using G = OrderBookShape<ARR, false, DYN, false, REW, false, false, false, true>;  // any other built-in midprice (config 3: OU)
  return brownian ? pick_mode<B>(mode) : pick_mode<G>(mode);
}
template <int ARR, int DYN>
StepKernel pick_precise_tier(int special_reward, bool brownian, bool inject, int mode) {
  if (special_reward == mbt::kRewardPnl) return pick_precise_special<ARR, DYN, mbt::kRewardPnl>(brownian, mode);
  if (special_reward == mbt::kRewardQuadratic) return pick_precise_special<ARR, DYN, mbt::kRewardQuadratic>(brownian, mode);
  return pick_precise<ARR, DYN, false>(inject, mode);
}
template <int ARR>
StepKernel pick_precise_dyn(int dyn, bool exo, int special_reward, bool brownian, bool inject, int mode) {
  switch (dyn) {
    case MBT_DYN_LIMIT: return exo ? pick_precise<ARR, mbt::kDynLimit, true>(inject, mode) : pick_precise_tier<ARR, mbt::kDynLimit>(special_reward, brownian, inject, mode);
    case MBT_DYN_LIMIT_AND_MARKET: return exo ? pick_precise<ARR, mbt::kDynLimitAndMarket, true>(inject, mode) : pick_precise_tier<ARR, mbt::kDynLimitAndMarket>(special_reward, brownian, inject, mode);
    default: return pick_precise_tier<ARR, mbt::kDynTouch>(special_reward, brownian, inject, mode);
  }
}
template <int ARR>
RolloutKernel rpick_precise(int dyn, bool exo) {
  switch (dyn) {
    case MBT_DYN_LIMIT:
      return exo ? mbt::rollout_kernel<OrderBookShape<ARR, false, mbt::kDynLimit, false, mbt::kRewardGeneral, true, false, true, true>>
                 : mbt::rollout_kernel<OrderBookShape<ARR, false, mbt::kDynLimit, false, mbt::kRewardGeneral, true, false, false, true>>;
    case MBT_DYN_LIMIT_AND_MARKET:
      return exo ? mbt::rollout_kernel<OrderBookShape<ARR, false, mbt::kDynLimitAndMarket, false, mbt::kRewardGeneral, true, false, true, true>>
                 : mbt::rollout_kernel<OrderBookShape<ARR, false, mbt::kDynLimitAndMarket, false, mbt::kRewardGeneral, true, false, false, true>>;
    default: return mbt::rollout_kernel<OrderBookShape<ARR, false, mbt::kDynTouch, false, mbt::kRewardGeneral, true, false, false, true>>;
  }
}
}  // namespace

StepKernel pick_step_precise(bool hawkes, int dyn, bool exo, int special_reward, bool brownian, bool inject, int mode) {
  return hawkes ? pick_precise_dyn<mbt::kArrHawkes>(dyn, exo, special_reward, brownian, inject, mode)
                : pick_precise_dyn<mbt::kArrPoisson>(dyn, exo, special_reward, brownian, inject, mode);
}
RolloutKernel pick_rollout_precise(bool hawkes, int dyn, bool exo) {
  return hawkes ? rpick_precise<mbt::kArrHawkes>(dyn, exo) : rpick_precise<mbt::kArrPoisson>(dyn, exo);
}
}  // namespace mbt_table
