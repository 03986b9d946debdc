// The exogenous-depth fill model's kernels (general tier) and the fused rollouts with a learned policy (policy_mlp.hpp).
#define MBT_KERNEL_TU 1
#include "kernel_table.hpp"

namespace mbt_table {
namespace {
template <int ARR, bool XL, int DYN, bool INJECT>
using Exo = OrderBookShape<ARR, XL, DYN, false, mbt::kRewardGeneral, true, INJECT, true>;

template <int ARR, bool XL, int DYN>
StepKernel pick_exogenous(bool inject, int mode) {
  if (inject) return pick_injected<Exo<ARR, XL, DYN, true>>(mode);
  using V = Exo<ARR, XL, DYN, false>;
  if (mode == kCaptured || mode == kCapturedStream) return as_step_kernel(mbt::captured_step_kernel<V, false>);
  return mode == kMirror ? mbt::step_kernel<V, false, true> : mbt::step_kernel<V, false, false>;
}
template <int ARR, bool XL>
StepKernel pick_exogenous_dyn(bool market, bool inject, int mode) {
  return market ? pick_exogenous<ARR, XL, mbt::kDynLimitAndMarket>(inject, mode) : pick_exogenous<ARR, XL, mbt::kDynLimit>(inject, mode);
}
template <int ARR, bool XL>
RolloutKernel rpick_exogenous(bool market) {
  return market ? mbt::rollout_kernel<Exo<ARR, XL, mbt::kDynLimitAndMarket, false>> : mbt::rollout_kernel<Exo<ARR, XL, mbt::kDynLimit, false>>;
}

// Learned policies: two tiers x arrivals x {limit, limit + market}: Brownian midprice with plain PnL (the reference's default
// environment, BASELINE configs[1]: the environment part needs ~95 registers there) and the general tier (runtime midprice
// coefficients, every reward); both with run-time normalisation flags.
template <int ARR, bool XL, int DYN>
LearnedRolloutKernel pick_learned_tier(bool brownian_pnl) {
  using B = OrderBookShape<ARR, XL, DYN, true, mbt::kRewardPnl, true, false>;
  using G = OrderBookShape<ARR, XL, DYN, false, mbt::kRewardGeneral, true, false>;
  return brownian_pnl ? mbt::learned_rollout_kernel<B> : mbt::learned_rollout_kernel<G>;
}
template <int ARR, bool XL>
LearnedRolloutKernel pick_learned_dyn(bool market, bool brownian_pnl) {
  return market ? pick_learned_tier<ARR, XL, mbt::kDynLimitAndMarket>(brownian_pnl) : pick_learned_tier<ARR, XL, mbt::kDynLimit>(brownian_pnl);
}
}  // namespace

StepKernel pick_step_exogenous(int arrivals, bool market, bool inject, int mode) {
  if (arrivals == 2) return pick_exogenous_dyn<mbt::kArrHawkes, true>(market, inject, mode);
  if (arrivals == 1) return pick_exogenous_dyn<mbt::kArrHawkes, false>(market, inject, mode);
  return pick_exogenous_dyn<mbt::kArrPoisson, false>(market, inject, mode);
}
RolloutKernel pick_rollout_exogenous(int arrivals, bool market) {
  if (arrivals == 2) return rpick_exogenous<mbt::kArrHawkes, true>(market);
  if (arrivals == 1) return rpick_exogenous<mbt::kArrHawkes, false>(market);
  return rpick_exogenous<mbt::kArrPoisson, false>(market);
}
LearnedRolloutKernel pick_rollout_learned(int arrivals, bool market, bool brownian_pnl) {
  if (arrivals == 2) return pick_learned_dyn<mbt::kArrHawkes, true>(market, brownian_pnl);
  if (arrivals == 1) return pick_learned_dyn<mbt::kArrHawkes, false>(market, brownian_pnl);
  return pick_learned_dyn<mbt::kArrPoisson, false>(market, brownian_pnl);
}
}  // namespace mbt_table
