// Fused rollout kernels (step_kernel.hpp: rollout_kernel), float32 state, built-in order-book models: arrivals {Poisson,
// Hawkes float32 intensities, Hawkes exact intensities} x dynamics x {Brownian, other} x reward weight x normalised = 108.
#define MBT_KERNEL_TU 1
#include "kernel_table.hpp"

namespace mbt_table {
namespace {
template <int ARR, bool XL, int DYN, bool BM, int REW, bool NORM>
using V = OrderBookShape<ARR, XL, DYN, BM, REW, NORM, false>;

template <int ARR, bool XL, int DYN, bool BM>
RolloutKernel rpick_rew(int rew, bool norm) {
  switch (rew) {
    case mbt::kRewardPnl: return norm ? mbt::rollout_kernel<V<ARR, XL, DYN, BM, mbt::kRewardPnl, true>> : mbt::rollout_kernel<V<ARR, XL, DYN, BM, mbt::kRewardPnl, false>>;
    case mbt::kRewardQuadratic: return norm ? mbt::rollout_kernel<V<ARR, XL, DYN, BM, mbt::kRewardQuadratic, true>> : mbt::rollout_kernel<V<ARR, XL, DYN, BM, mbt::kRewardQuadratic, false>>;
    default: return norm ? mbt::rollout_kernel<V<ARR, XL, DYN, BM, mbt::kRewardGeneral, true>> : mbt::rollout_kernel<V<ARR, XL, DYN, BM, mbt::kRewardGeneral, false>>;
  }
}
template <int ARR, bool XL>
RolloutKernel rpick_dyn(int dyn, bool bm, int rew, bool norm) {
  switch (dyn) {
    case MBT_DYN_LIMIT: return bm ? rpick_rew<ARR, XL, mbt::kDynLimit, true>(rew, norm) : rpick_rew<ARR, XL, mbt::kDynLimit, false>(rew, norm);
    case MBT_DYN_LIMIT_AND_MARKET: return bm ? rpick_rew<ARR, XL, mbt::kDynLimitAndMarket, true>(rew, norm) : rpick_rew<ARR, XL, mbt::kDynLimitAndMarket, false>(rew, norm);
    default: return bm ? rpick_rew<ARR, XL, mbt::kDynTouch, true>(rew, norm) : rpick_rew<ARR, XL, mbt::kDynTouch, false>(rew, norm);
  }
}
}  // namespace

RolloutKernel pick_rollout_order_book(int arrivals, int dyn, bool brownian, int reward_weight, bool norm) {
  if (arrivals == 2) return rpick_dyn<mbt::kArrHawkes, true>(dyn, brownian, reward_weight, norm);
  if (arrivals == 1) return rpick_dyn<mbt::kArrHawkes, false>(dyn, brownian, reward_weight, norm);
  return rpick_dyn<mbt::kArrPoisson, false>(dyn, brownian, reward_weight, norm);
}
}  // namespace mbt_table
