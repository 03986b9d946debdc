// The resident small-batch step kernels (step_kernel.hpp: resident_step_kernel; opt-in, MBT_RESIDENT_STEP=1): the float32 tier of
// the built-in order-book models with production noise - arrivals {Poisson, Hawkes with exact intensities} x dynamics x {Brownian,
// other midprice} x reward weight x normalised = 72 kernels.  Everything else (precise_state, speed dynamics, the exogenous-depth
// fill model, float32 Hawkes intensities, run-time compiled plugins, injected noise) keeps the one-launch-per-step path.
#define MBT_KERNEL_TU 1
#include "kernel_table.hpp"

namespace mbt_table {
namespace {
template <int ARR, bool XL, int DYN, bool BM, int REW, bool NORM>
using V = OrderBookShape<ARR, XL, DYN, BM, REW, NORM, false>;

template <int ARR, bool XL, int DYN, bool BM>
ResidentKernel pick_rew(int rew, bool norm) {
  switch (rew) {
    case mbt::kRewardPnl: return norm ? mbt::resident_step_kernel<V<ARR, XL, DYN, BM, mbt::kRewardPnl, true>> : mbt::resident_step_kernel<V<ARR, XL, DYN, BM, mbt::kRewardPnl, false>>;
    case mbt::kRewardQuadratic: return norm ? mbt::resident_step_kernel<V<ARR, XL, DYN, BM, mbt::kRewardQuadratic, true>> : mbt::resident_step_kernel<V<ARR, XL, DYN, BM, mbt::kRewardQuadratic, false>>;
    default: return norm ? mbt::resident_step_kernel<V<ARR, XL, DYN, BM, mbt::kRewardGeneral, true>> : mbt::resident_step_kernel<V<ARR, XL, DYN, BM, mbt::kRewardGeneral, false>>;
  }
}
template <int ARR, bool XL>
ResidentKernel pick_dyn(int dyn, bool bm, int rew, bool norm) {
  switch (dyn) {
    case MBT_DYN_LIMIT: return bm ? pick_rew<ARR, XL, mbt::kDynLimit, true>(rew, norm) : pick_rew<ARR, XL, mbt::kDynLimit, false>(rew, norm);
    case MBT_DYN_LIMIT_AND_MARKET: return bm ? pick_rew<ARR, XL, mbt::kDynLimitAndMarket, true>(rew, norm) : pick_rew<ARR, XL, mbt::kDynLimitAndMarket, false>(rew, norm);
    default: return bm ? pick_rew<ARR, XL, mbt::kDynTouch, true>(rew, norm) : pick_rew<ARR, XL, mbt::kDynTouch, false>(rew, norm);
  }
}
}  // namespace

ResidentKernel pick_resident(int arrivals, int dyn, bool brownian, int reward_weight, bool norm) {
  if (arrivals == 2) return pick_dyn<mbt::kArrHawkes, true>(dyn, brownian, reward_weight, norm);
  if (arrivals == 0) return pick_dyn<mbt::kArrPoisson, false>(dyn, brownian, reward_weight, norm);
  return nullptr;  // (float32 Hawkes intensities: the opt-out tier has no resident form)
}
}  // namespace mbt_table
