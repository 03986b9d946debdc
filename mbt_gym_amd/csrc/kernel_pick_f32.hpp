// The float32-state order-book step kernels of ONE arrival family (kernels_step_poisson.hip, kernels_step_hawkes.hip,
// kernels_step_hawkes_exact.hip): dynamics {limit, limit + market, touch} x {Brownian, other midprice} x reward weight
// {PnL, quadratic inventory penalties, general} x normalised x noise {Philox: plain / stream / mirror; injected} = 144
// kernels per family.  XL: the Hawkes intensities are held exactly (Variant::EXACT_LAM).
#pragma once
#define MBT_KERNEL_TU 1
#include "kernel_table.hpp"

namespace mbt_table {

template <int ARR, bool XL, int DYN, bool BM, int REW, bool NORM, bool INJECT>
using OrderBookVariant = OrderBookShape<ARR, XL, DYN, BM, REW, NORM, INJECT>;

template <int ARR, bool XL, int DYN, bool BM, int REW, bool NORM>
StepKernel pick_noise(bool inject, int mode) {
  if (inject) return pick_injected<OrderBookVariant<ARR, XL, DYN, BM, REW, NORM, true>>(mode);
  return pick_mode<OrderBookVariant<ARR, XL, DYN, BM, REW, NORM, false>>(mode);
}
template <int ARR, bool XL, int DYN, bool BM, int REW>
StepKernel pick_flags(bool norm, bool inject, int mode) {
  return norm ? pick_noise<ARR, XL, DYN, BM, REW, true>(inject, mode) : pick_noise<ARR, XL, DYN, BM, REW, false>(inject, mode);
}
template <int ARR, bool XL, int DYN, bool BM>
StepKernel pick_rew(int rew, bool norm, bool inject, int mode) {
  switch (rew) {
    case mbt::kRewardPnl: return pick_flags<ARR, XL, DYN, BM, mbt::kRewardPnl>(norm, inject, mode);
    case mbt::kRewardQuadratic: return pick_flags<ARR, XL, DYN, BM, mbt::kRewardQuadratic>(norm, inject, mode);
    default: return pick_flags<ARR, XL, DYN, BM, mbt::kRewardGeneral>(norm, inject, mode);
  }
}
template <int ARR, bool XL, int DYN>
StepKernel pick_pen(bool bm, int rew, bool norm, bool inject, int mode) {
  return bm ? pick_rew<ARR, XL, DYN, true>(rew, norm, inject, mode) : pick_rew<ARR, XL, DYN, false>(rew, norm, inject, mode);
}
template <int ARR, bool XL>
StepKernel pick_dyn(int dyn, bool bm, int rew, bool norm, bool inject, int mode) {
  switch (dyn) {
    case MBT_DYN_LIMIT: return pick_pen<ARR, XL, mbt::kDynLimit>(bm, rew, norm, inject, mode);
    case MBT_DYN_LIMIT_AND_MARKET: return pick_pen<ARR, XL, mbt::kDynLimitAndMarket>(bm, rew, norm, inject, mode);
    default: return pick_pen<ARR, XL, mbt::kDynTouch>(bm, rew, norm, inject, mode);
  }
}

}  // namespace mbt_table
