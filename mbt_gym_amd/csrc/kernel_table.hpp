// Which instantiation of the step / rollout kernels serves a configuration - the HOST-side table.
//
// The library instantiates several hundred kernels (what changes the memory layout, the amount of noise or the weight of the
// arithmetic is a template parameter, step_kernel.hpp: Variant).  They are compiled in SEPARATE translation units
// (csrc/kernels_*.hip, built in parallel by build.py) - each defines a few `pick_*` functions that return the address of
// the right instantiation and is the only place where those instantiations are named; mbt_env.hip (the C ABI) holds the
// configuration logic and never names a kernel template itself.  A kernel is launched through the pointer a pick function
// returned: its host stub and its code object live in the translation unit that instantiated it, so no relocatable device
// code is needed.  Every kernel TU defines MBT_KERNEL_TU before including the kernel headers, which leaves out the
// non-template helper kernels (reset, reductions, ...): those are defined once, in mbt_env.hip.
#pragma once
#include "../../include/mbt_env.h"
#include "speed_kernel.hpp"
#include "step_kernel.hpp"

namespace mbt_table {

using StepKernel = void (*)(const mbt::StepBuffers, const mbt::StepParams);
using RolloutKernel = void (*)(const mbt::StepBuffers, const mbt::StepParams, const mbt::RolloutParams);
using LearnedRolloutKernel = void (*)(const mbt::StepBuffers, const mbt::StepParams, const mbt::RolloutParams, const mbt::LearnedPolicyParams);
using ResidentKernel = void (*)(const mbt::StepBuffers, const mbt::StepParams, const mbt::ResidentParams);
// the graph-capturable step (step_kernel.hpp: captured_step_kernel) takes a third argument; the pick functions hand its address out
// under the two-argument type (as_step_kernel) and mbt_env.hip turns it back (as_captured_kernel) before the launch - a round trip
// between function pointer types, never a call through the wrong one
using CapturedKernel = void (*)(const mbt::StepBuffers, const mbt::StepParams, const mbt::CapturedParams);

// Which instantiation of a production-noise step kernel: default-policy loads | non-temporal loads (launches beyond the
// Infinity Cache, tune_for_size) | the small-batch host-API kernel that mirrors its outputs into host memory and raises a
// completion flag (mbt_env_step_host; step_kernel.hpp: signal_host).  Injected-noise kernels (parity mode) exist in the
// first form only: asked for a mirror they answer nullptr and the host path takes its two-launch fallback.
// kCaptured / kCapturedStream: the graph-capturable instantiations (the clock on the device, mbt_env_step_device_captured) with
// default-policy / non-temporal loads; production noise only.
enum LoadMode : int { kPlain = 0, kStream = 1, kMirror = 2, kCaptured = 3, kCapturedStream = 4 };

inline StepKernel as_step_kernel(CapturedKernel k) { return reinterpret_cast<StepKernel>(k); }
inline CapturedKernel as_captured_kernel(StepKernel k) { return reinterpret_cast<CapturedKernel>(k); }

template <class V>
StepKernel pick_mode(int mode) {
  if (mode == kCaptured) return as_step_kernel(mbt::captured_step_kernel<V, false>);
  if (mode == kCapturedStream) return as_step_kernel(mbt::captured_step_kernel<V, true>);
  return mode == kStream ? mbt::step_kernel<V, true, false> : mode == kMirror ? mbt::step_kernel<V, false, true> : mbt::step_kernel<V, false, false>;
}
template <class V_INJECT>
StepKernel pick_injected(int mode) {
  return (mode == kMirror || mode == kCaptured || mode == kCapturedStream) ? nullptr : mbt::step_kernel<V_INJECT>;
}

// ---- THE mapping from a configuration's values to a kernel's shape (step_kernel.hpp: Variant<tags...>) ---------------------------
// The pick functions of csrc/kernels_*.hip branch over run-time values down to compile-time ones; this is where those become the list of
// named tags a kernel is instantiated (and shown by a profiler) with.  ARR: mbt::kArrPoisson / kArrHawkes, XL: Hawkes intensities held
// exactly; DYN: mbt::kDyn*; BM: plain Brownian midprice; REW: mbt::kReward*.
template <int ARR, bool XL, int DYN, bool BM, int REW, bool NORM, bool INJECT, bool EXO = false, bool PRECISE = false>
using OrderBookShape = mbt::shape::make<
    mbt::shape::when<ARR == mbt::kArrHawkes && !XL, mbt::shape::hawkes>, mbt::shape::when<ARR == mbt::kArrHawkes && XL, mbt::shape::hawkes_exact>,
    mbt::shape::when<DYN == mbt::kDynLimitAndMarket, mbt::shape::limit_and_market>, mbt::shape::when<DYN == mbt::kDynTouch, mbt::shape::touch>,
    mbt::shape::when<BM, mbt::shape::brownian>, mbt::shape::when<REW == mbt::kRewardPnl, mbt::shape::pnl>, mbt::shape::when<REW == mbt::kRewardQuadratic, mbt::shape::quadratic>,
    mbt::shape::when<NORM, mbt::shape::normalised>, mbt::shape::when<INJECT, mbt::shape::injected>, mbt::shape::when<EXO, mbt::shape::exogenous>,
    mbt::shape::when<PRECISE, mbt::shape::precise>>;
// ... and for trading-with-speed dynamics (speed_kernel.hpp: SpeedVariant<tags...>)
template <class List> struct speed_variant_of;
template <class... Tags> struct speed_variant_of<mbt::shape::tags<Tags...>> { using type = mbt::SpeedVariant<Tags...>; };
template <bool STATE, bool NORM, bool INJECT, bool PRECISE = false, bool POW = true, bool HOST_IMPACT = false>
using SpeedShape = typename speed_variant_of<typename mbt::shape::join_all<
    mbt::shape::when<STATE, mbt::shape::impact_state>, mbt::shape::when<NORM, mbt::shape::normalised>, mbt::shape::when<INJECT, mbt::shape::injected>,
    mbt::shape::when<PRECISE, mbt::shape::precise>, mbt::shape::when<POW, mbt::shape::powers>, mbt::shape::when<HOST_IMPACT, mbt::shape::host_impact>>::type>::type;

// ---- predicates on a configuration that both the table and the C ABI use ------------------------------------------------
// how heavy the reward is (Variant::REWARD)
inline int reward_weight(const mbt_config& c) {
  if (c.reward_kind == MBT_REW_PNL) return mbt::kRewardPnl;
  const bool quadratic = (c.reward_kind == MBT_REW_RUNNING_PENALTY || c.reward_kind == MBT_REW_CJ_MM) && c.inventory_exponent == 2.0;
  return quadratic ? mbt::kRewardQuadratic : mbt::kRewardGeneral;
}
inline bool host_impact(const mbt_config& c) { return c.impact_kind == MBT_IMPACT_HOST || c.impact_kind == MBT_IMPACT_HOST_STATE; }
inline bool impact_has_state(const mbt_config& c) { return c.impact_kind >= MBT_IMPACT_TEMPORARY_AND_PERMANENT && c.impact_kind != MBT_IMPACT_HOST; }
// does this speed-dynamics configuration raise anything to a power other than 1 (impact, IMP:55) or 2 (inventory penalty,
// RW:59-68), or use the exponential utility?  (No reference configuration does; the kernels without are a quarter the code.)
inline bool speed_powers(const mbt_config& c) {
  return (c.impact_kind == MBT_IMPACT_TEMPORARY_POWER && c.impact_exponent != 1.0) || (c.reward_kind != MBT_REW_PNL && c.inventory_exponent != 2.0) ||
         c.reward_kind == MBT_REW_EXP_UTILITY;
}
// ExogenousMmFillProbabilityModel: the general tier only (runtime midprice coefficients, every reward, runtime
// normalisation flags), 8 step (+ 4 mirror) + 4 rollout kernels per intensity tier.
inline bool exogenous_fill(const mbt_config& c) {
  return c.fill_kind == MBT_FILL_EXOGENOUS_MM && (c.dynamics_kind == MBT_DYN_LIMIT || c.dynamics_kind == MBT_DYN_LIMIT_AND_MARKET);
}
// Hawkes intensities held exactly in the float32 tier (Variant::EXACT_LAM): the default; mbt_config::hawkes_float32_intensities opts out
inline bool exact_intensities(const mbt_config& c) {
  return c.arrival_kind == MBT_ARR_HAWKES && c.dynamics_kind != MBT_DYN_SPEED && !c.precise_state && !c.hawkes_float32_intensities;
}

// ---- the table: one function per translation unit family -----------------------------------------------------------------
// Limit-order-book family, float32 state: arrivals {Poisson, Hawkes with float32 intensities, Hawkes with exact intensities}
// x dynamics {limit, limit + market, touch} x {Brownian, other midprice} x reward weight {PnL, quadratic inventory penalties,
// general} x normalised x noise; WHICH other midprice and reward are runtime parameters inside them.
StepKernel pick_step_poisson(int dyn, bool brownian, int reward_weight, bool norm, bool inject, int mode);         // kernels_step_poisson.hip
StepKernel pick_step_hawkes(int dyn, bool brownian, int reward_weight, bool norm, bool inject, int mode);          // kernels_step_hawkes.hip
StepKernel pick_step_hawkes_exact(int dyn, bool brownian, int reward_weight, bool norm, bool inject, int mode);    // kernels_step_hawkes_exact.hip
// the exogenous-depth fill model (general tier): arrivals 0 Poisson | 1 Hawkes float32 | 2 Hawkes exact intensities
StepKernel pick_step_exogenous(int arrivals, bool market, bool inject, int mode);                                   // kernels_misc.hip
RolloutKernel pick_rollout_exogenous(int arrivals, bool market);                                                    // kernels_misc.hip
// precise_state (the reference's float64 state): special_reward = kRewardPnl / kRewardQuadratic when the specialised
// instantiation applies (production noise, raw spaces, no exogenous fill model), kRewardGeneral otherwise
StepKernel pick_step_precise(bool hawkes, int dyn, bool exo, int special_reward, bool brownian, bool inject, int mode);  // kernels_precise.hip
RolloutKernel pick_rollout_precise(bool hawkes, int dyn, bool exo);                                                 // kernels_precise.hip
// trading-with-speed dynamics
StepKernel pick_step_speed(const mbt_config& c, int mode);                                                          // kernels_speed.hip
RolloutKernel pick_rollout_speed(const mbt_config& c);                                                              // kernels_speed.hip
// fused rollouts, float32 state: arrivals as for pick_step_exogenous
RolloutKernel pick_rollout_order_book(int arrivals, int dyn, bool brownian, int reward_weight, bool norm);          // kernels_rollout.hip
// the resident small-batch step (opt-in): float32 tier, production noise; nullptr where the family has no resident form
ResidentKernel pick_resident(int arrivals, int dyn, bool brownian, int reward_weight, bool norm);                   // kernels_resident.hip
// a linear / MLP policy evaluated in the kernel (policy_mlp.hpp)
LearnedRolloutKernel pick_rollout_learned(int arrivals, bool market, bool brownian_pnl);                            // kernels_misc.hip

}  // namespace mbt_table
