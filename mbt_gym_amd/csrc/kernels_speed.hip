// Trading-with-speed dynamics (speed_kernel.hpp): every step and rollout instantiation.
// kPlain: rows of 20 bytes LOADED through LDS (`staged`: cache-resident sizes, only with an impact state); kStream:
// non-temporal direct loads (sizes beyond the Infinity Cache); kMirror: the small-batch host-API kernel (staged like kPlain).
// POW: the instantiation that can raise to arbitrary powers (speed_powers); every reference configuration runs without.
#define MBT_KERNEL_TU 1
#include "kernel_table.hpp"

namespace mbt_table {
namespace {
template <class V, class V_INJECT, bool STATE>
StepKernel pick_speed_mode(bool inject, int mode) {
  if (inject) return (mode == kMirror || mode == kCaptured || mode == kCapturedStream) ? nullptr : mbt::speed_step_kernel<V_INJECT>;
  if (mode == kCaptured || mode == kCapturedStream) {
    if constexpr (V::HOST_IMPACT) return nullptr;  // (the host is consulted every step)
    else return mode == kCaptured ? as_step_kernel(mbt::captured_speed_step_kernel<V, STATE>) : as_step_kernel(mbt::captured_speed_step_kernel<V, false, true>);
  }
  if (mode == kStream) return mbt::speed_step_kernel<V, false, true>;
  if (mode == kMirror) return mbt::speed_step_kernel<V, STATE, false, true>;
  return mbt::speed_step_kernel<V, STATE>;
}
template <bool STATE, bool POW>
StepKernel pick_speed_pow(bool norm, bool inject, int mode) {
  return norm ? pick_speed_mode<SpeedShape<STATE, true, false, false, POW>, SpeedShape<STATE, true, true, false, POW>, STATE>(inject, mode)
              : pick_speed_mode<SpeedShape<STATE, false, false, false, POW>, SpeedShape<STATE, false, true, false, POW>, STATE>(inject, mode);
}
template <bool STATE>
StepKernel pick_speed(bool powers, bool norm, bool inject, int mode) {
  return powers ? pick_speed_pow<STATE, true>(norm, inject, mode) : pick_speed_pow<STATE, false>(norm, inject, mode);
}
// (the precise_state tier of the speed family is the same kernel; POW as for the float32 tier)
template <bool STATE>
StepKernel pick_speed_precise(bool powers, bool inject, int mode) {
  return powers ? pick_speed_mode<SpeedShape<STATE, true, false, true, true>, SpeedShape<STATE, true, true, true, true>, STATE>(inject, mode)
                : pick_speed_mode<SpeedShape<STATE, true, false, true, false>, SpeedShape<STATE, true, true, true, false>, STATE>(inject, mode);
}
// (a host-callback price impact model: the precise_state kernels with SpeedVariant::HOST_IMPACT, general reward form)
template <bool STATE>
StepKernel pick_speed_host_impact(bool inject, int mode) {
  return pick_speed_mode<SpeedShape<STATE, true, false, true, true, true>, SpeedShape<STATE, true, true, true, true, true>, STATE>(inject, mode);
}
template <bool STATE, bool POW>
RolloutKernel pick_speed_rollout(bool norm) {
  return norm ? mbt::speed_rollout_kernel<SpeedShape<STATE, true, false, false, POW>> : mbt::speed_rollout_kernel<SpeedShape<STATE, false, false, false, POW>>;
}
}  // namespace

StepKernel pick_step_speed(const mbt_config& c, int mode) {
  const bool norm = c.normalise_action != 0 || c.normalise_observation != 0;
  const bool inject = c.noise_mode == MBT_NOISE_INJECTED;
  if (host_impact(c)) return impact_has_state(c) ? pick_speed_host_impact<true>(inject, mode) : pick_speed_host_impact<false>(inject, mode);
  if (c.precise_state) return impact_has_state(c) ? pick_speed_precise<true>(speed_powers(c), inject, mode) : pick_speed_precise<false>(speed_powers(c), inject, mode);
  return impact_has_state(c) ? pick_speed<true>(speed_powers(c), norm, inject, mode) : pick_speed<false>(speed_powers(c), norm, inject, mode);
}
RolloutKernel pick_rollout_speed(const mbt_config& c) {
  const bool norm = c.normalise_action != 0 || c.normalise_observation != 0;
  if (c.precise_state)
    return impact_has_state(c) ? mbt::speed_rollout_exact_kernel<SpeedShape<true, true, false, true>> : mbt::speed_rollout_exact_kernel<SpeedShape<false, true, false, true>>;
  if (impact_has_state(c)) return speed_powers(c) ? pick_speed_rollout<true, true>(norm) : pick_speed_rollout<true, false>(norm);
  return speed_powers(c) ? pick_speed_rollout<false, true>(norm) : pick_speed_rollout<false, false>(norm);
}
}  // namespace mbt_table
