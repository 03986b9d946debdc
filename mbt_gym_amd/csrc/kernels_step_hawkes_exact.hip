// Step kernels, float32 state, Hawkes arrivals with EXACT intensities (the default: Variant::EXACT_LAM, 76 B per env-step;
// arrivals are the float64 reference's on the same draws, ARR:110-123).
#include "kernel_pick_f32.hpp"

namespace mbt_table {
StepKernel pick_step_hawkes_exact(int dyn, bool brownian, int reward_weight, bool norm, bool inject, int mode) {
  return pick_dyn<mbt::kArrHawkes, true>(dyn, brownian, reward_weight, norm, inject, mode);
}
}  // namespace mbt_table
