// Step kernels, float32 state, Hawkes arrivals with FLOAT32 intensities (mbt_config::hawkes_float32_intensities = 1: 60 B per
// env-step; a draw within the float32 error of lambda dt can decide differently from the float64 reference).
#include "kernel_pick_f32.hpp"

namespace mbt_table {
StepKernel pick_step_hawkes(int dyn, bool brownian, int reward_weight, bool norm, bool inject, int mode) {
  return pick_dyn<mbt::kArrHawkes, false>(dyn, brownian, reward_weight, norm, inject, mode);
}
}  // namespace mbt_table
