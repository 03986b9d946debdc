// Step kernels, float32 state, Poisson-type arrivals (Poisson, PoissonNonLinear, user arrival expressions share the layout).
#include "kernel_pick_f32.hpp"

namespace mbt_table {
StepKernel pick_step_poisson(int dyn, bool brownian, int reward_weight, bool norm, bool inject, int mode) {
  return pick_dyn<mbt::kArrPoisson, false>(dyn, brownian, reward_weight, norm, inject, mode);
}
}  // namespace mbt_table
