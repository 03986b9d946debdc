// The fused rollout kernel's trajectory RECORDING under each store policy (GPU box only): the production rollout_kernel of the
// Avellaneda-Stoikov workload (BASELINE configs[1]) with the AS closed-form policy, compiled three times with
// -DMBT_RECORD_STORE_POLICY = 0 (plain write-back stores) | 1 (sc1: written through the L2) | 2 (nt) - `make` builds
// mb_rollout_p0 / _p1 / _p2 - at 2^18 and 2^20 lanes, next to the same launch WITHOUT a recording (the arithmetic alone) and to
// the write-only floor of tools/microbench/mb_floor.hip `record`.  28 B written per lane and step.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../mbt_gym_amd/csrc/step_kernel.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

using AS = mbt::Variant<mbt::shape::brownian, mbt::shape::pnl>;

int run(int lg, uint32_t steps) {
  const uint32_t n = 1u << lg, n_pairs = n / 2, blocks = n_pairs / mbt::kBlockThreads;
  float *s0, *s1, *act, *rew, *obs_t, *act_t, *rew_t; double* ws; unsigned long long* clip;
  CK(hipMalloc(&s0, size_t(n) * 16)); CK(hipMalloc(&s1, size_t(n) * 16)); CK(hipMalloc(&act, size_t(n) * 8)); CK(hipMalloc(&rew, size_t(n) * 4));
  CK(hipMalloc(&ws, blocks * 4 * 8)); CK(hipMalloc(&clip, 8 * mbt::kClipSlots));
  CK(hipMalloc(&obs_t, size_t(n) * 16 * (steps + 1))); CK(hipMalloc(&act_t, size_t(n) * 8 * steps)); CK(hipMalloc(&rew_t, size_t(n) * 4 * steps));
  CK(hipMemset(ws, 0, blocks * 32)); CK(hipMemset(clip, 0, 8 * mbt::kClipSlots)); CK(hipMemset(act, 0, size_t(n) * 8));
  std::vector<float> h(size_t(n) * 4);
  for (uint32_t i = 0; i < n; ++i) { h[4 * i] = 0; h[4 * i + 1] = 0; h[4 * i + 2] = 0; h[4 * i + 3] = 100.f; }
  CK(hipMemcpy(s0, h.data(), size_t(n) * 16, hipMemcpyHostToDevice));
  mbt::StepParams P{};
  P.n = n; P.n_pairs = n_pairs; P.key0 = 50; P.dt = 1e-3f; P.vol_sqrt_dt = 2.f * sqrtf(1e-3f);
  P.arr_thr_bid = P.arr_thr_ask = 0.14f; P.arr_thr_w_bid = P.arr_thr_w_ask = uint32_t(0.14 * 16777216.0) << 8;
  P.kappa_log2e_neg = -1.5f * 1.4426950408889634f; P.kappa_f64 = 1.5; P.fill_depth_per_log2 = float(-0.6931471805599453 / 1.5); P.fill_band_abs = float(2e-7 / 1.5);
  P.q_max = 1000.f; P.c_max = 1e8f; P.reward_scale = 1.f; P.exponent_is_two = 1; P.exponent = 2.f; P.mid_add = 1.f; P.arr_dt = 1e-3f; P.arr_dt_f64 = 1e-3;
  mbt::StepBuffers B{};
  B.state_in = s0; B.state_out = s1; B.action = act; B.reward = rew; B.wave_sums = ws; B.clip_count = clip;
  mbt::RolloutParams R{};
  R.n_steps = steps; R.last_is_terminal = 1; R.t_start = 0.0; R.dt_f64 = 1e-3; R.terminal_time = 1e-3 * steps;
  R.policy = mbt::kPolicyAvellanedaStoikov; R.as_c1 = 0.1f * 4.f; R.as_c2 = float(2.0 / 0.1 * std::log(1.0 + 0.1 / 1.5));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t lds_caps[2] = {0u, 32u * 1024u};  // full occupancy | five workgroups per CU (the AS step kernel's setting from 2^20 lanes up)
  for (int recorded = 0; recorded < 2; ++recorded) {
    for (uint32_t cap : lds_caps) {
      R.obs_traj = recorded ? obs_t : nullptr; R.act_traj = recorded ? act_t : nullptr; R.rew_traj = recorded ? rew_t : nullptr;
      for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((mbt::rollout_kernel<AS>), dim3(blocks), dim3(mbt::kBlockThreads), cap, 0, B, P, R);
      CK(hipDeviceSynchronize());
      const int reps = 9;
      float best = 1e30f, all[reps];
      for (int i = 0; i < reps; ++i) {
        P.philox_step = 1000u * i;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((mbt::rollout_kernel<AS>), dim3(blocks), dim3(mbt::kBlockThreads), cap, 0, B, P, R);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&all[i], e0, e1));
        best = all[i] < best ? all[i] : best;
      }
      std::sort(all, all + reps);
      const double us = all[reps / 2] * 1e3 / steps, us_min = best * 1e3 / steps;
      if (recorded) printf("  2^%d lanes x %4u steps, recorded (policy %d)%s  median %7.3f (min %7.3f) us/step  %9.3e env-steps/s  %6.0f GB/s written\n", lg, steps, MBT_RECORD_STORE_POLICY, cap ? ", 5 WG/CU" : "         ", us, us_min, n / us * 1e6, 28.0 * n / us * 1e-3);
      else printf("  2^%d lanes x %4u steps, returns only%s           median %7.3f (min %7.3f) us/step  %9.3e env-steps/s\n", lg, steps, cap ? ", 5 WG/CU" : "         ", us, us_min, n / us * 1e6);
    }
  }
  hipFree(s0); hipFree(s1); hipFree(act); hipFree(rew); hipFree(ws); hipFree(clip); hipFree(obs_t); hipFree(act_t); hipFree(rew_t);
  return 0;
}

int main() {
  printf("rollout_kernel<AS>, recording store policy %d\n", MBT_RECORD_STORE_POLICY);
  if (run(18, 400)) return 1;
  if (run(20, 200)) return 1;
  return 0;
}
