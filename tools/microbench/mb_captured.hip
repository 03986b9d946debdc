// What does it cost a step kernel to take its clock from device memory and to advance it there (step_kernel.hpp:
// captured_step_kernel)?  The production Avellaneda-Stoikov instantiation (state stepped in place, 32 KB of dynamic LDS from 2^20
// lanes up, like mbt_env.hip: tune_for_size) with the clock READ as in production (one scalar load in front of the generator) and
// several ways of finding out that the launch is over, so that one workgroup may write the next step's clock:
//   0 plain            step_kernel (the clock in the kernel arguments): the baseline
//   1 read only        the clock is read, nobody advances it (the cost of the read alone; not a working scheme)
//   2 wg returning     thread 0 of every workgroup, behind a workgroup barrier: returning atomic on the line of its group of 32 workgroups,
//                      the group's last on a top-level word (round 6, first version)
//   3 wave forget+poll every wave a fire-and-forget atomic on its group's line; the workgroup dispatched last polls the lines (second version)
//   4 wave returning   every wave a returning atomic on its group's line (group = blockIdx / 32), the last of a group on the top level
//   5 wave ret. spread the same with group = blockIdx % 256: workgroups that finish together count on different lines
//   6 two slots        no counting at all: the launch reads clock slot (parity) and workgroup 0 writes slot (parity ^ 1); the parity is a
//                      kernel argument that alternates from launch to launch
//   7 wave forget only every wave a fire-and-forget atomic, nobody looks (the cost of the atomics alone; not a working scheme)
//   8 one flat word    every wave a returning atomic on ONE word
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off mb_captured.hip -o mb_captured && ./mb_captured [log2 lanes ...]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../mbt_gym_amd/csrc/step_kernel.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

using AS = mbt::Variant<mbt::shape::brownian, mbt::shape::pnl>;

template <int SCHEME>
__global__ __launch_bounds__(mbt::kBlockThreads) void kernel(const mbt::StepBuffers B, const mbt::StepParams P0, mbt::CapturedParams C, uint32_t parity) {
  using namespace mbt;
  CapturedStep s = {0.0, 0.0, 0u, false};
  C.parity = SCHEME == 6 ? parity : 0u;  // (scheme 6 reads slot `parity` and writes the other; the counting schemes read and write slot 0)
  step_tile<AS, false, false, true>(B, P0, blockIdx.x, &C, &s);
  constexpr uint32_t kWaves = kBlockThreads / 64;
  ClockSlot* next = &C.clock->slot[SCHEME == 6 ? parity ^ 1u : 0u];
  bool last = false;
  if (SCHEME == 1) return;
  if (SCHEME == 2) {
    __syncthreads();
    if (threadIdx.x != 0u) return;
    const uint32_t group = blockIdx.x >> 5, groups = (gridDim.x + 31u) >> 5;
    const uint32_t members = gridDim.x - (group << 5) < 32u ? gridDim.x - (group << 5) : 32u;
    uint32_t* mine = C.counters + 16u * (1u + group);
    if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u != members) return;
    __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(C.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u != groups) return;
    __hip_atomic_store(C.counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = true;
  } else if (SCHEME == 3 || SCHEME == 7) {
    if ((threadIdx.x & 63u) == 0u) (void)__hip_atomic_fetch_add(C.counters + 16u * (1u + (blockIdx.x >> 5)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (SCHEME == 7) {  // (nobody re-arms: the counters just count on)
      return;
    }
    if (blockIdx.x + 1u != gridDim.x || threadIdx.x >= 64u) return;
    const uint32_t groups = (gridDim.x + 31u) >> 5;
    const uint64_t t0 = wall_clock64();
    for (;;) {
      bool all_in = true;
      for (uint32_t g = threadIdx.x; g < groups; g += 64u) {
        const uint32_t members = gridDim.x - (g << 5) < 32u ? gridDim.x - (g << 5) : 32u;
        all_in &= __hip_atomic_load(C.counters + 16u * (1u + g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members * kWaves;
      }
      if (__builtin_amdgcn_ballot_w64(!all_in) == 0ull) break;
      if (wall_clock64() - t0 > 200000000ull) break;
      __builtin_amdgcn_s_sleep(1);
    }
    for (uint32_t g = threadIdx.x; g < groups; g += 64u) __hip_atomic_store(C.counters + 16u * (1u + g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = threadIdx.x == 0u;
  } else if (SCHEME == 4 || SCHEME == 5) {
    if ((threadIdx.x & 63u) != 0u) return;
    const uint32_t lines = SCHEME == 4 ? (gridDim.x + 31u) >> 5 : (gridDim.x < 256u ? gridDim.x : 256u);
    const uint32_t group = SCHEME == 4 ? blockIdx.x >> 5 : blockIdx.x % lines;
    const uint32_t members = SCHEME == 4 ? (gridDim.x - (group << 5) < 32u ? gridDim.x - (group << 5) : 32u) : (gridDim.x - group + lines - 1u) / lines;
    uint32_t* mine = C.counters + 16u * (1u + group);
    if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u != members * kWaves) return;
    __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(C.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u != lines) return;
    __hip_atomic_store(C.counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = true;
  } else if (SCHEME == 6) {
    last = blockIdx.x == 0u && threadIdx.x == 0u;
  } else if (SCHEME == 8) {
    if ((threadIdx.x & 63u) != 0u) return;
    if (__hip_atomic_fetch_add(C.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u != gridDim.x * kWaves) return;
    __hip_atomic_store(C.counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = true;
  }
  if (last) {
    next->time = s.t_next;
    next->episode_step = C.clock->slot[C.parity].episode_step + 1u;
    next->philox_step = s.philox_step + 1u;
    next->steps = C.clock->slot[C.parity].steps + 1u;
  }
}

template <typename F>
float time_it(F launch, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) launch(i);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch(i);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    best = ms * 1e3f / iters < best ? ms * 1e3f / iters : best;
  }
  return best;
}

int run(uint32_t log2n) {
  const uint32_t n = 1u << log2n;
  const uint32_t n_pairs = n / 2, blocks = (n_pairs + mbt::kBlockThreads - 1) / mbt::kBlockThreads;
  float *s0, *act, *rew; double* ws; unsigned long long* clip; uint32_t* counters; mbt::DeviceClock* clock;
  CK(hipMalloc(&s0, size_t(n) * 16)); CK(hipMalloc(&act, size_t(n) * 8)); CK(hipMalloc(&rew, size_t(n) * 4));
  CK(hipMalloc(&ws, blocks * 4 * 8)); CK(hipMalloc(&clip, 8 * mbt::kClipSlots)); CK(hipMalloc(&counters, 64 * (2 + blocks))); CK(hipMalloc(&clock, sizeof(mbt::DeviceClock)));
  CK(hipMemset(ws, 0, blocks * 32)); CK(hipMemset(clip, 0, 8 * mbt::kClipSlots)); CK(hipMemset(counters, 0, 64 * (2 + blocks))); CK(hipMemset(clock, 0, sizeof(mbt::DeviceClock)));
  std::vector<float> h(size_t(n) * 4);
  for (uint32_t i = 0; i < n; ++i) { h[4 * i] = 0; h[4 * i + 1] = 0; h[4 * i + 2] = 0; h[4 * i + 3] = 100.f; }
  CK(hipMemcpy(s0, h.data(), size_t(n) * 16, hipMemcpyHostToDevice));
  std::vector<float> ha(size_t(n) * 2, 0.7f);
  CK(hipMemcpy(act, ha.data(), size_t(n) * 8, hipMemcpyHostToDevice));
  mbt::StepParams P{};
  P.n = n; P.n_pairs = n_pairs; P.key0 = 50; P.dt = 1e-3f; P.vol_sqrt_dt = 2.f * sqrtf(1e-3f);
  P.arr_thr_bid = P.arr_thr_ask = 0.14f; P.arr_thr_w_bid = P.arr_thr_w_ask = 0x23D70A00u; P.kappa_log2e_neg = -1.5f * 1.4426950408889634f; P.kappa_f64 = 1.5;
  P.fill_depth_per_log2 = -0.6931471805599453f / 1.5f; P.fill_band_abs = 2e-7f / 1.5f; P.q_max = 1000.f; P.c_max = 1e8f; P.reward_scale = 1.f; P.mid_add = 1.f;
  mbt::StepBuffers B{};
  B.state_in = B.state_out = s0; B.action = act; B.reward = rew; B.wave_sums = ws; B.clip_count = clip;
  mbt::CapturedParams C{};
  C.clock = clock; C.counters = counters; C.dt_f64 = 1e-9; C.terminal_time = 1e9; C.dim = 4; C.tile_lanes = mbt::kTileLanes; C.n_waves = blocks * 4;
  const uint32_t lds = log2n >= 20 && log2n < 23 ? 32u * 1024u : 0u;
  const int iters = n >= (1u << 22) ? 300 : 2000;
  float t[9];
  t[0] = time_it([&](int i) { P.philox_step = i; hipLaunchKernelGGL((mbt::step_kernel<AS>), dim3(blocks), dim3(mbt::kBlockThreads), lds, 0, B, P); }, iters);
#define RUN(S) t[S] = time_it([&](int i) { hipLaunchKernelGGL((kernel<S>), dim3(blocks), dim3(mbt::kBlockThreads), lds, 0, B, P, C, uint32_t(i & 1)); }, iters);
  RUN(1) RUN(2) CK(hipMemset(counters, 0, 64 * (2 + blocks))); RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) CK(hipMemset(counters, 0, 64 * (2 + blocks))); RUN(8)
  CK(hipDeviceSynchronize());
  printf("2^%-2u lanes (%5u workgroups): plain %6.2f | read only %6.2f | wg returning %6.2f | wave forget+poll %6.2f | wave returning %6.2f | wave ret. spread %6.2f | "
         "two slots %6.2f | wave forget only %6.2f | one flat word %6.2f  us\n", log2n, blocks, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]);
  hipFree(s0); hipFree(act); hipFree(rew); hipFree(ws); hipFree(clip); hipFree(counters); hipFree(clock);
  return 0;
}

int main(int argc, char** argv) {
  std::vector<uint32_t> sizes;
  for (int i = 1; i < argc; ++i) sizes.push_back(static_cast<uint32_t>(atoi(argv[i])));
  if (sizes.empty()) sizes = {10, 16, 18, 20, 22};
  for (int pass = 0; pass < 2; ++pass)
    for (uint32_t s : sizes)
      if (run(s) != 0) return 1;
  return 0;
}
