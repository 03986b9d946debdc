// Micro-benchmarks behind the kernel design decisions in DESIGN.md (GPU box only):
//   copy44      - a kernel that moves exactly the step's algorithmic traffic (2 float4 + 1 float4 in, 2 float4 +
//                 float2 out per pair) and does nothing else: the memory floor for this access pattern
//   philox3     - the Philox blocks + Box-Muller of a pair with no memory traffic: the RNG floor (named after the
//                 first, three-block layout)
//   step(philox)- the production kernel;  step(inject) - the same arithmetic with noise loaded from HBM
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../../mbt_gym_amd/csrc/step_kernel.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// the step kernel's own lane mapping (thread j of a 256-thread block: lanes tile*512 + j and + 256), nothing but the traffic
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(mbt::kBlockThreads) void copy44(const float4* s_in_, float4* s_out_, const float4* act_, float2* rew_, uint32_t n_pairs) {
  const v4f* s_in = reinterpret_cast<const v4f*>(s_in_);
  v4f* s_out = reinterpret_cast<v4f*>(s_out_);
  const v2f* act = reinterpret_cast<const v2f*>(act_);
  float* rew = reinterpret_cast<float*>(rew_);
  const uint32_t l0 = blockIdx.x * mbt::kTileLanes + threadIdx.x, l1 = l0 + mbt::kBlockThreads;
  v4f a = s_in[l0], b = s_in[l1];
  const v2f c = act[l0], d = act[l1];
  a.x += c.x; b.x += d.x;
  s_out[l0] = a; s_out[l1] = b;
  rew[l0] = c.y; rew[l1] = d.y;
}

__global__ __launch_bounds__(mbt::kBlockThreads) void philox3(float* sink, uint32_t n_pairs, uint32_t step, uint32_t k0, uint32_t k1) {
  const uint32_t p = blockIdx.x * mbt::kBlockThreads + threadIdx.x;
  mbt::LaneNoise a, b;
  mbt::philox_pair_noise(p, step, k0, k1, a, b);
  const float s = a.ua_bid + a.ua_ask + a.uf_bid + a.uf_ask + a.z + b.ua_bid + b.ua_ask + b.uf_bid + b.uf_ask + b.z;
  if (s == 12345.678f) sink[p] = s;  // never true: keeps the work alive without a store
}

// Experiment: a workgroup owns T consecutive tiles, issues all their loads first and then finishes them one after the
// other, so that the stores of tile t overlap the generator of tile t + 1 (the grid is then a fraction of one residency).
template <class V, int T>
__global__ __launch_bounds__(mbt::kBlockThreads) void step_kernel_tiles(const mbt::StepBuffers B, const mbt::StepParams P) {
  using namespace mbt;
  LaneLoads L0[T], L1[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const uint32_t lane0 = (blockIdx.x * T + t) * kTileLanes + threadIdx.x;
    L0[t] = load_lane<V>(B, P, lane0);
    L1[t] = load_lane<V>(B, P, lane0 + kBlockThreads);
  }
  bool clipped = false, c = false;
  float r_sum = 0.f;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const uint32_t tile = blockIdx.x * T + t;
    const uint32_t lane0 = tile * kTileLanes + threadIdx.x, lane1 = lane0 + kBlockThreads;
    const uint64_t pair = P.pair_offset + tile * kBlockThreads + threadIdx.x;
    LaneNoise nz0, nz1;
    philox_pair_noise(pair, P.philox_step, P.key0, P.key1, nz0, nz1);
    LaneDraw d0 = make_draw<V>(nz0, P), d1 = make_draw<V>(nz1, P);
    tie_loads_to_draws(L0[t], L1[t], d0, d1);
    r_sum += finish_lane<V>(B, P, lane0, L0[t], d0, c, nullptr);  // (D = 4 only: no staging)
    clipped |= c;
    r_sum += finish_lane<V>(B, P, lane1, L1[t], d1, c, nullptr);
    clipped |= c;
  }
  if (__builtin_expect(clipped, 0)) atomicAdd(B.clip_count, 1ull);
  const float total = wave_sum(r_sum);
  if ((threadIdx.x & 63u) == 0u) unsafeAtomicAdd(&B.wave_sums[blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6)], static_cast<double>(total));
}

// Experiment (the north star's "LDS-staged RNG draws"): the workgroup's noise goes THROUGH LDS between the generator and
// the arithmetic that consumes it.  MODE 0: each thread stages the ten draws of its own pair (the cost of staging alone).
// MODE 1: producer / consumer waves - waves 0-1 generate the draws of all 512 lanes of the tile (two pairs per thread)
// while waves 2-3 have nothing to compute and only keep their loads in flight; after the barrier every thread finishes
// its own pair from LDS.  Philox is counter-based: no state is shared between lanes, so staging cannot remove a single
// generator instruction; it adds 2 x 10 LDS accesses per pair and a workgroup barrier.  D = 4 variants only.
template <class V, int MODE>
__global__ __launch_bounds__(mbt::kBlockThreads) void step_kernel_lds_rng(const mbt::StepBuffers B, const mbt::StepParams P) {
  using namespace mbt;
  __shared__ float staged[2 * kTileLanes * 5];  // [lane in tile][ua_bid, ua_ask, uf_bid, uf_ask, z]
  const uint32_t lane0 = blockIdx.x * kTileLanes + threadIdx.x, lane1 = lane0 + kBlockThreads;
  LaneLoads L0 = load_lane<V>(B, P, lane0), L1 = load_lane<V>(B, P, lane1);
  auto put = [&](uint32_t slot, const LaneNoise& n) {
    float* p = staged + slot * 5;
    p[0] = n.ua_bid; p[1] = n.ua_ask; p[2] = n.uf_bid; p[3] = n.uf_ask; p[4] = n.z;
  };
  auto get = [&](uint32_t slot) {
    const float* p = staged + slot * 5;
    return LaneNoise{p[0], p[1], p[2], p[3], p[4]};
  };
  if (MODE == 0) {
    LaneNoise a, b;
    philox_pair_noise(P.pair_offset + blockIdx.x * kBlockThreads + threadIdx.x, P.philox_step, P.key0, P.key1, a, b);
    put(threadIdx.x, a);
    put(threadIdx.x + kBlockThreads, b);
  } else if (threadIdx.x < kBlockThreads / 2) {  // producer waves: two pairs each
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t j = threadIdx.x + h * (kBlockThreads / 2);
      LaneNoise a, b;
      philox_pair_noise(P.pair_offset + blockIdx.x * kBlockThreads + j, P.philox_step, P.key0, P.key1, a, b);
      put(j, a);
      put(j + kBlockThreads, b);
    }
  }
  __syncthreads();
  const LaneNoise nz0 = get(threadIdx.x), nz1 = get(threadIdx.x + kBlockThreads);
  LaneDraw d0 = make_draw<V>(nz0, P), d1 = make_draw<V>(nz1, P);
  bool c0, c1;
  float r = finish_lane<V>(B, P, lane0, L0, d0, c0, nullptr, nz0.z);
  r += finish_lane<V>(B, P, lane1, L1, d1, c1, nullptr, nz1.z);
  const uint32_t clips = __builtin_popcountll(__builtin_amdgcn_ballot_w64(c0)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(c1));
  const float total = wave_sum(r);
  if ((threadIdx.x & 63u) == 0u) {
    unsafeAtomicAdd(&B.wave_sums[blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6)], static_cast<double>(total));
    if (__builtin_expect(clips != 0u, 0)) atomicAdd(&B.clip_count[blockIdx.x & (kClipSlots - 1u)], static_cast<unsigned long long>(clips));
  }
}

template <typename F>
float time_it(F launch, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch(i);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) launch(i);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (1u << atoi(argv[1])) : (1u << 20);
  const uint32_t n_pairs = n / 2, blocks = (n_pairs + mbt::kBlockThreads - 1) / mbt::kBlockThreads;
  float *s0, *s1, *act, *rew, *ua, *uf, *z; double* ws; unsigned long long* clip;
  CK(hipMalloc(&s0, n * 32)); CK(hipMalloc(&s1, n * 32)); /* room for D = 6 / 8 rows */ CK(hipMalloc(&act, n * 8)); CK(hipMalloc(&rew, n * 4));
  CK(hipMalloc(&ua, n * 8)); CK(hipMalloc(&uf, n * 8)); CK(hipMalloc(&z, n * 4));
  CK(hipMalloc(&ws, blocks * 4 * 8)); CK(hipMalloc(&clip, 8 * mbt::kClipSlots));
  CK(hipMemset(s0, 0, n * 32)); CK(hipMemset(s1, 0, n * 32)); CK(hipMemset(act, 0, n * 8)); CK(hipMemset(ws, 0, blocks * 32)); CK(hipMemset(clip, 0, 8 * mbt::kClipSlots));
  CK(hipMemset(ua, 0, n * 8)); CK(hipMemset(uf, 0, n * 8)); CK(hipMemset(z, 0, n * 4));
  std::vector<float> h(n * 4);
  for (uint32_t i = 0; i < n; ++i) { h[4 * i] = 0; h[4 * i + 1] = 0; h[4 * i + 2] = 0; h[4 * i + 3] = 100.f; }
  CK(hipMemcpy(s0, h.data(), n * 16, hipMemcpyHostToDevice));
  std::vector<float> ha(n * 2, 0.7f);
  CK(hipMemcpy(act, ha.data(), n * 8, hipMemcpyHostToDevice));

  mbt::StepParams P{};
  P.n = n; P.n_pairs = n_pairs; P.key0 = 50; P.dt = 1e-3f; P.vol_sqrt_dt = 2.f * sqrtf(1e-3f);
  P.arr_thr_bid = P.arr_thr_ask = 0.14f; P.kappa_log2e_neg = -1.5f * 1.4426950408889634f; P.kappa_f64 = 1.5; P.q_max = 1000.f; P.c_max = 1e8f;
  P.reward_scale = 1.f; P.exponent_is_two = 1; P.exponent = 2.f; P.mid_add = 1.f; P.reward_kind = 2; P.alpha_cjmm = 0.001f; P.phi = 0.01f; P.quad_new = 1e-3f * 0.01f + 0.001f; P.quad_init = 0.001f * 1e-3f; P.arr_dt = 1e-3f; P.arr_dt_f64 = 1e-3;
  mbt::StepBuffers B{};
  B.action = act; B.reward = rew; B.u_arr = ua; B.u_fill = uf; B.z = z; B.wave_sums = ws; B.clip_count = clip;
  float* st[2] = {s0, s1};
  const int iters = 500;
  const double bytes = 44.0 * n;

  float t = time_it([&](int i) { hipLaunchKernelGGL(copy44, dim3(blocks), dim3(mbt::kBlockThreads), 0, 0, (const float4*)st[i & 1], (float4*)st[(i & 1) ^ 1], (const float4*)act, (float2*)rew, n_pairs); }, iters);
  printf("copy44        %8.2f us  %7.0f GB/s\n", t, bytes / t * 1e-3);
  t = time_it([&](int i) { hipLaunchKernelGGL(philox3, dim3(blocks), dim3(mbt::kBlockThreads), 0, 0, rew, n_pairs, (uint32_t)i, 50u, 0u); }, iters);
  printf("philox3       %8.2f us\n", t);
#define RUN(VARIANT, LABEL, BYTES)                                                                                  \
  t = time_it([&](int i) { B.state_in = st[i & 1]; B.state_out = st[(i & 1) ^ 1]; P.philox_step = i;                   \
                           hipLaunchKernelGGL((mbt::step_kernel<VARIANT>), dim3(blocks), dim3(mbt::kBlockThreads), 0, 0, B, P); }, iters); \
  printf("%-28s %8.2f us  %7.0f GB/s\n", LABEL, t, BYTES * n / t * 1e-3);
  using AS = mbt::Variant<mbt::shape::brownian, mbt::shape::pnl>;
  using ASI = mbt::Variant<mbt::shape::brownian, mbt::shape::pnl, mbt::shape::injected>;
  using CJ = mbt::Variant<mbt::shape::brownian, mbt::shape::quadratic>;
  RUN(AS, "step AS philox (44 B)", 44.0)
#define RUNT(VARIANT, T, LABEL)                                                                                          \
  t = time_it([&](int i) { B.state_in = st[i & 1]; B.state_out = st[(i & 1) ^ 1]; P.philox_step = i;                   \
                           hipLaunchKernelGGL((step_kernel_tiles<VARIANT, T>), dim3(blocks / T), dim3(mbt::kBlockThreads), 0, 0, B, P); }, iters); \
  printf("%-28s %8.2f us  %7.0f GB/s\n", LABEL, t, 44.0 * n / t * 1e-3);
#define RUNL(VARIANT, LDS, LABEL)                                                                                        \
  t = time_it([&](int i) { B.state_in = st[i & 1]; B.state_out = st[(i & 1) ^ 1]; P.philox_step = i;                   \
                           hipLaunchKernelGGL((mbt::step_kernel<VARIANT>), dim3(blocks), dim3(mbt::kBlockThreads), LDS, 0, B, P); }, iters); \
  printf("%-28s %8.2f us  %7.0f GB/s\n", LABEL, t, 44.0 * n / t * 1e-3);
  // occupancy limited through a dynamic LDS allocation (160 KB per CU): 7, 6, 5, 4, 3 workgroups per CU instead of 8
  RUNL(AS, 22 * 1024, "  AS, 7 workgroups/CU")
  RUNL(AS, 26 * 1024, "  AS, 6 workgroups/CU")
  RUNL(AS, 32 * 1024, "  AS, 5 workgroups/CU")
  RUNL(AS, 40 * 1024, "  AS, 4 workgroups/CU")
  RUNL(AS, 53 * 1024, "  AS, 3 workgroups/CU")
  RUNL(CJ, 32 * 1024, "  CjMm, 5 workgroups/CU")
#define RUNR(VARIANT, MODE, LABEL)                                                                                       \
  t = time_it([&](int i) { B.state_in = st[i & 1]; B.state_out = st[(i & 1) ^ 1]; P.philox_step = i;                   \
                           hipLaunchKernelGGL((step_kernel_lds_rng<VARIANT, MODE>), dim3(blocks), dim3(mbt::kBlockThreads), 0, 0, B, P); }, iters); \
  printf("%-28s %8.2f us  %7.0f GB/s\n", LABEL, t, 44.0 * n / t * 1e-3);
  RUNR(AS, 0, "  AS, draws staged in LDS")
  RUNR(AS, 1, "  AS, producer/consumer waves")
  RUNT(AS, 1, "  tiles/block 1")
  RUNT(AS, 2, "  tiles/block 2")
  RUNT(AS, 4, "  tiles/block 4")
  RUNT(CJ, 2, "  CjMm tiles/block 2")
  RUN(CJ, "step CjMm philox (44 B)", 44.0)
  RUN(ASI, "step AS inject (64 B)", 64.0)
  {  // upper bound of any scheme that overlaps the launch/drain bubbles of consecutive steps: the two HALVES of the batch
     // stepped as independent environments on two streams (no join between steps).  Not used by the library - a step
     // that a consumer can observe needs both halves - but it bounds what a single in-order stream leaves on the table.
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const uint32_t half_blocks = blocks / 2;
    mbt::StepParams PA = P, PB = P;
    PA.n = PB.n = n / 2; PA.n_pairs = PB.n_pairs = n_pairs / 2; PB.pair_offset = n_pairs / 2;
    mbt::StepBuffers BA = B, BB = B;
    BB.action = act + n; BB.reward = rew + n / 2; BB.wave_sums = ws + half_blocks * 4;  // second halves of the same buffers
    hipEvent_t e0, e1, eb;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&eb);
    auto both = [&](int i) {
      BA.state_in = st[i & 1]; BA.state_out = st[(i & 1) ^ 1];
      BB.state_in = st[i & 1] + (size_t)n * 2; BB.state_out = st[(i & 1) ^ 1] + (size_t)n * 2;  // rows n/2.. (4 floats each)
      PA.philox_step = PB.philox_step = i;
      hipLaunchKernelGGL((mbt::step_kernel<AS>), dim3(half_blocks), dim3(mbt::kBlockThreads), 0, sa, BA, PA);
      hipLaunchKernelGGL((mbt::step_kernel<AS>), dim3(half_blocks), dim3(mbt::kBlockThreads), 0, sb, BB, PB);
    };
    for (int i = 0; i < 20; ++i) both(i);
    hipDeviceSynchronize();
    hipEventRecord(e0, sa);
    for (int i = 0; i < iters; ++i) both(i);
    hipEventRecord(e1, sa); hipEventRecord(eb, sb);
    hipEventSynchronize(e1); hipEventSynchronize(eb);
    float ma = 0, mb2 = 0; hipEventElapsedTime(&ma, e0, e1); hipEventElapsedTime(&mb2, e0, eb);
    t = (ma > mb2 ? ma : mb2) * 1e3f / iters;
    printf("%-28s %8.2f us  %7.0f GB/s\n", "  AS, halves on two streams", t, 44.0 * n / t * 1e-3);
  }
  // where the Hawkes + OU kernel (BASELINE config 3) spends its time: each ingredient alone
  P.hawkes_base_bid = P.hawkes_base_ask = 10.f; P.hawkes_speed = 60.f; P.hawkes_jump = 40.f; P.ou_speed = 0.01f; P.ou_level = 100.f;
  using OU = mbt::Variant<mbt::shape::pnl>;
  using HK = mbt::Variant<mbt::shape::hawkes, mbt::shape::brownian, mbt::shape::pnl>;
  using HKOU = mbt::Variant<mbt::shape::hawkes, mbt::shape::pnl>;
  RUN(OU, "step Poisson + OU (44 B)", 44.0)
  RUN(HK, "step Hawkes + BM (60 B)", 60.0)
  RUN(HKOU, "step Hawkes + OU (60 B)", 60.0)
  return 0;
}
