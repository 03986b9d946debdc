// How many lanes should a thread of a one-round step kernel carry?  The speed family advances FOUR lanes per thread (one Philox call yields the
// four normals of a quad): 2^20 lanes are 4096 waves, four per SIMD, all resident at once.  The order-book family advances TWO (8192 waves, eight
// per SIMD) and moves its bytes faster (44 B per lane in 6.6 us against 40 B in 6.8).  Is that the lanes per thread?  A kernel with the speed family's
// D = 4 traffic - a 16-byte row and a 4-byte action in, the row (in place) and a 4-byte reward out: 40 B per lane - a Philox call per thread and a few
// dozen dependent FMAs per lane, with 4, 2 or 1 lanes per thread of a 1024-lane tile (lanes l * 256 + t of the tile, as speed_kernel.hpp lays them out;
// with fewer lanes per thread the SAME Philox call is repeated by the threads that share a quad, so the draws do not change):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mb_lanes_per_thread.hip -o mb_lanes_per_thread && ./mb_lanes_per_thread [log2 lanes]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../../mbt_gym_amd/csrc/philox.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Big {  // the size of the production kernels' arguments (StepBuffers + StepParams: ~1.3 KB by value)
  float f[320];
};
// MODE 0: sixteen parameters from all over a 1.3 KB by-value block   1: sixteen from its first 64 bytes   2: the block in device memory behind a pointer
// (the same address at every launch), sixteen from all over it
template <int MODE, int WORK>
__global__ __launch_bounds__(256) void kernel_big(float4* state, const float* action, float* reward, uint32_t step, const Big big, const Big* __restrict__ resident) {
  const uint32_t lane = blockIdx.x * 1024u + threadIdx.x;
  const mbt::PhiloxWords w = mbt::philox4x32_10(blockIdx.x * 256u + threadIdx.x, 0u, step, 3u, 17u, 29u);
  float z4[4];
  mbt::box_muller(w.w0, w.w1, z4[0], z4[1]);
  mbt::box_muller(w.w2, w.w3, z4[2], z4[3]);
  float4 row[4];
  float act[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) { row[l] = state[lane + l * 256u]; act[l] = action[lane + l * 256u]; }
  // sixteen parameters from all over the block, as the production kernels read theirs
  float c = 0.0f;
  if (MODE == 3) {  // the same sixteen, read through the kernarg segment pointer AFTER the vector loads were issued (the compiler hoists plain argument
                    // reads above everything and waits for them before the first vector load): a scheduling barrier + the pointer passed through an empty asm
    typedef const float __attribute__((address_space(4))) * kernarg_floats_t;
    __builtin_amdgcn_sched_barrier(0);
    kernarg_floats_t late = (kernarg_floats_t)(__builtin_amdgcn_kernarg_segment_ptr()) + 32 / 4;  // `big` sits behind three pointers and a word
    asm volatile("" : "+s"(late));
#pragma unroll
    for (int k = 0; k < 16; ++k) c += late[k * 20 + 3];
  } else {
#pragma unroll
  for (int k = 0; k < 16; ++k)  // MODE >= 10: sixteen reads spread over (MODE - 10) 64-byte lines of the by-value block
    c += MODE == 0 ? big.f[k * 20 + 3] : (MODE == 1 ? big.f[k] : (MODE == 2 ? resident->f[k * 20 + 3] : big.f[(k % (MODE - 10)) * 16 + k / (MODE - 10)]));
  }
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    float x = act[l] * 0.001f + c, acc = row[l].x;
#pragma unroll
    for (int k = 0; k < WORK; ++k) acc = __builtin_fmaf(acc, 0.999f, x + z4[l] * 1e-3f);
    float4 next = row[l];
    next.x = acc; next.y += x; next.w += z4[l] * 0.002f;
    state[lane + l * 256u] = next;
    reward[lane + l * 256u] = acc - row[l].x;
  }
}

template <int PER_THREAD, int WORK>
__global__ __launch_bounds__(256) void kernel(float4* state, const float* action, float* reward, uint32_t step) {
  constexpr uint32_t kParts = 4 / PER_THREAD;  // workgroups that share a 1024-lane tile
  const uint32_t tile = blockIdx.x / kParts, part = blockIdx.x % kParts;
  const uint32_t base = tile * 1024u + threadIdx.x;
  const mbt::PhiloxWords w = mbt::philox4x32_10(tile * 256u + threadIdx.x, 0u, step, 3u, 17u, 29u);
  float z4[4];
  mbt::box_muller(w.w0, w.w1, z4[0], z4[1]);
  mbt::box_muller(w.w2, w.w3, z4[2], z4[3]);
  float4 row[PER_THREAD];
  float act[PER_THREAD];
#pragma unroll
  for (int l = 0; l < PER_THREAD; ++l) {
    const uint32_t lane = base + (part * PER_THREAD + l) * 256u;
    row[l] = state[lane];
    act[l] = action[lane];
  }
#pragma unroll
  for (int l = 0; l < PER_THREAD; ++l) {
    const uint32_t lane = base + (part * PER_THREAD + l) * 256u;
    const float z = PER_THREAD == 4 ? z4[l] : (PER_THREAD == 2 ? (part ? z4[2 + l] : z4[l]) : (part == 0 ? z4[0] : part == 1 ? z4[1] : part == 2 ? z4[2] : z4[3]));
    float x = act[l] * 0.001f, acc = row[l].x;
#pragma unroll
    for (int k = 0; k < WORK; ++k) acc = __builtin_fmaf(acc, 0.999f, x + z * 1e-3f);
    float4 next = row[l];
    next.x = acc;
    next.y += x;
    next.w += z * 0.002f;
    state[lane] = next;
    reward[lane] = acc - row[l].x;
  }
}

template <int PER_THREAD, int WORK>
float run(float4* state, const float* action, float* reward, uint32_t n, int iters) {
  const uint32_t blocks = n / 1024u * (4 / PER_THREAD);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 200; ++i) kernel<PER_THREAD, WORK><<<blocks, 256>>>(state, action, reward, i);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) kernel<PER_THREAD, WORK><<<blocks, 256>>>(state, action, reward, i);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms * 1e3f / iters < best ? ms * 1e3f / iters : best;
  }
  return best;
}

template <int MODE, int WORK>
float run_big(float4* state, const float* action, float* reward, uint32_t n, int iters) {
  Big big;
  for (int i = 0; i < 320; ++i) big.f[i] = 1e-9f * i;
  Big* resident;
  hipMalloc(&resident, sizeof(Big));
  hipMemcpy(resident, &big, sizeof(Big), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 200; ++i) kernel_big<MODE, WORK><<<n / 1024u, 256>>>(state, action, reward, i, big, resident);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) kernel_big<MODE, WORK><<<n / 1024u, 256>>>(state, action, reward, i, big, resident);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms * 1e3f / iters < best ? ms * 1e3f / iters : best;
  }
  return best;
}

int main(int argc, char** argv) {
  const uint32_t log2n = argc > 1 ? atoi(argv[1]) : 20, n = 1u << log2n;
  float4* state; float *action, *reward;
  CK(hipMalloc(&state, size_t(n) * 16)); CK(hipMalloc(&action, size_t(n) * 4)); CK(hipMalloc(&reward, size_t(n) * 4));
  CK(hipMemset(state, 0, size_t(n) * 16)); CK(hipMemset(action, 0, size_t(n) * 4));
  const int iters = 2000;
  printf("2^%u lanes, 40 B per lane, us per launch (best of 3 x %d back-to-back launches); a Philox call + Box-Muller per thread, `work` dependent FMAs per lane\n", log2n, iters);
  printf("work   4 lanes/thread (4 waves/SIMD)   2 lanes/thread (8)   1 lane/thread (16)\n");
  printf("%4d   %10.2f %24.2f %20.2f\n", 16, run<4, 16>(state, action, reward, n, iters), run<2, 16>(state, action, reward, n, iters), run<1, 16>(state, action, reward, n, iters));
  printf("%4d   %10.2f %24.2f %20.2f\n", 64, run<4, 64>(state, action, reward, n, iters), run<2, 64>(state, action, reward, n, iters), run<1, 64>(state, action, reward, n, iters));
  printf("%4d   %10.2f %24.2f %20.2f\n", 128, run<4, 128>(state, action, reward, n, iters), run<2, 128>(state, action, reward, n, iters), run<1, 128>(state, action, reward, n, iters));
  printf("4 lanes/thread, 1.3 KB of kernel arguments by value, sixteen read from all over them:  work 16: %.2f   work 64: %.2f\n", run_big<0, 16>(state, action, reward, n, iters), run_big<0, 64>(state, action, reward, n, iters));
  printf("                the same block by value, sixteen read from its first 64 bytes:         work 16: %.2f   work 64: %.2f\n", run_big<1, 16>(state, action, reward, n, iters), run_big<1, 64>(state, action, reward, n, iters));
  printf("                by value, sixteen reads over 2 / 4 / 8 / 16 adjacent 64-byte lines (work 16):  %.2f / %.2f / %.2f / %.2f\n", run_big<12, 16>(state, action, reward, n, iters),
         run_big<14, 16>(state, action, reward, n, iters), run_big<18, 16>(state, action, reward, n, iters), run_big<26, 16>(state, action, reward, n, iters));
  printf("                by value, the sixteen from all over it read AFTER the vector loads were issued:  work 16: %.2f   work 64: %.2f\n", run_big<3, 16>(state, action, reward, n, iters), run_big<3, 64>(state, action, reward, n, iters));
  printf("                the block resident in device memory behind a pointer, sixteen read:    work 16: %.2f   work 64: %.2f\n", run_big<2, 16>(state, action, reward, n, iters), run_big<2, 64>(state, action, reward, n, iters));
  return 0;
}
