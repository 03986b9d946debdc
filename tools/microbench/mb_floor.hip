// Memory floors of EVERY step-kernel traffic pattern (GPU box only).
//
// A step moves, per lane: D floats of state in, A floats of action in, D floats of state out, 1 float of reward out.
// The FLAT kernels below move exactly those bytes between the same four buffers with nothing but whole-line 16-byte
// accesses (a tile of lanes is one contiguous span of each buffer, so row structure is irrelevant to a copy): no
// arithmetic, no row-width penalty - what the chip charges for the bytes alone, with the production kernel's store policy
// (write-through, sc1) and workgroup shape (256 threads, one 512-lane tile - 1024 for speed dynamics).  A step kernel
// within a few per cent of its flat kernel is at its floor; one that is not is losing time to HOW it touches its rows
// (e.g. five dword loads for a 20-byte row), and the ROWWISE kernels show which access shapes do that.
//
//   pattern   D  A  bytes/lane   config
//   as/cjp    4  2  44           2^20 lanes (BASELINE configs[1], [2])
//   market    4  4  52           2^21 lanes per GPU (configs[4])
//   hawkes    6  2  60           2^22 lanes (configs[3])
//   speed     4  1  40           2^20 lanes
//   speed+y   5  1  48           2^20 lanes
//
// Two regimes, as in production (mbt_env.hip: tune_for_size).  While a launch's working set fits the Infinity Cache the kernels
// use default-policy loads at full occupancy; beyond it (every pattern at 2^24 lanes) the production kernels load with the
// NON-TEMPORAL bit (their STREAM instantiation) and the 16-byte-row order-book kernels run capped at five workgroups per CU
// (32 KB of dynamic LDS).  Round 3's floors lacked both, so the 2^24 "floor" was 11 % ABOVE the kernel it was meant to bound
// (126.9 vs 113 us); the HBM-resident section now measures each pattern with the production policy (NT, + the cap where
// production caps) next to the default-policy figure.
//
// usage: mb_floor [log2 lanes]      (default: every pattern at its config size, then at 2^24 with the streaming policy)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <bool NT, typename T>
__device__ __forceinline__ T ld(const T* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void st16(v4f* p, v4f v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st8(v2f* p, v2f v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st4(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }

// ---- flat: whole-line accesses only -----------------------------------------------------------------------------------
template <int D, int A, int TILE, bool NT = false>
__global__ __launch_bounds__(256) void flat_kernel(const float* __restrict__ s_in, float* __restrict__ s_out, const float* __restrict__ act,
                                                   float* __restrict__ rew) {
  constexpr int SV = TILE * D / 4 / 256;            // float4 of state per thread
  constexpr int AV2 = TILE * A / 2 / 256;           // float2 of action per thread (A = 1, TILE = 1024: 2; A = 2, TILE = 512: 2; A = 4: 4)
  const size_t tile = blockIdx.x;
  const v4f* sin4 = reinterpret_cast<const v4f*>(s_in + tile * TILE * D);
  v4f* sout4 = reinterpret_cast<v4f*>(s_out + tile * TILE * D);
  const v4f* act4 = reinterpret_cast<const v4f*>(act + tile * TILE * A);
  v4f s[SV];
#pragma unroll
  for (int k = 0; k < SV; ++k) s[k] = ld<NT>(&sin4[threadIdx.x + k * 256]);
  float r_acc = 0.f;
  if (AV2 >= 2) {
    v4f a[AV2 / 2 > 0 ? AV2 / 2 : 1];
#pragma unroll
    for (int k = 0; k < AV2 / 2; ++k) a[k] = ld<NT>(&act4[threadIdx.x + k * 256]);
#pragma unroll
    for (int k = 0; k < AV2 / 2; ++k) r_acc += a[k].x + a[k].w;
  }
#pragma unroll
  for (int k = 0; k < SV; ++k) { s[k].x += r_acc; st16(&sout4[threadIdx.x + k * 256], s[k]); }
  if (TILE == 512) {  // 512 rewards = 2 KB: 8 bytes per thread
    v2f r = {r_acc, s[0].y};
    st8(reinterpret_cast<v2f*>(rew + tile * TILE) + threadIdx.x, r);
  } else {  // 1024 rewards: 16 bytes per thread
    v4f r = {r_acc, s[0].y, s[0].z, s[0].w};
    st16(reinterpret_cast<v4f*>(rew + tile * TILE) + threadIdx.x, r);
  }
}

// ---- rowwise: the access shapes the production kernels use for rows that are not 16 bytes wide -------------------------
// speed dynamics, D = 5: thread j owns lanes j + {0,256,512,768} of a 1024-lane tile; a 20-byte row is FIVE dword loads per
// lane (MODE 0, what speed_kernel.hpp does), or the tile is loaded flat into LDS and rows are read from there (MODE 1);
// stores always leave through LDS as whole lines (as in production).
template <int MODE, bool NT = false>
__global__ __launch_bounds__(256) void rows5_kernel(const float* __restrict__ s_in, float* __restrict__ s_out, const float* __restrict__ act,
                                                    float* __restrict__ rew) {
  constexpr int D = 5, TILE = 1024;
  __shared__ __attribute__((aligned(16))) float lds[TILE * D];
  const size_t tile = blockIdx.x;
  const float* in = s_in + tile * TILE * D;
  float row[4][D];
  float a[4];
  if (MODE == 0) {
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int lane = threadIdx.x + 256 * l;
#pragma unroll
      for (int c = 0; c < D; ++c) row[l][c] = ld<NT>(&in[lane * D + c]);
      a[l] = ld<NT>(&act[tile * TILE + lane]);
    }
  } else {
    const v4f* in4 = reinterpret_cast<const v4f*>(in);
    v4f t[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] = in4[threadIdx.x + k * 256];
#pragma unroll
    for (int l = 0; l < 4; ++l) a[l] = act[tile * TILE + threadIdx.x + 256 * l];
#pragma unroll
    for (int k = 0; k < 5; ++k) reinterpret_cast<v4f*>(lds)[threadIdx.x + k * 256] = t[k];
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int lane = threadIdx.x + 256 * l;
#pragma unroll
      for (int c = 0; c < D; ++c) row[l][c] = lds[lane * D + c];
    }
    __syncthreads();
  }
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const int lane = threadIdx.x + 256 * l;
    row[l][0] += a[l];
#pragma unroll
    for (int c = 0; c < D; ++c) lds[lane * D + c] = row[l][c];
    st4(rew + tile * TILE + lane, row[l][1]);
  }
  __syncthreads();
  v4f* out4 = reinterpret_cast<v4f*>(s_out + tile * TILE * D);
#pragma unroll
  for (int k = 0; k < 5; ++k) st16(&out4[threadIdx.x + k * 256], reinterpret_cast<const v4f*>(lds)[threadIdx.x + k * 256]);
}

// order book with Hawkes intensities, D = 6: thread j owns lanes j and j + 256 of a 512-lane tile; a 24-byte row is THREE
// dwordx2 loads per lane (MODE 0, step_kernel.hpp) or comes out of a flat-loaded LDS tile (MODE 1).
template <int MODE, bool NT = false>
__global__ __launch_bounds__(256) void rows6_kernel(const float* __restrict__ s_in, float* __restrict__ s_out, const float* __restrict__ act,
                                                    float* __restrict__ rew) {
  constexpr int D = 6, TILE = 512;
  __shared__ __attribute__((aligned(16))) float lds[TILE * D];
  const size_t tile = blockIdx.x;
  const float* in = s_in + tile * TILE * D;
  v2f row[2][3];
  v2f a[2];
  if (MODE == 0) {
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int lane = threadIdx.x + 256 * l;
      const v2f* r = reinterpret_cast<const v2f*>(in) + lane * 3;
      row[l][0] = ld<NT>(&r[0]); row[l][1] = ld<NT>(&r[1]); row[l][2] = ld<NT>(&r[2]);
      a[l] = ld<NT>(&reinterpret_cast<const v2f*>(act)[tile * TILE + lane]);
    }
  } else {
    const v4f* in4 = reinterpret_cast<const v4f*>(in);
    v4f t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = in4[threadIdx.x + k * 256];
#pragma unroll
    for (int l = 0; l < 2; ++l) a[l] = reinterpret_cast<const v2f*>(act)[tile * TILE + threadIdx.x + 256 * l];
#pragma unroll
    for (int k = 0; k < 3; ++k) reinterpret_cast<v4f*>(lds)[threadIdx.x + k * 256] = t[k];
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const v2f* r = reinterpret_cast<const v2f*>(lds) + (threadIdx.x + 256 * l) * 3;
      row[l][0] = r[0]; row[l][1] = r[1]; row[l][2] = r[2];
    }
    __syncthreads();
  }
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const int lane = threadIdx.x + 256 * l;
    row[l][0].x += a[l].x;
    v2f* r = reinterpret_cast<v2f*>(lds) + lane * 3;
    r[0] = row[l][0]; r[1] = row[l][1]; r[2] = row[l][2];
    st4(rew + tile * TILE + lane, a[l].y);
  }
  __syncthreads();
  v4f* out4 = reinterpret_cast<v4f*>(s_out + tile * TILE * D);
#pragma unroll
  for (int k = 0; k < 3; ++k) st16(&out4[threadIdx.x + k * 256], reinterpret_cast<const v4f*>(lds)[threadIdx.x + k * 256]);
}

// the production shape for 16-byte rows (pair per thread, rows j and j + 256; action dwordx2 or dwordx4; rewards dword)
template <int A, bool NT = false>
__global__ __launch_bounds__(256) void rows4_kernel(const float* __restrict__ s_in, float* __restrict__ s_out, const float* __restrict__ act,
                                                    float* __restrict__ rew) {
  const size_t i = size_t(blockIdx.x) * 512 + threadIdx.x;
  const v4f* in4 = reinterpret_cast<const v4f*>(s_in);
  v4f a = ld<NT>(&in4[i]), b = ld<NT>(&in4[i + 256]);
  float ra, rb;
  if (A == 4) {
    const v4f c = ld<NT>(&reinterpret_cast<const v4f*>(act)[i]), d = ld<NT>(&reinterpret_cast<const v4f*>(act)[i + 256]);
    a.x += c.x + c.z; b.x += d.x + d.w; ra = c.y; rb = d.y;
  } else {
    const v2f c = ld<NT>(&reinterpret_cast<const v2f*>(act)[i]), d = ld<NT>(&reinterpret_cast<const v2f*>(act)[i + 256]);
    a.x += c.x; b.x += d.x; ra = c.y; rb = d.y;
  }
  st16(reinterpret_cast<v4f*>(s_out) + i, a);
  st16(reinterpret_cast<v4f*>(s_out) + i + 256, b);
  st4(rew + i, ra);
  st4(rew + i + 256, rb);
}

// speed dynamics, D = 4: quad per thread (rows j + {0,256,512,768}), action / reward dword
template <bool NT = false>
__global__ __launch_bounds__(256) void rows4_quad_kernel(const float* __restrict__ s_in, float* __restrict__ s_out, const float* __restrict__ act,
                                                         float* __restrict__ rew) {
  const size_t i = size_t(blockIdx.x) * 1024 + threadIdx.x;
  const v4f* in4 = reinterpret_cast<const v4f*>(s_in);
  v4f s[4];
  float a[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) { s[l] = ld<NT>(&in4[i + 256 * l]); a[l] = ld<NT>(&act[i + 256 * l]); }
#pragma unroll
  for (int l = 0; l < 4; ++l) { s[l].x += a[l]; st16(reinterpret_cast<v4f*>(s_out) + i + 256 * l, s[l]); st4(rew + i + 256 * l, s[l].y); }
}

template <typename F>
float time_it(F launch, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 30; ++i) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch(i);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

struct Buffers { float *s0, *s1, *act, *rew; };

Buffers make_buffers(size_t n, int d, int a) {
  Buffers b;
  CK(hipMalloc(&b.s0, n * d * 4)); CK(hipMalloc(&b.s1, n * d * 4)); CK(hipMalloc(&b.act, n * a * 4)); CK(hipMalloc(&b.rew, n * 4));
  // non-trivial data: the floor depends on WHAT is copied (all-zero buffers are 10-15 % cheaper, profiles/r01_microbench.txt)
  std::vector<uint32_t> h(n * d);
  uint32_t x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x >> 9) | 0x3f800000u; }
  CK(hipMemcpy(b.s0, h.data(), n * d * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b.s1, h.data(), n * d * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b.act, h.data(), n * a * 4, hipMemcpyHostToDevice));
  CK(hipMemset(b.rew, 0, n * 4));
  return b;
}
void free_buffers(Buffers& b) { (void)hipFree(b.s0); (void)hipFree(b.s1); (void)hipFree(b.act); (void)hipFree(b.rew); }

template <typename K>
void run(const char* label, K kernel, int tile, size_t n, int d, int a, int iters, uint32_t dynamic_lds = 0) {
  Buffers b = make_buffers(n, d, a);
  float* st[2] = {b.s0, b.s1};
  const uint32_t blocks = static_cast<uint32_t>(n / tile);
  const float t = time_it([&](int i) { hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), dynamic_lds, 0, st[i & 1], st[(i & 1) ^ 1], b.act, b.rew); }, iters);
  const double bytes = 4.0 * (2 * d + a + 1) * n;
  printf("  %-52s %9.2f us  %7.0f GB/s  (%2d B/lane)\n", label, t, bytes / t * 1e-3, 4 * (2 * d + a + 1));
  free_buffers(b);
}

void all_patterns(int lg_as, int lg_market, int lg_hawkes, int lg_speed) {
  const int it = 400;
  printf("as/cjp  D=4 A=2, 2^%d lanes\n", lg_as);
  run("flat (whole lines)", flat_kernel<4, 2, 512>, 512, size_t(1) << lg_as, 4, 2, it);
  run("rows: pair/thread, dwordx4 + dwordx2 (production)", rows4_kernel<2>, 512, size_t(1) << lg_as, 4, 2, it);
  printf("market  D=4 A=4, 2^%d lanes\n", lg_market);
  run("flat (whole lines)", flat_kernel<4, 4, 512>, 512, size_t(1) << lg_market, 4, 4, it);
  run("rows: pair/thread, dwordx4 + dwordx4 (production)", rows4_kernel<4>, 512, size_t(1) << lg_market, 4, 4, it);
  printf("hawkes  D=6 A=2, 2^%d lanes\n", lg_hawkes);
  run("flat (whole lines)", flat_kernel<6, 2, 512>, 512, size_t(1) << lg_hawkes, 6, 2, it / 2);
  run("rows: 3 x dwordx2 loads, LDS-staged stores (production)", rows6_kernel<0>, 512, size_t(1) << lg_hawkes, 6, 2, it / 2);
  run("rows: flat loads -> LDS -> rows, LDS-staged stores", rows6_kernel<1>, 512, size_t(1) << lg_hawkes, 6, 2, it / 2);
  printf("speed   D=4 A=1, 2^%d lanes\n", lg_speed);
  run("flat (whole lines)", flat_kernel<4, 1, 1024>, 1024, size_t(1) << lg_speed, 4, 1, it);
  run("rows: quad/thread, dwordx4 + dword (production)", rows4_quad_kernel<false>, 1024, size_t(1) << lg_speed, 4, 1, it);
  printf("speed+y D=5 A=1, 2^%d lanes\n", lg_speed);
  run("flat (whole lines)", flat_kernel<5, 1, 1024>, 1024, size_t(1) << lg_speed, 5, 1, it);
  run("rows: 5 x dword loads, LDS-staged stores (production)", rows5_kernel<0>, 1024, size_t(1) << lg_speed, 5, 1, it);
  run("rows: flat loads -> LDS -> rows, LDS-staged stores", rows5_kernel<1>, 1024, size_t(1) << lg_speed, 5, 1, it);
}

// The production policy beyond the Infinity Cache: non-temporal loads everywhere; five workgroups per CU (32 KB of dynamic LDS)
// for the order-book kernels with 16-byte rows (mbt_env.hip: tune_for_size).
void streaming_patterns(int lg) {
  const int it = 200;
  const size_t n = size_t(1) << lg;
  const uint32_t cap = 32u * 1024u;
  printf("as/cjp  D=4 A=2, 2^%d lanes\n", lg);
  run("flat, nt loads", flat_kernel<4, 2, 512, true>, 512, n, 4, 2, it);
  run("flat, nt loads, 5 WG/CU", flat_kernel<4, 2, 512, true>, 512, n, 4, 2, it, cap);
  run("rows (production shape), nt loads", rows4_kernel<2, true>, 512, n, 4, 2, it);
  run("rows (production shape), nt loads, 5 WG/CU = PRODUCTION", rows4_kernel<2, true>, 512, n, 4, 2, it, cap);
  printf("market  D=4 A=4, 2^%d lanes\n", lg);
  run("flat, nt loads, 5 WG/CU", flat_kernel<4, 4, 512, true>, 512, n, 4, 4, it, cap);
  run("rows (production shape), nt loads, 5 WG/CU = PRODUCTION", rows4_kernel<4, true>, 512, n, 4, 4, it, cap);
  printf("hawkes  D=6 A=2, 2^%d lanes\n", lg);
  run("flat, nt loads", flat_kernel<6, 2, 512, true>, 512, n, 6, 2, it / 2);
  run("rows: 3 x dwordx2 nt loads, LDS-staged stores = PRODUCTION", rows6_kernel<0, true>, 512, n, 6, 2, it / 2);
  printf("speed   D=4 A=1, 2^%d lanes\n", lg);
  run("flat, nt loads", flat_kernel<4, 1, 1024, true>, 1024, n, 4, 1, it);
  run("rows: quad/thread, nt loads = PRODUCTION", rows4_quad_kernel<true>, 1024, n, 4, 1, it);
  printf("speed+y D=5 A=1, 2^%d lanes\n", lg);
  run("flat, nt loads", flat_kernel<5, 1, 1024, true>, 1024, n, 5, 1, it);
  run("rows: 5 x dword nt loads, LDS-staged stores = PRODUCTION", rows5_kernel<0, true>, 1024, n, 5, 1, it);
}

int main(int argc, char** argv) {
  if (argc > 1) {
    const int lg = atoi(argv[1]);
    all_patterns(lg, lg, lg, lg);
    return 0;
  }
  printf("== every pattern at its BASELINE config size ==\n");
  all_patterns(20, 21, 22, 20);
  printf("== every pattern at 2^24 lanes (HBM-resident), default-policy loads at full occupancy (NOT what production runs there) ==\n");
  all_patterns(24, 24, 24, 24);
  printf("== every pattern at 2^24 lanes (HBM-resident), the production policy: non-temporal loads (+ 5 WG/CU where production caps) ==\n");
  streaming_patterns(24);
  return 0;
}
