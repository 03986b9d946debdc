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
//   hawkes+r  6  2  76           2^22 lanes: the float32 tier's Hawkes rows with EXACT intensities (round 5): + an (N, 2) int32
//                                remainder buffer read and written in place (Variant::EXACT_LAM, 8 + 8 B per lane)
//   record    -  -  28 written   the fused rollout's trajectory recording (round 5): WRITE-ONLY, time-major (steps, N, 4) observation
//                                rows + (steps, N, 2) actions + (steps, N) rewards from ONE long-running kernel, under each store policy
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
#include <algorithm>
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

// Hawkes rows with exact intensities (step_kernel.hpp: Variant::EXACT_LAM): the rows6 shape + an (N, 2) int32 remainder buffer
// that every lane reads and writes in place (dwordx2 per lane: a wave covers four whole lines; written through like the rows).
template <bool NT = false>
__global__ __launch_bounds__(256) void rows6_resid_kernel(const float* __restrict__ s_in, float* __restrict__ s_out, const float* __restrict__ act,
                                                          float* __restrict__ rew, float* __restrict__ resid) {
  constexpr int D = 6, TILE = 512;
  __shared__ __attribute__((aligned(16))) float lds[TILE * D];
  const size_t tile = blockIdx.x;
  const float* in = s_in + tile * TILE * D;
  v2f row[2][3], a[2], lo[2];
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const int lane = threadIdx.x + 256 * l;
    const v2f* r = reinterpret_cast<const v2f*>(in) + lane * 3;
    row[l][0] = ld<NT>(&r[0]); row[l][1] = ld<NT>(&r[1]); row[l][2] = ld<NT>(&r[2]);
    a[l] = ld<NT>(&reinterpret_cast<const v2f*>(act)[tile * TILE + lane]);
    lo[l] = ld<NT>(&reinterpret_cast<const v2f*>(resid)[tile * TILE + lane]);
  }
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const int lane = threadIdx.x + 256 * l;
    row[l][0].x += a[l].x;
    row[l][2].x += lo[l].x;
    lo[l].y += row[l][2].y;
    v2f* r = reinterpret_cast<v2f*>(lds) + lane * 3;
    r[0] = row[l][0]; r[1] = row[l][1]; r[2] = row[l][2];
    st4(rew + tile * TILE + lane, a[l].y);
    st8(reinterpret_cast<v2f*>(resid) + tile * TILE + lane, lo[l]);
  }
  __syncthreads();
  v4f* out4 = reinterpret_cast<v4f*>(s_out + tile * TILE * D);
#pragma unroll
  for (int k = 0; k < 3; ++k) st16(&out4[threadIdx.x + k * 256], reinterpret_cast<const v4f*>(lds)[threadIdx.x + k * 256]);
}

// ---- write-only: the fused rollout's recording ---------------------------------------------------------------------------
// One launch, `steps` iterations per thread; in each, a thread (pair of lanes j, j + 256 of its 512-lane tile, like rollout_body)
// stores its two observation rows (dwordx4), its two actions (dwordx2) and its two rewards (dword) into the time-major
// recording - 28 B per lane and step, every wave-level store one contiguous span of whole lines - and reads nothing.
// POLICY: 0 plain write-back stores | 1 sc1 (agent scope: written through the L2, the step kernels' policy) | 2 nt | 3 sc0 sc1
// (system scope) | 4 nt sc1.  WORK: multiply-adds between two steps' stores (0: the pure store stream; ~400: the arithmetic of one
// env-step of a pair, to see how much of it hides behind the stores at this occupancy).
template <int POLICY>
__device__ __forceinline__ void rec16(v4f* p, v4f v) {
  if (POLICY == 0) *p = v;
  else if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
  else if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
}
template <int POLICY>
__device__ __forceinline__ void rec8(v2f* p, v2f v) {
  if (POLICY == 0) *p = v;
  else if (POLICY == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  else if (POLICY == 2) asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
  else if (POLICY == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
}
template <int POLICY>
__device__ __forceinline__ void rec4(float* p, float v) {
  if (POLICY == 0) *p = v;
  else if (POLICY == 1) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  else if (POLICY == 2) asm volatile("global_store_dword %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
  else if (POLICY == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dword %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
}

template <int POLICY>
__global__ __launch_bounds__(256) void record_kernel(float* __restrict__ obs, float* __restrict__ act, float* __restrict__ rew, uint32_t n, uint32_t steps, int work) {
  const uint32_t l0 = blockIdx.x * 512 + threadIdx.x, l1 = l0 + 256;
  v4f a = {float(l0), 1.f, 0.f, 100.f}, b = {float(l1), -1.f, 0.f, 100.f};
  float x = 1.0f + 1e-7f * threadIdx.x, y = 0.999f;
  for (uint32_t k = 0; k < steps; ++k) {
    for (int w = 0; w < work; ++w) { x = __builtin_fmaf(x, y, 0.25f); y = __builtin_fmaf(y, 0.9999f, 1e-4f * x); }  // dependent chain: not vectorisable away
    a.x += x; a.w += y; b.x -= y; b.w += x; a.z = b.z = float(k);
    const size_t t = k;
    rec16<POLICY>(reinterpret_cast<v4f*>(obs + t * n * 4) + l0, a);
    rec16<POLICY>(reinterpret_cast<v4f*>(obs + t * n * 4) + l1, b);
    rec8<POLICY>(reinterpret_cast<v2f*>(act + t * n * 2) + l0, v2f{a.w, a.x});
    rec8<POLICY>(reinterpret_cast<v2f*>(act + t * n * 2) + l1, v2f{b.w, b.x});
    rec4<POLICY>(rew + t * n + l0, x);
    rec4<POLICY>(rew + t * n + l1, y);
  }
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

void hawkes_exact_pattern(int lg, bool nt) {
  const size_t n = size_t(1) << lg;
  Buffers b = make_buffers(n, 6, 2);
  float* resid;
  CK(hipMalloc(&resid, n * 8));
  CK(hipMemcpy(resid, b.s0, n * 8, hipMemcpyDeviceToDevice));
  float* st[2] = {b.s0, b.s1};
  const uint32_t blocks = static_cast<uint32_t>(n / 512);
  const float t = nt ? time_it([&](int i) { hipLaunchKernelGGL(rows6_resid_kernel<true>, dim3(blocks), dim3(256), 0, 0, st[i & 1], st[(i & 1) ^ 1], b.act, b.rew, resid); }, 200)
                     : time_it([&](int i) { hipLaunchKernelGGL(rows6_resid_kernel<false>, dim3(blocks), dim3(256), 0, 0, st[i & 1], st[(i & 1) ^ 1], b.act, b.rew, resid); }, 200);
  printf("  %-52s %9.2f us  %7.0f GB/s  (76 B/lane)\n", nt ? "rows + (N, 2) remainders in place, nt loads" : "rows + (N, 2) remainders in place (production)", t, 76.0 * n / t * 1e-3);
  CK(hipFree(resid));
  free_buffers(b);
}

template <int POLICY>
void record_one(const char* label, int lg, uint32_t steps, int work) {
  const size_t n = size_t(1) << lg;
  float *obs, *act, *rew;
  CK(hipMalloc(&obs, n * 16 * steps)); CK(hipMalloc(&act, n * 8 * steps)); CK(hipMalloc(&rew, n * 4 * steps));
  const uint32_t blocks = static_cast<uint32_t>(n / 512);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(record_kernel<POLICY>, dim3(blocks), dim3(256), 0, 0, obs, act, rew, uint32_t(n), steps, work);  // (first touch of the pages)
  CK(hipDeviceSynchronize());
  const int reps = 9;
  float all[reps];
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(record_kernel<POLICY>, dim3(blocks), dim3(256), 0, 0, obs, act, rew, uint32_t(n), steps, work);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&all[i], e0, e1));
  }
  std::sort(all, all + reps);
  const double us_per_step = all[reps / 2] * 1e3 / steps, us_min = all[0] * 1e3 / steps;
  printf("  %-44s work %3d  median %7.3f (min %7.3f) us/step  %6.0f GB/s written\n", label, work, us_per_step, us_min, 28.0 * n / us_per_step * 1e-3);
  CK(hipFree(obs)); CK(hipFree(act)); CK(hipFree(rew));
}

void record_patterns(int lg, uint32_t steps) {
  printf("record  28 B written per lane and step, 2^%d lanes x %u steps (%.1f GB per launch)\n", lg, steps, 28.0 * (size_t(1) << lg) * steps * 1e-9);
  for (int work : {0}) {
    record_one<0>("plain write-back stores", lg, steps, work);
    record_one<1>("sc1 (written through the L2)", lg, steps, work);
    record_one<2>("nt", lg, steps, work);
    record_one<3>("sc0 sc1 (system scope)", lg, steps, work);
    record_one<4>("sc1 nt", lg, steps, work);
  }
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'r') {  // mb_floor record: the write-only patterns alone
    record_patterns(18, 400);
    record_patterns(20, 200);
    hawkes_exact_pattern(22, false);
    hawkes_exact_pattern(22, true);
    return 0;
  }
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
  printf("== Hawkes rows with exact intensities (76 B per lane), 2^22 lanes: 319 MB per launch ==\n");
  hawkes_exact_pattern(22, false);
  hawkes_exact_pattern(22, true);
  printf("== the fused rollout's recording: write-only, one long-running kernel ==\n");
  record_patterns(18, 400);
  record_patterns(20, 200);
  return 0;
}
