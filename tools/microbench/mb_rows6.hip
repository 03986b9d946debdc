// D = 6 rows (Hawkes: 24-byte rows, 8-byte aligned) - does the strided per-lane access cost anything, and would staging
// the tile through LDS (fully coalesced 16-byte global accesses, rows picked out of LDS) buy it back?
// Both kernels move 4*(6 + 2 + 6 + 1) = 60 B per lane with the step kernel's tile mapping (GPU box only).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// A: every lane reads its row as three 8-byte vectors (stride 24 B across lanes)
__global__ __launch_bounds__(256) void rows_direct(const float* s_in, float* s_out, const v2f* act, float* rew) {
  const uint32_t l0 = blockIdx.x * 512 + threadIdx.x;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t l = l0 + 256 * h;
    const v2f* row = reinterpret_cast<const v2f*>(s_in) + (size_t)l * 3;
    v2f a = row[0], b = row[1], c = row[2];
    const v2f ac = act[l];
    a.x += ac.x;
    v2f* out = reinterpret_cast<v2f*>(s_out) + (size_t)l * 3;
    out[0] = a; out[1] = b; out[2] = c;
    rew[l] = ac.y;
  }
}

// B: the tile (512 rows = 768 float4) is moved with contiguous 16-byte accesses and transposed through LDS
__global__ __launch_bounds__(256) void rows_lds(const float* s_in, float* s_out, const v2f* act, float* rew) {
  __shared__ __attribute__((aligned(16))) float tile[512 * 6];
  const v4f* src = reinterpret_cast<const v4f*>(s_in) + (size_t)blockIdx.x * 768;
  v4f* t4 = reinterpret_cast<v4f*>(tile);
#pragma unroll
  for (int k = 0; k < 3; ++k) t4[threadIdx.x + 256 * k] = src[threadIdx.x + 256 * k];
  const uint32_t l0 = blockIdx.x * 512 + threadIdx.x;
  const v2f ac0 = act[l0], ac1 = act[l0 + 256];
  __syncthreads();
  v2f r[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 3; ++k) r[h][k] = reinterpret_cast<const v2f*>(tile)[(threadIdx.x + 256 * h) * 3 + k];
  r[0][0].x += ac0.x;
  r[1][0].x += ac1.x;
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 3; ++k) reinterpret_cast<v2f*>(tile)[(threadIdx.x + 256 * h) * 3 + k] = r[h][k];
  __syncthreads();
  v4f* dst = reinterpret_cast<v4f*>(s_out) + (size_t)blockIdx.x * 768;
#pragma unroll
  for (int k = 0; k < 3; ++k) dst[threadIdx.x + 256 * k] = t4[threadIdx.x + 256 * k];
  rew[l0] = ac0.y;
  rew[l0 + 256] = ac1.y;
}

// B with the coalesced 16-byte stores written through the L2 (sc1) - whole cache lines per instruction, unlike A's pieces
__global__ __launch_bounds__(256) void rows_lds_through(const float* s_in, float* s_out, const v2f* act, float* rew) {
  __shared__ __attribute__((aligned(16))) float tile[512 * 6];
  const uint32_t l0 = blockIdx.x * 512 + threadIdx.x;
  v2f r[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 3; ++k) r[h][k] = reinterpret_cast<const v2f*>(s_in)[(size_t)(l0 + 256 * h) * 3 + k];  // direct loads
  const v2f ac0 = act[l0], ac1 = act[l0 + 256];
  r[0][0].x += ac0.x;
  r[1][0].x += ac1.x;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 3; ++k) reinterpret_cast<v2f*>(tile)[(threadIdx.x + 256 * h) * 3 + k] = r[h][k];
  __syncthreads();
  v4f* dst = reinterpret_cast<v4f*>(s_out) + (size_t)blockIdx.x * 768;
  const v4f* t4 = reinterpret_cast<const v4f*>(tile);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const v4f v = t4[threadIdx.x + 256 * k];
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(&dst[threadIdx.x + 256 * k]), "v"(v) : "memory");
  }
  asm volatile("global_store_dword %0, %1, off sc1" : : "v"(&rew[l0]), "v"(ac0.y) : "memory");
  asm volatile("global_store_dword %0, %1, off sc1" : : "v"(&rew[l0 + 256]), "v"(ac1.y) : "memory");
}

template <typename F>
float time_it(F launch, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch(i);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) launch(i);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (1u << atoi(argv[1])) : (1u << 22);
  float *s0, *s1, *act, *rew;
  CK(hipMalloc(&s0, (size_t)n * 24)); CK(hipMalloc(&s1, (size_t)n * 24)); CK(hipMalloc(&act, (size_t)n * 8)); CK(hipMalloc(&rew, (size_t)n * 4));
  CK(hipMemset(s0, 0, (size_t)n * 24)); CK(hipMemset(s1, 0, (size_t)n * 24)); CK(hipMemset(act, 0, (size_t)n * 8));
  {  // non-trivial data: the achievable rate depends on what is moved (mb_copy.hip)
    uint32_t* h = (uint32_t*)malloc((size_t)n * 24);
    uint32_t x = 12345u;
    for (size_t k = 0; k < (size_t)n * 6; ++k) { x = x * 1664525u + 1013904223u; h[k] = (x >> 9) | 0x3f800000u; }
    CK(hipMemcpy(s0, h, (size_t)n * 24, hipMemcpyHostToDevice)); CK(hipMemcpy(s1, h, (size_t)n * 24, hipMemcpyHostToDevice));
    CK(hipMemcpy(act, h, (size_t)n * 8, hipMemcpyHostToDevice));
    free(h);
  }
  float* st[2] = {s0, s1};
  const double bytes = 60.0 * n;
  float t = time_it([&](int i) { hipLaunchKernelGGL(rows_direct, dim3(n / 512), dim3(256), 0, 0, st[i & 1], st[(i & 1) ^ 1], (const v2f*)act, rew); }, 300);
  printf("D=6 rows, direct 3 x 8 B per lane   %8.2f us  %6.0f GB/s\n", t, bytes / t * 1e-3);
  t = time_it([&](int i) { hipLaunchKernelGGL(rows_lds, dim3(n / 512), dim3(256), 0, 0, st[i & 1], st[(i & 1) ^ 1], (const v2f*)act, rew); }, 300);
  printf("D=6 rows, tile staged through LDS   %8.2f us  %6.0f GB/s\n", t, bytes / t * 1e-3);
  t = time_it([&](int i) { hipLaunchKernelGGL(rows_lds_through, dim3(n / 512), dim3(256), 0, 0, st[i & 1], st[(i & 1) ^ 1], (const v2f*)act, rew); }, 300);
  printf("D=6 rows, direct loads, LDS-staged write-through stores %8.2f us  %6.0f GB/s\n", t, bytes / t * 1e-3);
  return 0;
}
