// Memory-floor experiments for the step's access pattern (GPU box only): block size, lane->row mapping,
// non-temporal hints.  Every kernel moves 44 B per trajectory: 16 B state in, 8 B action in, 16 B state out, 4 B reward.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// pair per thread, adjacent rows (the production mapping)
template <int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void copy_pair(const float4* s_in, float4* s_out, const float4* act, float2* rew, uint32_t n_pairs) {
  const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n_pairs) return;
  typedef float v4 __attribute__((ext_vector_type(4)));
  typedef float v2 __attribute__((ext_vector_type(2)));
  v4 a, b, c;
  const v4* in = reinterpret_cast<const v4*>(s_in);
  v4* out = reinterpret_cast<v4*>(s_out);
  if (NT) { a = __builtin_nontemporal_load(&in[2 * p]); b = __builtin_nontemporal_load(&in[2 * p + 1]); c = __builtin_nontemporal_load(reinterpret_cast<const v4*>(act) + p); }
  else { a = in[2 * p]; b = in[2 * p + 1]; c = reinterpret_cast<const v4*>(act)[p]; }
  a.x += c.x; b.x += c.z;
  v2 r = {c.y, c.w};
  if (NT) { __builtin_nontemporal_store(a, &out[2 * p]); __builtin_nontemporal_store(b, &out[2 * p + 1]); __builtin_nontemporal_store(r, reinterpret_cast<v2*>(rew) + p); }
  else { out[2 * p] = a; out[2 * p + 1] = b; reinterpret_cast<v2*>(rew)[p] = r; }
}

// NOTE: every kernel here uses clang ext_vector types.  With HIP's float4 (a struct of a union) a read-modify-write of
// one member made clang's "promote alloca to LDS" pass route the vector through LDS (ds_write/ds_read in the ISA),
// which halves the speed of such a micro-kernel and says nothing about the access pattern.  Check the .s for `ds_`.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// one row per thread: perfectly coalesced 16 B / lane
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void copy_row(const float4* s_in_, float4* s_out_, const float2* act_, float* rew, uint32_t n) {
  const v4f* s_in = reinterpret_cast<const v4f*>(s_in_);
  v4f* s_out = reinterpret_cast<v4f*>(s_out_);
  const v2f* act = reinterpret_cast<const v2f*>(act_);
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  v4f a = s_in[i];
  const v2f c = act[i];
  a.x += c.x;
  s_out[i] = a;
  rew[i] = c.y;
}

// pair per thread, rows t and t + BLOCK of the block's 2*BLOCK-row tile: every instruction is fully coalesced
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void copy_split(const float4* s_in_, float4* s_out_, const float2* act_, float* rew, uint32_t n) {
  const v4f* s_in = reinterpret_cast<const v4f*>(s_in_);
  v4f* s_out = reinterpret_cast<v4f*>(s_out_);
  const v2f* act = reinterpret_cast<const v2f*>(act_);
  const uint32_t i = blockIdx.x * (2 * BLOCK) + threadIdx.x;
  if (i + BLOCK >= n) return;
  v4f a = s_in[i], b = s_in[i + BLOCK];
  const v2f c = act[i], d = act[i + BLOCK];
  a.x += c.x; b.x += d.x;
  s_out[i] = a; s_out[i + BLOCK] = b;
  rew[i] = c.y; rew[i + BLOCK] = d.y;
}

// the tile-split copy with explicit cache-policy bits on the stores (gfx942/gfx950: sc0 / sc1 = coherence scope bits, nt =
// non-temporal): does writing THROUGH the XCD's L2 (instead of leaving 21 MB of dirty lines to the end-of-kernel
// write-back) shorten the launch-to-launch time?
#define STORE_ASM(BITS)                                                                                         \
  asm volatile("global_store_dwordx4 %0, %1, off " BITS : : "v"(&s_out[i]), "v"(a) : "memory");              \
  asm volatile("global_store_dwordx4 %0, %1, off " BITS : : "v"(&s_out[i + BLOCK]), "v"(b) : "memory");      \
  asm volatile("global_store_dword %0, %1, off " BITS : : "v"(&rew[i]), "v"(c.y) : "memory");                \
  asm volatile("global_store_dword %0, %1, off " BITS : : "v"(&rew[i + BLOCK]), "v"(d.y) : "memory");
template <int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK) void copy_split_policy(const float4* s_in_, float4* s_out_, const float2* act_, float* rew, uint32_t n) {
  const v4f* s_in = reinterpret_cast<const v4f*>(s_in_);
  v4f* s_out = reinterpret_cast<v4f*>(s_out_);
  const v2f* act = reinterpret_cast<const v2f*>(act_);
  const uint32_t i = blockIdx.x * (2 * BLOCK) + threadIdx.x;
  if (i + BLOCK >= n) return;
  v4f a = s_in[i], b = s_in[i + BLOCK];
  const v2f c = act[i], d = act[i + BLOCK];
  a.x += c.x; b.x += d.x;
  if (MODE == 0) { STORE_ASM("") }
  if (MODE == 1) { STORE_ASM("sc0") }
  if (MODE == 2) { STORE_ASM("sc1") }
  if (MODE == 3) { STORE_ASM("sc0 sc1") }
  if (MODE == 4) { STORE_ASM("nt") }
  if (MODE == 5) { STORE_ASM("sc1 nt") }
  if (MODE == 6) {  // no assembly: relaxed SYSTEM-scope atomic stores (64-bit halves of a row, 32-bit rewards) also carry sc0 sc1
    typedef unsigned long long u64;
    const u64* pa = reinterpret_cast<const u64*>(&a);
    const u64* pb = reinterpret_cast<const u64*>(&b);
    u64* oa = reinterpret_cast<u64*>(&s_out[i]);
    u64* ob = reinterpret_cast<u64*>(&s_out[i + BLOCK]);
    __hip_atomic_store(oa, pa[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(oa + 1, pa[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(ob, pb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(ob + 1, pb[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&rew[i], c.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&rew[i + BLOCK], d.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ... and on the LOADS (stores fixed at sc1): all four loads of the thread in one asm block, one wait at its end
#define LOAD_ASM(BITS)                                                                                                     \
  asm volatile("global_load_dwordx4 %0, %4, off " BITS "\n\tglobal_load_dwordx4 %1, %5, off " BITS                        \
               "\n\tglobal_load_dwordx2 %2, %6, off " BITS "\n\tglobal_load_dwordx2 %3, %7, off " BITS "\n\ts_waitcnt vmcnt(0)" \
               : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)                                                                     \
               : "v"(&s_in[i]), "v"(&s_in[i + BLOCK]), "v"(&act[i]), "v"(&act[i + BLOCK])                                  \
               : "memory");
template <int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK) void copy_split_load_policy(const float4* s_in_, float4* s_out_, const float2* act_, float* rew, uint32_t n) {
  const v4f* s_in = reinterpret_cast<const v4f*>(s_in_);
  v4f* s_out = reinterpret_cast<v4f*>(s_out_);
  const v2f* act = reinterpret_cast<const v2f*>(act_);
  const uint32_t i = blockIdx.x * (2 * BLOCK) + threadIdx.x;
  if (i + BLOCK >= n) return;
  v4f a, b;
  v2f c, d;
  if (MODE == 0) { LOAD_ASM("") }
  if (MODE == 1) { LOAD_ASM("sc0") }
  if (MODE == 2) { LOAD_ASM("sc1") }
  if (MODE == 3) { LOAD_ASM("sc0 sc1") }
  if (MODE == 4) { LOAD_ASM("nt") }
  if (MODE == 5) { LOAD_ASM("sc1 nt") }
  a.x += c.x; b.x += d.x;
  STORE_ASM("sc1")
}

__global__ void empty_kernel(uint32_t n) {}

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
  const uint32_t n = argc > 1 ? (1u << atoi(argv[1])) : (1u << 20);
  const uint32_t n_pairs = n / 2;
  float *s0, *s1, *act, *rew;
  // MB_FINEGRAINED=1: state and reward buffers in fine-grained device memory (system-coherent: is that the same as sc1 stores?)
  const bool fine = getenv("MB_FINEGRAINED") != nullptr;
  if (fine) {
    CK(hipExtMallocWithFlags((void**)&s0, n * 16, hipDeviceMallocFinegrained)); CK(hipExtMallocWithFlags((void**)&s1, n * 16, hipDeviceMallocFinegrained));
    CK(hipExtMallocWithFlags((void**)&rew, n * 4, hipDeviceMallocFinegrained)); CK(hipMalloc(&act, n * 8));
    printf("(fine-grained state / reward buffers)\n");
  } else {
    CK(hipMalloc(&s0, n * 16)); CK(hipMalloc(&s1, n * 16)); CK(hipMalloc(&act, n * 8)); CK(hipMalloc(&rew, n * 4));
  }
  CK(hipMemset(s0, 0, n * 16)); CK(hipMemset(s1, 0, n * 16)); CK(hipMemset(act, 0, n * 8));
  if (getenv("MB_RANDOM_DATA") != nullptr) {  // does the floor depend on WHAT is copied?  (all-zero buffers vs noise)
    uint32_t* h = (uint32_t*)malloc((size_t)n * 16);
    uint32_t x = 12345u;
    for (size_t k = 0; k < (size_t)n * 4; ++k) { x = x * 1664525u + 1013904223u; h[k] = (x >> 9) | 0x3f800000u; }  // floats in [1,2)
    CK(hipMemcpy(s0, h, (size_t)n * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(s1, h, (size_t)n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(act, h, (size_t)n * 8, hipMemcpyHostToDevice));
    free(h);
    printf("(random data)\n");
  }
  float* st[2] = {s0, s1};
  const int iters = 500;
  const double bytes = 44.0 * n;
  float t;
#define REPORT(LABEL) printf("%-34s %8.2f us  %7.0f GB/s\n", LABEL, t, bytes / t * 1e-3)
  t = time_it([&](int i) { hipLaunchKernelGGL(empty_kernel, dim3(2048), dim3(256), 0, 0, n); }, iters);
  printf("%-34s %8.2f us\n", "empty 2048x256", t);
  t = time_it([&](int i) { hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(1024), 0, 0, n); }, iters);
  printf("%-34s %8.2f us\n", "empty 512x1024", t);
#define PAIR(BLOCK, NT, LABEL) \
  t = time_it([&](int i) { hipLaunchKernelGGL((copy_pair<BLOCK, NT>), dim3((n_pairs + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, (const float4*)st[i & 1], (float4*)st[(i & 1) ^ 1], (const float4*)act, (float2*)rew, n_pairs); }, iters); REPORT(LABEL);
  PAIR(64, false, "pair/thread block 64")
  PAIR(128, false, "pair/thread block 128")
  PAIR(256, false, "pair/thread block 256")
  PAIR(512, false, "pair/thread block 512")
  PAIR(1024, false, "pair/thread block 1024")
  PAIR(256, true, "pair/thread block 256 nontemporal")
#define ROW(BLOCK, LABEL) \
  t = time_it([&](int i) { hipLaunchKernelGGL((copy_row<BLOCK>), dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, (const float4*)st[i & 1], (float4*)st[(i & 1) ^ 1], (const float2*)act, rew, n); }, iters); REPORT(LABEL);
  ROW(256, "row/thread block 256")
  ROW(512, "row/thread block 512")
  ROW(1024, "row/thread block 1024")
#define SPLIT(BLOCK, LABEL) \
  t = time_it([&](int i) { hipLaunchKernelGGL((copy_split<BLOCK>), dim3((n + 2 * BLOCK - 1) / (2 * BLOCK)), dim3(BLOCK), 0, 0, (const float4*)st[i & 1], (float4*)st[(i & 1) ^ 1], (const float2*)act, rew, n); }, iters); REPORT(LABEL);
  SPLIT(256, "2 rows/thread coalesced block 256")
  SPLIT(512, "2 rows/thread coalesced block 512")
#define POLICY(MODE, LABEL) \
  t = time_it([&](int i) { hipLaunchKernelGGL((copy_split_policy<256, MODE>), dim3((n + 511) / 512), dim3(256), 0, 0, (const float4*)st[i & 1], (float4*)st[(i & 1) ^ 1], (const float2*)act, rew, n); }, iters); REPORT(LABEL);
  POLICY(0, "  tile-split, stores (asm) default")
  POLICY(1, "  tile-split, stores sc0")
  POLICY(2, "  tile-split, stores sc1")
  POLICY(3, "  tile-split, stores sc0 sc1")
  POLICY(4, "  tile-split, stores nt")
  POLICY(5, "  tile-split, stores sc1 nt")
  POLICY(6, "  tile-split, system-scope atomic stores")
#define LPOLICY(MODE, LABEL) \
  t = time_it([&](int i) { hipLaunchKernelGGL((copy_split_load_policy<256, MODE>), dim3((n + 511) / 512), dim3(256), 0, 0, (const float4*)st[i & 1], (float4*)st[(i & 1) ^ 1], (const float2*)act, rew, n); }, iters); REPORT(LABEL);
  LPOLICY(0, "  stores sc1, loads (asm) default")
  LPOLICY(1, "  stores sc1, loads sc0")
  LPOLICY(2, "  stores sc1, loads sc1")
  LPOLICY(3, "  stores sc1, loads sc0 sc1")
  LPOLICY(4, "  stores sc1, loads nt")
  LPOLICY(5, "  stores sc1, loads sc1 nt")
  return 0;
}
