// A RESIDENT small-batch step: what would env.step() at N = 1000 cost if the kernel were already running?  (GPU box only.)
//
// Today one env.step() of a small batch is ONE launch whose last workgroup raises a flag in host memory (step_kernel.hpp:
// signal_host): ~6 us from the launch call to the kernel's first instruction, ~2.5 us of kernel, ~1 us until the host sees the
// flag.  The alternative measured here: a kernel of <= 4 workgroups that STAYS on the device, polls a doorbell word, reads the
// (N, 2) actions, does a step's worth of memory traffic and arithmetic on N lanes, mirrors (N, 4) observation rows and (N) rewards
// into host memory and raises a completion flag - the host rings the doorbell and spins on the flag.  Every wait on the device
// has a wall-clock time-out (the kernel leaves by itself after `idle_ticks` without a doorbell and after `life_ticks` in any case):
// a host that died cannot leave the device spinning.
//
//   (1) round trip per step for the doorbell / actions in HOST memory (mapped, coherent: the device polls over PCIe) and, if the
//       platform lets the host write device memory (fine-grained VRAM through the PCIe BAR), in DEVICE memory;
//   (2) the same with the one-launch-per-step scheme of the library, for reference (launch + flag kernel semantics);
//   (3) what the resident kernel costs a bandwidth-bound kernel on ANOTHER stream (a 44 B/lane copy at 2^20 lanes): its mean time
//       with and without the resident kernel polling beside it.
#include <hip/hip_runtime.h>

#include <immintrin.h>
#include <setjmp.h>
#include <signal.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr unsigned kExit = 0xFFFFFFFFu;
static sigjmp_buf g_probe;

struct Resident {
  const unsigned* doorbell;  // written by the host: the sequence number of the newest step, or kExit
  const float2* action;      // (n) actions, written by the host before it rings
  float4* state;             // (n) state rows, device memory
  float4* host_obs;          // (n) mirror in host memory
  float* host_reward;        // (n)
  unsigned* host_flag;       // completion: the sequence number of the newest finished step
  unsigned* host_exit;       // the kernel left before handling this sequence number (idle / lifetime time-out)
  unsigned* arrived;         // device counter of workgroups that finished the current step
  unsigned n, first_seq;
  unsigned long long idle_ticks, life_ticks;  // 100 MHz wall clock
};

// one "step" of a lane: the memory shape of the real kernel (row in, action in, row out, reward out, mirror) and ~150 dependent
// flops standing in for the Philox rounds and the dynamics
__device__ __forceinline__ void lane_step(const Resident& R, unsigned lane, unsigned seq) {
  float4 s = R.state[lane];
  const v2f av = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(R.action) + lane);  // (never cached: the host rewrites it every step)
  const float2 a = make_float2(av.x, av.y);
  float x = s.w, y = a.x + 1e-3f * seq;
#pragma unroll 1
  for (int k = 0; k < 75; ++k) { x = __builtin_fmaf(x, 0.9999f, y); y = __builtin_fmaf(y, 0.999f, 1e-4f * x); }
  s.x += a.y; s.y += 1.0f; s.z = 1e-3f * seq; s.w = x * 1e-9f + 100.0f;
  R.state[lane] = s;
  R.host_obs[lane] = s;
  R.host_reward[lane] = y * 1e-9f;
}

__global__ __launch_bounds__(256) void resident_kernel(const Resident R) {
  __shared__ unsigned s_cmd;
  const unsigned long long born = wall_clock64();
  for (unsigned seq = R.first_seq;; ++seq) {
    if (threadIdx.x == 0) {
      const unsigned long long t0 = wall_clock64();
      unsigned cmd;
      for (;;) {
        cmd = __hip_atomic_load(R.doorbell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (cmd == kExit || static_cast<int>(cmd - seq) >= 0) break;
        const unsigned long long now = wall_clock64();
        if (now - t0 > R.idle_ticks || now - born > R.life_ticks) { cmd = kExit; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      s_cmd = cmd;
    }
    __syncthreads();
    const unsigned cmd = s_cmd;
    __syncthreads();
    if (cmd == kExit) {
      if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(R.host_exit, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    for (unsigned lane = blockIdx.x * blockDim.x + threadIdx.x; lane < R.n; lane += gridDim.x * blockDim.x) lane_step(R, lane, seq);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned arrived = __hip_atomic_fetch_add(R.arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      if (arrived == gridDim.x) {
        __hip_atomic_store(R.arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(R.host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// the same step as ONE launch (the library's scheme): every workgroup mirrors, the last one raises the flag
__global__ __launch_bounds__(256) void one_launch_kernel(const Resident R, unsigned seq) {
  for (unsigned lane = blockIdx.x * blockDim.x + threadIdx.x; lane < R.n; lane += gridDim.x * blockDim.x) lane_step(R, lane, seq);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned arrived = __hip_atomic_fetch_add(R.arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (arrived == gridDim.x) {
      __hip_atomic_store(R.arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(R.host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ __launch_bounds__(256) void copy44(const v4f* s_in, v4f* s_out, const v2f* act, float* rew) {  // the AS step's traffic, 2 lanes per thread
  const unsigned l0 = blockIdx.x * 512 + threadIdx.x, l1 = l0 + 256;
  v4f a = s_in[l0], b = s_in[l1];
  const v2f c = act[l0], d = act[l1];
  a.x += c.x; b.x += d.x;
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(s_out + l0), "v"(a) : "memory");
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(s_out + l1), "v"(b) : "memory");
  rew[l0] = c.y; rew[l1] = d.y;
}

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

template <typename T>
static T* host_mapped(size_t count, T** dev) {
  T* p = nullptr;
  CHECK(hipHostMalloc(reinterpret_cast<void**>(&p), count * sizeof(T), hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(p, 0, count * sizeof(T));
  CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(dev), p, 0));
  return p;
}

static void report(const char* label, std::vector<double>& t) {
  std::sort(t.begin(), t.end());
  std::printf("  %-74s median %7.2f us   p10 %7.2f   p90 %7.2f   max %8.2f\n", label, t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10], t.back());
}

int main(int argc, char** argv) {
  const unsigned n = argc > 1 ? static_cast<unsigned>(std::atoi(argv[1])) : 1000u;
  const unsigned groups = std::min(4u, (n + 511u) / 512u);
  hipStream_t s, other;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
  Resident R{};
  unsigned *doorbell_d, *flag_d, *exit_d;
  float2* action_d; float4* obs_d; float* rew_d;
  unsigned* doorbell = host_mapped<unsigned>(16, &doorbell_d);
  unsigned* flag = host_mapped<unsigned>(16, &flag_d);
  unsigned* exited = host_mapped<unsigned>(16, &exit_d);
  float2* action = host_mapped<float2>(n, &action_d);
  float4* obs = host_mapped<float4>(n, &obs_d);
  float* rew = host_mapped<float>(n, &rew_d);
  CHECK(hipMalloc(&R.state, n * sizeof(float4)));
  CHECK(hipMemset(R.state, 0, n * sizeof(float4)));
  CHECK(hipMalloc(&R.arrived, 64));
  CHECK(hipMemset(R.arrived, 0, 64));
  R.doorbell = doorbell_d; R.action = action_d; R.host_obs = obs_d; R.host_reward = rew_d; R.host_flag = flag_d; R.host_exit = exit_d;
  R.n = n; R.idle_ticks = 200000ull /* 2 ms */; R.life_ticks = 400000000ull /* 4 s */;
  std::vector<float2> policy(n, make_float2(0.7f, 0.3f));
  const int reps = 3000;
  std::printf("resident small-batch step, N = %u lanes, %u workgroup(s); host times per step (actions in, doorbell, flag, %zu B mirrored out)\n", n, groups,
              size_t(n) * 20);

  // (2) the reference: one launch per step, host spins on the flag (the library's scheme)
  {
    std::vector<double> t;
    unsigned seq = 0;
    for (int r = 0; r < reps + 200; ++r) {
      const auto t0 = clk::now();
      std::memcpy(action, policy.data(), n * sizeof(float2));
      ++seq;
      hipLaunchKernelGGL(one_launch_kernel, dim3(groups), dim3(256), 0, s, R, seq);
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
      if (r >= 200) t.push_back(us(t0, clk::now()));
    }
    CHECK(hipStreamSynchronize(s));
    report("one launch per step, last workgroup raises the flag (the library today)", t);
  }

  // (1a) resident kernel, doorbell and actions in host memory
  auto resident_run = [&](const char* label, unsigned* bell, float2* act_host_view, bool background) {
    *flag = 0; *exited = 0;
    __atomic_store_n(bell, 0u, __ATOMIC_RELEASE);
    unsigned seq = 0;
    R.first_seq = 1;
    hipLaunchKernelGGL(resident_kernel, dim3(groups), dim3(256), 0, s, R);
    std::vector<double> t;
    unsigned relaunches = 0;
    for (int r = 0; r < reps + 200; ++r) {
      const auto t0 = clk::now();
      std::memcpy(act_host_view, policy.data(), n * sizeof(float2));
      ++seq;
      __atomic_store_n(bell, seq, __ATOMIC_RELEASE);
      _mm_sfence();  // (device memory is mapped write-combining on the host: without the fence the doorbell sat in a WC buffer for up to 2 ms)
      for (;;) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
        if (__atomic_load_n(exited, __ATOMIC_ACQUIRE) == seq) {  // it left (idle) before this step: start another one at this step
          __atomic_store_n(exited, 0u, __ATOMIC_RELEASE);
          R.first_seq = seq;
          hipLaunchKernelGGL(resident_kernel, dim3(groups), dim3(256), 0, s, R);
          ++relaunches;
        }
        if (us(t0, clk::now()) > 2e6) { std::printf("  %s: no answer for 2 s at step %u - giving up\n", label, seq); __atomic_store_n(bell, kExit, __ATOMIC_RELEASE); return; }
      }
      if (r >= 200) t.push_back(us(t0, clk::now()));
      (void)background;
    }
    __atomic_store_n(bell, kExit, __ATOMIC_RELEASE);
    CHECK(hipStreamSynchronize(s));
    report(label, t);
    if (relaunches) std::printf("    (%u relaunches after an idle exit)\n", relaunches);
  };
  resident_run("resident kernel, doorbell + actions in HOST memory (device polls over PCIe)", doorbell, action, false);

  // (1b) doorbell and actions in DEVICE memory the host can write (fine-grained VRAM through the BAR), if the platform allows it:
  // probed by writing through the device pointer under a SIGSEGV / SIGBUS guard
  {
    unsigned* bell_dev = nullptr;
    float2* act_dev = nullptr;
    const bool ok = hipExtMallocWithFlags(reinterpret_cast<void**>(&bell_dev), 4096, hipDeviceMallocFinegrained) == hipSuccess &&
                    hipExtMallocWithFlags(reinterpret_cast<void**>(&act_dev), std::max<size_t>(4096, n * sizeof(float2)), hipDeviceMallocFinegrained) == hipSuccess;
    (void)hipGetLastError();
    bool host_writable = false;
    if (ok) {
      CHECK(hipMemset(bell_dev, 0, 4096));
      CHECK(hipDeviceSynchronize());
      struct sigaction guard{}, old_segv{}, old_bus{};
      guard.sa_handler = [](int) { siglongjmp(g_probe, 1); };
      sigaction(SIGSEGV, &guard, &old_segv);
      sigaction(SIGBUS, &guard, &old_bus);
      if (sigsetjmp(g_probe, 1) == 0) {
        *reinterpret_cast<volatile unsigned*>(bell_dev) = 0u;
        host_writable = *reinterpret_cast<volatile unsigned*>(bell_dev) == 0u;
      }
      sigaction(SIGSEGV, &old_segv, nullptr);
      sigaction(SIGBUS, &old_bus, nullptr);
    }
    std::printf("  fine-grained device memory: %s; %s\n", ok ? "allocated" : "NOT available",
                host_writable ? "the host can write it" : "the host CANNOT write it (no doorbell in VRAM on this platform)");
    if (ok && host_writable) {
      Resident saved = R;
      R.doorbell = bell_dev; R.action = act_dev;
      resident_run("resident kernel, doorbell + actions in DEVICE memory (host writes through the BAR)", bell_dev, act_dev, false);
      R = saved;
    }
    if (bell_dev) (void)hipFree(bell_dev);
    if (act_dev) (void)hipFree(act_dev);
  }

  // (3) a bandwidth-bound kernel on another stream, with and without a resident kernel polling beside it
  {
    const unsigned lanes = 1u << 20;
    float *s0, *s1, *act, *rw;
    CHECK(hipMalloc(&s0, lanes * 16)); CHECK(hipMalloc(&s1, lanes * 16)); CHECK(hipMalloc(&act, lanes * 8)); CHECK(hipMalloc(&rw, lanes * 4));
    CHECK(hipMemset(s0, 1, lanes * 16)); CHECK(hipMemset(s1, 1, lanes * 16)); CHECK(hipMemset(act, 1, lanes * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto copies = [&](int iters) {
      float* st[2] = {s0, s1};
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(copy44, dim3(lanes / 512), dim3(256), 0, other, (const v4f*)st[i & 1], (v4f*)st[(i & 1) ^ 1], (const v2f*)act, rw);
      CHECK(hipEventRecord(e0, other));
      for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(copy44, dim3(lanes / 512), dim3(256), 0, other, (const v4f*)st[i & 1], (v4f*)st[(i & 1) ^ 1], (const v2f*)act, rw);
      CHECK(hipEventRecord(e1, other));
      CHECK(hipEventSynchronize(e1));
      float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
      return ms * 1e3 / iters;
    };
    const double alone = copies(3000);
    *flag = 0; *exited = 0;
    __atomic_store_n(doorbell, 0u, __ATOMIC_RELEASE);
    R.first_seq = 1; R.idle_ticks = 100000000ull /* 1 s: it keeps polling while the copies run */;
    hipLaunchKernelGGL(resident_kernel, dim3(groups), dim3(256), 0, s, R);
    const double beside_idle = copies(3000);
    // ... and with the resident kernel actually stepping (a host thread rings it as fast as it answers)
    std::atomic<bool> stop{false};
    unsigned stepped = 0;
    std::thread ringer([&] {
      unsigned seq = 0;
      while (!stop.load()) {
        std::memcpy(action, policy.data(), n * sizeof(float2));
        ++seq;
        __atomic_store_n(doorbell, seq, __ATOMIC_RELEASE);
        const auto t0 = clk::now();
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq && us(t0, clk::now()) < 1e5) {}
        ++stepped;
      }
    });
    const double beside_stepping = copies(3000);
    stop.store(true);
    ringer.join();
    __atomic_store_n(doorbell, kExit, __ATOMIC_RELEASE);
    CHECK(hipStreamSynchronize(s));
    std::printf("  a 44 B/lane copy at 2^20 lanes on ANOTHER stream: alone %.3f us; beside an idle resident kernel (polling) %.3f us (%+.2f %%); beside one that steps "
                "(%u steps meanwhile) %.3f us (%+.2f %%)\n", alone, beside_idle, 100.0 * (beside_idle / alone - 1.0), stepped, beside_stepping, 100.0 * (beside_stepping / alone - 1.0));
  }
  return 0;
}
