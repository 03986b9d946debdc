// How long after the last of k short kernels does the host know?  k launches of a ~7 us kernel between two host
// synchronisation points, several ways of waiting.  (bench.py at the driver's `--steps 20`: the fixed cost is ~20 % of the
// timed region.)   hipcc --offload-arch=gfx950 -O3 mb_sync.hip -o mb_sync && ./mb_sync
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                    \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                 \
    }                                                                               \
  } while (0)

__global__ void spin_kernel(unsigned long long ticks, float* sink) {  // wall_clock64: 100 MHz
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
  }
  if (sink != nullptr && threadIdx.x == 0 && blockIdx.x == 0) sink[0] = 1.0f;
}

__global__ void flag_kernel(volatile unsigned* flag, unsigned value) { *flag = value; }

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

int main() {
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t ev_begin, ev_end, ev_notime;
  CHECK(hipEventCreate(&ev_begin));
  CHECK(hipEventCreate(&ev_end));
  CHECK(hipEventCreateWithFlags(&ev_notime, hipEventDisableTiming));
  unsigned* flag_host = nullptr;
  CHECK(hipHostMalloc(reinterpret_cast<void**>(&flag_host), 64, hipHostMallocMapped | hipHostMallocCoherent));
  unsigned* flag_dev = nullptr;
  CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&flag_dev), flag_host, 0));
  *flag_host = 0;
  const unsigned long long ticks = 680;  // 6.8 us
  const int blocks = 2048, threads = 256, reps = 300;
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), 0, s, 50ull, nullptr);
  CHECK(hipStreamSynchronize(s));
  unsigned seq = 0;
  for (int k : {1, 5, 20, 100}) {
    for (int mode = 0; mode < 8; ++mode) {
      std::vector<double> t;
      for (int r = 0; r < reps; ++r) {
        CHECK(hipStreamSynchronize(s));
        const bool events = mode == 5 || mode == 6;
        const auto t0 = clk::now();
        if (events) CHECK(hipEventRecord(ev_begin, s));
        for (int i = 0; i < k; ++i) hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), 0, s, ticks, nullptr);
        switch (mode) {
          case 0: CHECK(hipStreamSynchronize(s)); break;                                  // blocking
          case 1: while (hipStreamQuery(s) == hipErrorNotReady) {} break;                 // poll the stream
          case 2: CHECK(hipEventRecord(ev_notime, s)); while (hipEventQuery(ev_notime) == hipErrorNotReady) {} break;
          case 3: CHECK(hipEventRecord(ev_notime, s)); CHECK(hipEventSynchronize(ev_notime)); break;
          case 4:  // a one-thread kernel writes a sequence number to host-coherent memory; the host spins on the memory
            ++seq;
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, s, flag_dev, seq);
            while (*reinterpret_cast<volatile unsigned*>(flag_host) != seq) {}
            break;
          case 5: CHECK(hipEventRecord(ev_end, s)); while (hipEventQuery(ev_end) == hipErrorNotReady) {} break;  // timing events both sides
          case 6:  // timing events + flag kernel
            CHECK(hipEventRecord(ev_end, s));
            ++seq;
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, s, flag_dev, seq);
            while (*reinterpret_cast<volatile unsigned*>(flag_host) != seq) {}
            break;
          case 7:  // flag, then the device-wide synchronise a framework adds on top
            ++seq;
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, s, flag_dev, seq);
            while (*reinterpret_cast<volatile unsigned*>(flag_host) != seq) {}
            CHECK(hipDeviceSynchronize());
            break;
        }
        t.push_back(us(t0, clk::now()));
      }
      std::sort(t.begin(), t.end());
      const char* names[] = {"hipStreamSynchronize", "poll hipStreamQuery", "event(no timing) + poll hipEventQuery", "event(no timing) + hipEventSynchronize",
                             "flag kernel -> host memory spin", "timing events both sides + poll", "timing events + flag kernel", "flag kernel + hipDeviceSynchronize"};
      std::printf("k=%3d  %-42s median %8.2f us  (kernels %7.1f us, fixed %6.2f)  p10 %8.2f  p90 %8.2f\n", k, names[mode], t[reps / 2], k * 6.8,
                  t[reps / 2] - k * 6.8, t[reps / 10], t[reps * 9 / 10]);
    }
  }
  return 0;
}
