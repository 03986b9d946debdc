// libmbtenv: host side of the C ABI declared in include/mbt_env.h.
//
// Owns the HBM-resident environment (ping-pong state, action/reward/noise buffers), the host clock that the
// reference keeps in state[0, TIME] (TradingEnvironment.py:216-220), and the launch of the fused step kernel.
// No CPU fallback exists in this file or anywhere in the product path: without a gfx950 device every entry
// point that needs one fails with MBT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is bound with dlopen (see rccl_api below), never linked

#include <dlfcn.h>
#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <vector>

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/mbt_env.h"
#include "kernel_table.hpp"

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) return fail(MBT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                                \
  } while (0)

int check_device(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(MBT_ERR_NO_DEVICE, "no HIP device visible: libmbtenv has no CPU path");
  if (device < 0 || device >= count) return fail(MBT_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, count);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(MBT_ERR_NO_DEVICE, "device %d is %s; libmbtenv is built for gfx950 only", device, prop.gcnArchName);
  return MBT_OK;
}

// smallest float32 >= x: for float32 u, (u < x in double) <=> (u < round_up_f32(x))
float round_up_f32(double x) {
  float f = static_cast<float>(x);
  if (static_cast<double>(f) < x) f = std::nextafterf(f, INFINITY);
  return f;
}

// precise_state keeps a float64 value x as hi = float32(x) in the state row and lo = (x - hi) * 2^(53 - exponent(hi)) as an
// int32 beside it (step_kernel.hpp: exact_join / exact_split); the same two functions for the host side (reset values,
// mbt_env_get_state_f64_host).
int f32_biased_exponent_host(float hi) {
  uint32_t bits;
  std::memcpy(&bits, &hi, sizeof bits);
  return static_cast<int>((bits >> 23) & 0xffu);
}
void exact_split_host(double x, float& hi, int32_t& lo) {
  hi = static_cast<float>(x);
  const int e = f32_biased_exponent_host(hi);
  lo = (e != 0 && e != 255) ? static_cast<int32_t>(std::ldexp(x - static_cast<double>(hi), (127 + 53) - e)) : 0;
}
double exact_join_host(float hi, int32_t lo) { return static_cast<double>(hi) + std::ldexp(static_cast<double>(lo), f32_biased_exponent_host(hi) - (127 + 53)); }

// Pinned host memory handed to callers (mbt_host_alloc): a host pointer inside one of these blocks is DMA-able as it is, so
// the "*_host" entry points copy straight into / out of it; any other host pointer is pageable as far as the library knows
// (a foreign pinned allocation is recognised through hipPointerGetAttributes) and goes through a pinned bounce buffer.
struct PinnedBlock {
  size_t bytes = 0;
  uintptr_t device_base = 0;  // the same block as the DEVICE addresses it (mapped, coherent): 0 if it could not be mapped
};
std::mutex g_pinned_mutex;
std::map<uintptr_t, PinnedBlock> g_pinned_blocks;  // base address -> block

// The device's address of [p, p + bytes) if that range lies inside one block of mbt_host_alloc (nullptr otherwise): memory the
// small-batch step kernel may write its outputs into DIRECTLY (mbt_env_step_host: no staging copy on the way out).
void* device_alias(const void* p, size_t bytes) {
  if (p == nullptr) return nullptr;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  std::lock_guard<std::mutex> guard(g_pinned_mutex);
  auto it = g_pinned_blocks.upper_bound(a);
  if (it == g_pinned_blocks.begin()) return nullptr;
  --it;
  if (a < it->first || a + bytes > it->first + it->second.bytes || it->second.device_base == 0) return nullptr;
  return reinterpret_cast<void*>(it->second.device_base + (a - it->first));
}

bool is_pinned_host(const void* p, size_t bytes) {
  if (p == nullptr) return false;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  {
    std::lock_guard<std::mutex> guard(g_pinned_mutex);
    auto it = g_pinned_blocks.upper_bound(a);
    if (it != g_pinned_blocks.begin()) {
      --it;
      if (a >= it->first && a + bytes <= it->first + it->second.bytes) return true;
    }
  }
  // foreign memory: pinned only if the runtime knows BOTH ends of [p, p + bytes) as host memory (a registration may cover
  // less than the caller's array)
  const void* ends[2] = {p, static_cast<const char*>(p) + (bytes > 0 ? bytes - 1 : 0)};
  for (const void* q : ends) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, q) != hipSuccess) {
      (void)hipGetLastError();  // an ordinary malloc'ed pointer: not an error of ours
      return false;
    }
    if (attr.type != hipMemoryTypeHost) return false;
  }
  return true;
}

// (per environment: callers pass the same buffers step after step, and for a pageable one the classification above costs
// driver calls every time - the answer for the last (pointer, size) of each role is remembered)
// (an address can change KIND under the memo: a block of mbt_host_alloc freed and the same address handed out again by malloc, or the
// reverse - every mbt_host_alloc / mbt_host_free advances a generation that the memo's answer is tied to)
std::atomic<uint64_t> g_pinned_generation{0};
struct PinnedMemo {
  const void* p = nullptr;
  size_t bytes = 0;
  uint64_t generation = ~uint64_t(0);
  bool pinned = false;
  bool lookup(const void* q, size_t n) {
    const uint64_t now = g_pinned_generation.load(std::memory_order_acquire);
    if (q == p && n == bytes && generation == now) return pinned;
    p = q;
    bytes = n;
    generation = now;
    return pinned = is_pinned_host(q, n);
  }
};

// up to this many lanes the host API stages through device-mapped pinned memory (mbt_env_step_host); measured per step,
// DMA path vs mapped staging: 52 vs 33 us at 8192 lanes, 85 vs 70 at 32768, 138 vs 131 at 65536, 149 vs 228 at 131072 (round 3: a second
// launch exported the outputs).  Round 4, the one-launch mirror: 74.5 (DMA) vs 69.5 us at 65536 lanes with a pinned action, 84.2 vs 69.9
// with a pageable one, 108.6 vs 95.5 through the SB3 adapter (profiles/r04_experiments.txt 7): the threshold moved to 65536.
// MBT_HOST_FAST_PATH_LANES overrides it (measurement knob).
constexpr uint32_t kHostFastPathLanes = 65536;

using mbt_table::StepKernel;
using mbt_table::RolloutKernel;
using mbt_table::LearnedRolloutKernel;
using mbt_table::ResidentKernel;
using mbt_table::kPlain;
using mbt_table::kStream;
using mbt_table::kMirror;
using mbt_table::exact_intensities;
using mbt_table::exogenous_fill;
using mbt_table::host_impact;
using mbt_table::impact_has_state;
using mbt_table::reward_weight;
using mbt_table::speed_powers;

// arrivals as the kernel table counts them: 0 Poisson-type | 1 Hawkes, float32 intensities | 2 Hawkes, exact intensities
int arrival_family(const mbt_config& c) { return c.arrival_kind != MBT_ARR_HAWKES ? 0 : (exact_intensities(c) ? 2 : 1); }

// The instantiations themselves live in csrc/kernels_*.hip (kernel_table.hpp); this is the configuration logic in front of them.
StepKernel pick_kernel(const mbt_config& c, int mode) {
  const bool norm = c.normalise_action != 0 || c.normalise_observation != 0;
  const bool inject = c.noise_mode == MBT_NOISE_INJECTED;
  if (c.dynamics_kind == MBT_DYN_SPEED) return mbt_table::pick_step_speed(c, mode);
  const bool brownian = c.midprice_kind == MBT_MID_BROWNIAN;
  const int tier = reward_weight(c);
  if (c.precise_state) {
    const bool special = !inject && !norm && !exogenous_fill(c) && tier != mbt::kRewardGeneral;
    return mbt_table::pick_step_precise(c.arrival_kind == MBT_ARR_HAWKES, c.dynamics_kind, exogenous_fill(c), special ? tier : mbt::kRewardGeneral, brownian, inject, mode);
  }
  if (exogenous_fill(c)) return mbt_table::pick_step_exogenous(arrival_family(c), c.dynamics_kind == MBT_DYN_LIMIT_AND_MARKET, inject, mode);
  switch (arrival_family(c)) {
    case 2: return mbt_table::pick_step_hawkes_exact(c.dynamics_kind, brownian, tier, norm, inject, mode);
    case 1: return mbt_table::pick_step_hawkes(c.dynamics_kind, brownian, tier, norm, inject, mode);
    default: return mbt_table::pick_step_poisson(c.dynamics_kind, brownian, tier, norm, inject, mode);
  }
}

RolloutKernel pick_rollout_kernel(const mbt_config& c) {
  const bool norm = c.normalise_action != 0 || c.normalise_observation != 0;
  if (c.dynamics_kind == MBT_DYN_SPEED) return mbt_table::pick_rollout_speed(c);
  if (c.precise_state) return mbt_table::pick_rollout_precise(c.arrival_kind == MBT_ARR_HAWKES, c.dynamics_kind, exogenous_fill(c));
  if (exogenous_fill(c)) return mbt_table::pick_rollout_exogenous(arrival_family(c), c.dynamics_kind == MBT_DYN_LIMIT_AND_MARKET);
  return mbt_table::pick_rollout_order_book(arrival_family(c), c.dynamics_kind, c.midprice_kind == MBT_MID_BROWNIAN, reward_weight(c), norm);
}

}  // namespace


struct mbt_env {
  mbt_config cfg;
  int dim = 4, act_dim = 2;
  uint32_t n = 0, n_pad = 0, n_pairs = 0, n_blocks = 0, n_waves = 0;
  bool speed = false;        // trading-with-speed family: one thread per QUAD of lanes, else one per PAIR
  double dt = 0.0, mid_dt = 0.0, arr_dt = 0.0, imp_dt = 0.0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_sums = nullptr;   // mbt_env_return_sums_begin / _end
  double* h_sums = nullptr;       // pinned: [sum of rewards, sum of squared per-lane returns]
  bool sums_pending = false;
  // device buffers
  // Launches that move 80 MB or more update the state IN PLACE (round 5): a lane reads its own row and writes its own row (rows
  // that leave through LDS are written by other threads of the SAME workgroup, behind the barrier that follows every thread's last
  // use of its loads; the speed kernels' per-wave spans are whole lines of the wave's own rows), so one buffer is enough, and the
  // set of lines a launch touches shrinks by a third to a half - which decides on which side of the 256 MB Infinity Cache a
  // launch lives: Hawkes + OU at 2^22 lanes (76 B rows: 319 -> 218 MB distinct) 49-54 -> 45.9 us, its precise_state tier 57.0 ->
  // 55.1, Avellaneda-Stoikov at 2^21 lanes 14.2 -> 12.1 us and at 2^23 (default-policy loads) 63.8 -> 51.5 us
  // (profiles/r05_in_place_state.txt).  Below that size the two-buffer scheme of rounds 1-4 stays (step k reads one, writes the
  // other): in place, the write-through stores hit lines the loads have just brought into the L2, which measured 1.5-3 % SLOWER
  // for the 2^20-lane kernels that are not the lightest one (CJP 6.67 -> 6.79 us, precise_state AS 9.38 -> 9.63).
  // Either way an observation is valid until the next step is ENQUEUED (mbt_env_obs_ptr) - a normalised observation never
  // outlived that (one `obs` buffer), and a zero-copy consumer that acts on an observation has consumed it before it can enqueue
  // the next step.  MBT_PING_PONG_STATE = 0 / 1 overrides the choice (measurement knob).
  float* state[2] = {nullptr, nullptr};
  int cur = 0;  // state[cur] holds the current state
  bool ping_pong = false;
  float* obs = nullptr;
  float* action = nullptr;
  // host-API fast path for small batches: pinned, device-mapped staging [actions (n_pad x A) | obs (n x D) | rewards (n)]
  float* h_stage = nullptr;   // host view
  float* d_stage = nullptr;   // the same memory as the device sees it
  size_t stage_action = 0, stage_obs = 0, stage_reward = 0;  // offsets in floats
  float* reward = nullptr;
  float* u_arr = nullptr;
  float* u_fill = nullptr;
  float* z = nullptr;
  float* q_init = nullptr;
  int32_t* resid = nullptr;    // precise_state: (n_pad, res) int32 remainders of the float64 state (step_kernel.hpp: exact_join)
  int res = 0;                 // remainder columns per lane: [cash, midprice] (+ [bid, ask intensity]), speed: [cash, inventory, midprice, y];
                               // float32 tier with exact Hawkes intensities (Variant::EXACT_LAM): [bid, ask intensity]
  int res_col[4] = {0, 3, 4, 5};  // the state column behind each remainder column
  uint8_t* events = nullptr;
  float* lane_returns = nullptr;
  double* wave_sums = nullptr;
  unsigned long long* clip_count = nullptr;
  double* reduce_out = nullptr;
  float* policy_table = nullptr;  // device copy of a tabulated policy
  size_t policy_table_floats = 0;
  // host clock (the reference keeps it in state[:, TIME])
  double time = 0.0, start_time = 0.0;
  uint32_t episode_step = 0, philox_step = 0;
  uint64_t seed = 0;
  bool was_reset = false, noise_ready = false, q_init_per_lane = false;
  bool record_events = false, track_returns = false;
  StepKernel kernel = nullptr;
  StepKernel kernel_mirror = nullptr;  // the small-batch host-API instantiation (nullptr: none - the two-launch fallback serves the host path)
  RolloutKernel rollout = nullptr;
  mbt::StepParams params;
  hipFunction_t jit_step = nullptr, jit_rollout = nullptr, jit_step_mirror = nullptr;  // run-time compiled kernels of mbt_env_create_jit (owned by the module cache)
  double user_fill_p[8] = {}, user_reward_p[8] = {}, user_arrival_p[8] = {}, user_mid_p[8] = {}, user_state_p[8] = {};  // parameters of the user's device expressions
  int user_state_columns = 0;      // state columns owned by user processes (mbt_user_code.state_columns), after the midprice
  bool user_draws = false;         // ... that read the two extra normals z1, z2
  double user_state_initial[2] = {0.0, 0.0};
  float* z_user = nullptr;         // injected-noise mode: (n_pad, 2) extra normals
  bool user_noise_ready = false;
  char* learned_dev = nullptr;                               // packed weights of a learned policy (policy_mlp.hpp), device
  std::vector<char> learned_host;                            // what learned_dev holds (re-uploaded only when the policy changes)
  uint32_t step_dynamic_lds = 0;   // occupancy control of the step kernel, see tune_for_size()
  bool stream_loads = false;       // the non-temporal-load instantiation of the step kernel, see tune_for_size()
  bool q0_per_lane_reset = false;  // the last explicit reset passed per-lane initial inventories (kept in q_init for auto-reset)
  // episode log of mbt_env_step_many_device: a ring of reductions in flight
  static constexpr uint32_t kLogSlots = 16;
  double* log_dev = nullptr;       // kLogSlots x 3 doubles, device
  double* log_host = nullptr;      // the same, pinned host
  hipEvent_t log_event[kLogSlots] = {};
  uint32_t log_head = 0, log_count = 0;  // oldest entry, entries in flight
  void* comm = nullptr;            // ncclComm_t of the episode log (mbt_env_set_communicator)
  // staging of mbt_env_rollout_host trajectories (grow-only)
  // pinned bounce buffers of the host API for callers that pass pageable memory: [actions | observation | rewards] (grow-only)
  float* h_bounce = nullptr;
  size_t bounce_floats = 0;
  float* traj_stage[3] = {nullptr, nullptr, nullptr};
  size_t traj_stage_floats[3] = {0, 0, 0};
  PinnedMemo memo_action, memo_obs, memo_reward;  // what the caller's host buffers were found to be (mbt_env_step_host)
  // small-batch host API: the step kernel mirrors its outputs into the stage and raises a flag there (step_kernel.hpp: signal_host)
  uint32_t* done_counter = nullptr;  // device: workgroups of the launch that have finished
  size_t stage_flag = 0;             // offset (in floats) of the flag word inside the stage
  uint32_t flag_seq = 0;             // value the next mirrored launch writes there
  bool action_in_stage = false;      // the newest actions sit in the stage, not (yet) in `action` (filed on demand: file_staged_action)
  // launch gate (mbt_env_set_launch_gate): bursts of launches enqueued behind a kernel that waits for a host flag
  uint32_t gate_chunk = 0, gate_seq = 0;
  uint32_t* h_gate = nullptr;        // pinned, device-mapped
  uint32_t* d_gate = nullptr;
  bool gate_closed = false;
  // host-callback plugins (Variant::HOST): per-step inputs computed by the caller's NumPy code
  int host_mask = 0;                 // mbt::kHostFill | kHostArrival | kHostReward
  double* host_fill_p = nullptr;     // (n_pad, 2) device
  float* host_arrivals = nullptr;    // (n_pad, 2) device
  double* host_scratch = nullptr;    // (n_pad, 2) device: depths out / rewards in
  bool host_fill_ready = false, host_arrivals_ready = false, host_reward_pending = false;
  int host_state_first = 0, host_state_count = 0;  // the state columns host-callback processes own: one block in registry order (TE:303-318)
  bool host_reward_replaces = false; // MBT_REW_HOST on the speed kernels: the kernel filed its PnL, mbt_env_set_host_rewards takes it back out
  bool host_mid = false;             // MBT_MID_HOST: cfg.midprice_kind reads MBT_MID_CONSTANT, the caller moves the midprice between launches
  // small batches (round 5): what goes to and comes from the caller's NumPy code travels through ONE pinned, device-mapped block per
  // direction instead of a DMA copy + synchronisation each - host_fill_p / host_arrivals / host_scratch then point INTO h_callback_in
  // (the kernels read it across the link), and the step's mirror instantiation writes rows + remainders + event bytes into the stage
  char* h_callback_in = nullptr;     // host view of [fill probabilities (n_pad, 2) f64 | arrivals (n_pad, 2) f32 | scratch (n_pad, 4) f64]
  // a kernel that reads the block in place may still be queued: set where one is enqueued and nobody waits (a device-API step of a
  // host-callback environment, the reward-filing kernel), cleared by mbt_env_step_host's completion flag / any stream wait; the
  // setters wait for the stream before they overwrite a region that is busy (never in the env.step() loop, whose flag wait precedes them)
  bool callback_inputs_busy = false;   // probabilities / arrivals / impacts: read by the step kernel
  bool callback_scratch_busy = false;  // scratch: read by host_reward_kernel / host_columns_kernel
  size_t stage_state = 0, stage_resid = 0, stage_events = 0;  // offsets (in floats) inside the stage; 0 = not mirrored
  bool stage_outputs_valid = false;  // the stage holds the state / remainders / events of the step that ran last
  // resident small-batch stepping (opt-in: MBT_RESIDENT_STEP=1; step_kernel.hpp: resident_step_kernel)
  ResidentKernel resident_kernel = nullptr;   // nullptr: this configuration has no resident form (or the mode is off)
  bool resident_active = false;               // a resident kernel is (or may still be) running on the stream
  bool resident_vram = false;                 // mailbox and action stage are device memory the host writes through the BAR
  mbt::ResidentMailbox* mailbox_host = nullptr;  // where the HOST writes the mailbox line ...
  mbt::ResidentMailbox* mailbox_dev = nullptr;   // ... and the same line as the device addresses it
  float* resident_action_host = nullptr;      // (n_pad, A) action stage of the resident kernel, host view / device view
  float* resident_action_dev = nullptr;
  void* resident_vram_block = nullptr;        // the fine-grained device allocation behind both (nullptr: they live in host memory)
  void* resident_host_block = nullptr;        // ... or the pinned host allocation
  size_t stage_exit = 0;                      // offset (in floats) of the word in the stage where a kernel says it left before a step
  bool action_in_resident_stage = false;      // the newest actions sit in resident_action_dev (file_staged_action)
  uint32_t resident_generation = 0;           // resident kernels launched so far
  int resident_answer_ms = 200;               // how long a step waits for the resident kernel before it falls back to one launch per step (MBT_RESIDENT_ANSWER_MS; 0 in a test: at once)
  // graph-capturable stepping (mbt_env_device_clock_begin ... _end; step_kernel.hpp: captured_step_kernel): between the two calls the
  // clock - time, episode step, Philox step - lives in `clock_dev` and the host's copy above is stale
  StepKernel kernel_captured = nullptr;       // (a CapturedKernel under the two-argument type: kernel_table.hpp, as_captured_kernel)
  hipFunction_t jit_step_captured = nullptr;
  mbt::DeviceClock* clock_dev = nullptr;
  mbt::DeviceClock* clock_host = nullptr;     // pinned: what travels to and from clock_dev
  uint32_t* clock_counters = nullptr;         // the captured launches' arrival counters (captured_arrive)
  bool device_clock = false;                  // the mode is on
  bool clock_auto_reset = false;
  float* terminal_obs = nullptr;              // (n_pad, D), on demand: the observation of an episode's last step (MBT_CLOCK_TERMINAL_OBSERVATION)
  bool clock_terminal_obs = false;
  bool capture_open = false;                  // the last mbt_env_step_device_captured was recorded by a stream capture ...
  unsigned long long capture_id = 0;          // ... this one (hipStreamGetCaptureInfo)
  uint32_t capture_parity = 0;                // the clock slot the next launch of that capture reads
};

namespace {

// index of the buffer the next launch writes the state into (see mbt_env::state)
inline int next_state(const mbt_env* e) { return e->ping_pong ? (e->cur ^ 1) : e->cur; }

// What the float64 code paths compute with (the precise_state tier; RewardFunction / process evaluation for host callers):
// the constructor arguments as the reference holds them.
void fill_precise_params(const mbt_config& c, double mid_dt, double arr_dt, double imp_dt, mbt::PreciseParams& X) {
  const int mk = c.midprice_kind;
  std::memset(&X, 0, sizeof X);
  X.mid_kind = mk;
  X.mu = c.drift;
  X.sigma = c.volatility;
  X.mid_dt = mid_dt;
  X.sqrt_mid_dt = std::sqrt(mid_dt);
  X.mu_dt = c.drift * mid_dt;                         // MID:63: self.drift * self.step_size
  X.sigma_sqrt_dt = c.volatility * std::sqrt(mid_dt);  // MID:64, MID:143: self.volatility * sqrt(self.step_size)
  X.mid_add = c.mid_coef_add;
  X.mid_mul = c.mid_coef_mul;
  X.ou_speed = c.ou_speed;
  X.ou_level = c.ou_level;
  X.jump_size = c.jump_size;
  X.hawkes_speed = c.hawkes_speed;
  X.hawkes_base_bid = c.intensity[0];
  X.hawkes_base_ask = c.intensity[1];
  X.hawkes_jump = c.hawkes_jump;
  X.arr_dt = arr_dt;
  X.half_spread = c.market_half_spread;
  X.q_max = c.max_inventory;
  X.c_max = c.max_cash;
  X.phi = c.phi;
  X.alpha = c.alpha;
  X.exponent = c.inventory_exponent;
  X.risk_aversion = c.risk_aversion;
  X.reward_scale = c.reward_scale;
  X.temp_coef = c.temporary_impact;
  X.impact_exponent = c.impact_exponent;
  X.perm_coef = c.permanent_impact;
  X.trans_coef = c.transient_impact;
  X.resilience = c.resilience;
  X.kernel_coef = c.kernel_coefficient;
  X.impact_dt = imp_dt;
  X.speed_dt = mid_dt;
}

void fill_static_params(mbt_env* e) {
  const mbt_config& c = e->cfg;
  mbt::StepParams& P = e->params;
  std::memset(&P, 0, sizeof P);
  P.n = e->n;
  P.n_pairs = e->n_pairs;
  P.pair_offset = c.trajectory_offset >> 1;
  P.mid_dt_f64 = e->mid_dt;
  for (int j = 0; j < 8; ++j) {
    P.user_fill_p[j] = e->user_fill_p[j];
    P.user_reward_p[j] = e->user_reward_p[j];
    P.user_arrival_p[j] = e->user_arrival_p[j];
    P.user_mid_p[j] = e->user_mid_p[j];
    P.user_state_p[j] = e->user_state_p[j];
  }
  P.dt = static_cast<float>(e->dt);
  // the midprice model as coefficients of midprice_increment() (step_kernel.hpp)
  const int mk = c.midprice_kind;
  const bool sde = mk == MBT_MID_LINEAR_SDE;  // the family itself: every coefficient comes from the caller
  // (MBT_MID_USER: the run-time compiled kernel replaces the increment; the coefficients below are then unused)
  const bool ou = mk == MBT_MID_OU || mk == MBT_MID_OU_JUMP, jump = mk == MBT_MID_BROWNIAN_JUMP || mk == MBT_MID_OU_JUMP;
  P.mid_add = sde ? static_cast<float>(c.mid_coef_add) : (mk == MBT_MID_GBM || mk == MBT_MID_CONSTANT) ? 0.0f : 1.0f;
  P.mid_mul = sde ? static_cast<float>(c.mid_coef_mul) : mk == MBT_MID_GBM ? 1.0f : 0.0f;
  P.drift_dt = (ou || mk == MBT_MID_CONSTANT) ? 0.0f : static_cast<float>(c.drift * e->mid_dt);
  P.vol_sqrt_dt = mk == MBT_MID_CONSTANT ? 0.0f : static_cast<float>(c.volatility * std::sqrt(e->mid_dt));
  P.ou_speed = (ou || sde) ? static_cast<float>(c.ou_speed) : 0.0f;
  P.ou_level = (ou || sde) ? static_cast<float>(c.ou_level) : 0.0f;
  P.jump_size = (jump || sde) ? static_cast<float>(c.jump_size) : 0.0f;
  if (c.arrival_kind == MBT_ARR_POISSON_NONLINEAR) {  // ARR:83
    P.arr_thr_bid = round_up_f32(1.0 - std::exp(-c.intensity[0] * e->arr_dt));
    P.arr_thr_ask = round_up_f32(1.0 - std::exp(-c.intensity[1] * e->arr_dt));
  } else {  // ARR:56
    P.arr_thr_bid = round_up_f32(c.intensity[0] * e->arr_dt);
    P.arr_thr_ask = round_up_f32(c.intensity[1] * e->arr_dt);
  }
  {  // the Philox uniforms are (w >> 8) * 2^-24: u < thr <=> (w >> 8) < K = ceil(thr * 2^24) (exact in double) <=> w < (K << 8)
    const auto as_count = [](float thr) -> uint32_t {
      const double x = std::ceil(static_cast<double>(thr) * 16777216.0);
      return x <= 0.0 ? 0u : (x >= 16777216.0 ? 16777216u : static_cast<uint32_t>(x));
    };
    const uint32_t k_bid = as_count(P.arr_thr_bid), k_ask = as_count(P.arr_thr_ask);
    P.arr_always_bid = k_bid >= 16777216u ? 1 : 0;
    P.arr_always_ask = k_ask >= 16777216u ? 1 : 0;
    P.arr_thr_w_bid = P.arr_always_bid ? 0xFFFFFFFFu : (k_bid << 8);
    P.arr_thr_w_ask = P.arr_always_ask ? 0xFFFFFFFFu : (k_ask << 8);
  }
  P.arr_dt_f64 = e->arr_dt;
  P.arr_dt = static_cast<float>(e->arr_dt);
  P.hawkes_base_bid = static_cast<float>(c.intensity[0]);
  P.hawkes_base_ask = static_cast<float>(c.intensity[1]);
  P.hawkes_speed = static_cast<float>(c.hawkes_speed);
  P.hawkes_jump = static_cast<float>(c.hawkes_jump);
  P.kappa_log2e_neg = static_cast<float>(-c.fill_exponent * 1.4426950408889634);
  P.kappa_f64 = c.fill_exponent;
  P.fill_depth_per_log2 = static_cast<float>(-0.6931471805599453 / c.fill_exponent);
  P.fill_band_abs = static_cast<float>(2e-7 / c.fill_exponent);
  for (int side = 0; side < 2; ++side) {  // the state columns are float32; the exact re-decision keeps the float64 depths
    P.exo_depth[side] = static_cast<float>(c.exogenous_depth[side]);
    P.exo_depth_f64[side] = c.exogenous_depth[side];
  }
  P.exo_base = static_cast<float>(c.base_fill_probability);
  P.exo_base_f64 = c.base_fill_probability;
  P.half_spread = static_cast<float>(c.market_half_spread);
  P.q_max = static_cast<float>(c.max_inventory);
  P.c_max = static_cast<float>(c.max_cash);
  P.reward_kind = c.reward_kind;
  P.alpha_running = c.reward_kind == MBT_REW_RUNNING_PENALTY ? static_cast<float>(c.alpha) : 0.0f;
  P.alpha_cjmm = c.reward_kind == MBT_REW_CJ_MM ? static_cast<float>(c.alpha) : 0.0f;
  P.quad_new = static_cast<float>(e->dt * c.phi + (c.reward_kind == MBT_REW_CJ_MM ? c.alpha : 0.0));
  P.quad_init = 0.0f;  // needs the episode length: reset()
  P.exponent_is_two = c.inventory_exponent == 2.0 ? 1 : 0;
  P.phi = static_cast<float>(c.phi);
  P.alpha = static_cast<float>(c.alpha);
  P.exponent = static_cast<float>(c.inventory_exponent);
  P.risk_aversion = static_cast<float>(c.risk_aversion);
  P.reward_scale = static_cast<float>(c.reward_scale);
  P.impact_kind = c.impact_kind;
  P.impact_exponent_is_one = c.impact_exponent == 1.0 ? 1 : 0;
  P.speed_dt = static_cast<float>(e->mid_dt);
  P.impact_dt = static_cast<float>(e->imp_dt);
  P.temp_coef = static_cast<float>(c.temporary_impact);
  P.impact_exponent = static_cast<float>(c.impact_exponent);
  P.perm_coef = static_cast<float>(c.permanent_impact);
  P.trans_coef = static_cast<float>(c.transient_impact);
  P.resilience = static_cast<float>(c.resilience);
  P.kernel_coef = static_cast<float>(c.kernel_coefficient);
  fill_precise_params(c, e->mid_dt, e->arr_dt, e->imp_dt, P.X);
  P.norm_act = c.normalise_action;
  P.norm_obs = c.normalise_observation;
  for (int j = 0; j < 4; ++j) {
    P.act_lo[j] = c.act_lo[j];
    P.act_grad[j] = (c.act_hi[j] - c.act_lo[j]) / 2.0f;  // float32 arithmetic, as TE:193 does on the Box bounds
  }
  for (int j = 0; j < 8; ++j) {
    P.obs_lo[j] = c.obs_lo[j];
    P.obs_grad[j] = (c.obs_hi[j] - c.obs_lo[j]) / 2.0f;  // TE:185
  }
}

// Two regimes, chosen by how many bytes ONE launch touches.  While that is about the size of the 256 MB Infinity Cache or
// less, the state the next step reads is still cached: default-policy loads, full occupancy (8 workgroups per CU hide the
// generator behind the loads).  Measured (profiles/r02_regimes.json): at 184 MB (AS, 2^22 lanes) and 252 MB (Hawkes + OU,
// 2^22 lanes) default loads win (26.9 vs 28.2 us, 36.7 vs 45.1 us), at 369 MB (AS, 2^23 lanes) streaming loads win
// (53.4 vs 63.5 us); the switch sits at 320 MB.  Beyond that every byte crosses HBM and nothing is re-used: the step kernel's STREAM
// instantiation is used, whose loads carry the non-temporal bit (they do not displace lines on the way in) and occupancy is capped at 5 workgroups per CU through a
// dynamic LDS allocation - fewer, longer-lived streams per channel (129.9 -> 124.3 us at 2^24 lanes for occupancy alone,
// profiles/r01_microbench.txt; the non-temporal loads: 128.3 -> 118.7 us on the copy kernel).  MBT_STREAM_LOADS = 0 / 1 and
// MBT_STEP_DYNAMIC_LDS = bytes override the choice (measurement knobs).
void tune_for_size(mbt_env* e) {
  const size_t bytes_per_launch = size_t(e->n_pad) * 4u * (2u * e->dim + e->act_dim + 1u + 2u * e->res);  // moved (read + written)
  e->ping_pong = bytes_per_launch < (size_t(80) << 20);  // see mbt_env::state
  if (const char* v = std::getenv("MBT_PING_PONG_STATE")) e->ping_pong = std::atoi(v) != 0;
  // (round 4: rows that are NOT 16 bytes wide lose with non-temporal loads far beyond 320 MB - Hawkes + OU at 2^22 lanes: 252 MB
  // float32 43.5 vs 36.6 us, 386 MB precise_state 69.5 vs 56.5 us - and win only around 1 GB, 2^24 lanes: 165.8 vs 177.2 us)
  // (round 5, state updated in place: what decides is the set of DISTINCT lines a launch touches, which then stays in the Infinity
  // Cache until the next launch or does not - AS at 2^23 lanes, 235 MB distinct / 369 MB moved: default-policy loads 51.5 us,
  // non-temporal 53-55; at 2^24, 470 MB distinct: 124-132 vs 114.5.  16-byte rows switch at 300 MB distinct; the other row widths
  // keep their measured 640 MB moved.)
  const size_t distinct_bytes = size_t(e->n_pad) * 4u * ((e->ping_pong ? 2u : 1u) * e->dim + e->act_dim + 1u + e->res);
  const bool hbm_resident = e->dim == 4 ? distinct_bytes > (size_t(300) << 20) : bytes_per_launch > (size_t(640) << 20);
  e->stream_loads = hbm_resident;
  e->step_dynamic_lds = (hbm_resident && !e->speed && e->dim == 4) ? 32u * 1024u : 0u;  // AS 2^24: 115.7 -> 114.2 us; Hawkes (D = 6) loses with it
  // Inside the cache the same cap pays for ONE kernel: the lightest one (Brownian midprice, Poisson arrivals, limit orders,
  // plain PnL - BASELINE configs[1], the benchmark workload, and with normalised spaces the reference's default
  // environment), from 2^20 lanes up: 6.76 / 6.85 / 6.84 -> 6.70 / 6.62 / 6.62 us at 2^20 (three alternating runs of bench.py),
  // 14.53 -> 14.25 at 2^21, 26.74 -> 26.59 at 2^22; normalised: 9.37 / 9.37 / 9.34 -> 8.97 / 9.01 / 8.90 us at 2^20; at 2^18
  // it costs (3.52 -> 3.69 us), and the kernels with more arithmetic per lane need their waves (CJP 6.7 -> 7.6 us, speed + impact
  // state 8.0 -> 10.0: profiles/r02_occupancy_cap_all_configs.txt).
  const mbt_config& c = e->cfg;
  const bool lightest = !e->speed && e->dim == 4 && !c.precise_state && c.midprice_kind == MBT_MID_BROWNIAN &&
                        (c.arrival_kind == MBT_ARR_POISSON || c.arrival_kind == MBT_ARR_POISSON_NONLINEAR) && c.fill_kind == MBT_FILL_EXPONENTIAL &&
                        c.dynamics_kind == MBT_DYN_LIMIT && c.reward_kind == MBT_REW_PNL && c.noise_mode == MBT_NOISE_PHILOX;
  if (!hbm_resident && lightest && bytes_per_launch >= (size_t(40) << 20)) e->step_dynamic_lds = 32u * 1024u;
  if (const char* v = std::getenv("MBT_STREAM_LOADS")) e->stream_loads = std::atoi(v) != 0;
  if (const char* v = std::getenv("MBT_STEP_DYNAMIC_LDS")) e->step_dynamic_lds = static_cast<uint32_t>(std::strtoul(v, nullptr, 10));
}

// What the reward functions capture at reset (RW:72-74, RW:111-113): they measure the episode against THEIR OWN
// terminal_time constructor argument, which need not be the environment's.
void fill_episode_params(mbt_env* e) {
  const mbt_config& c = e->cfg;
  mbt::StepParams& P = e->params;
  const double reward_T = c.reward_terminal_time > 0.0 ? c.reward_terminal_time : c.terminal_time;
  const double length = reward_T - e->start_time;
  P.q_init_scalar = static_cast<float>(c.initial_inventory);
  P.dt_over_episode = static_cast<float>(e->dt / length);  // RW:106, RW:113
  P.quad_init = c.reward_kind == MBT_REW_CJ_MM ? static_cast<float>(c.alpha * e->dt / length) : 0.0f;
  P.episode_length = static_cast<float>(length);  // RW:67, RW:74
  P.X.episode_length = length;
}

void key_from_seed(mbt_env* e) {
  e->params.key0 = static_cast<uint32_t>(e->seed);
  e->params.key1 = static_cast<uint32_t>(e->seed >> 32);
}

// Zero-filled device buffer.  The fill is ordered on the environment's own stream: that stream is non-blocking, so a
// fill on the null stream would not be ordered against the kernels that use the buffer next.
template <typename T>
int dev_alloc(T** p, size_t count, hipStream_t stream) {
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
  HIP_TRY(hipMemsetAsync(*p, 0, count * sizeof(T), stream));
  return MBT_OK;
}

// Opens the launch gate: everything enqueued behind it starts to run.
void gate_open(mbt_env* e) {
  if (!e->gate_closed) return;
  __atomic_store_n(e->h_gate, e->gate_seq, __ATOMIC_RELEASE);
  e->gate_closed = false;
}

// The newest actions were handed over through the stage (mbt_env_step_host, small batches); anything that reads the
// library's own action buffer afterwards - step_device(NULL), an action-repeat rollout, a consumer of mbt_env_action_ptr -
// finds them there.
int resident_stop(mbt_env* e);
int file_staged_action(mbt_env* e) {
  if (!e->action_in_stage) return MBT_OK;
  const int rc_stop = resident_stop(e);  // (the copy below is queued on the stream a resident kernel would hold)
  if (rc_stop != MBT_OK) return rc_stop;
  if (e->action_in_resident_stage) {  // (the resident kernel's own stage; a device address in either placement)
    HIP_TRY(hipMemcpyAsync(e->action, e->resident_action_dev, size_t(e->n) * e->act_dim * sizeof(float), e->resident_vram ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(e->action, e->h_stage + e->stage_action, size_t(e->n) * e->act_dim * sizeof(float), hipMemcpyHostToDevice, e->stream));
  }
  HIP_TRY(hipStreamSynchronize(e->stream));  // the stage is overwritten by the next host step
  e->action_in_stage = e->action_in_resident_stage = false;
  return MBT_OK;
}

// ---- resident small-batch stepping (opt-in) ------------------------------------------------------------------------------------
// Tells a resident kernel to leave and waits until it has (it polls its mailbox every microsecond or so): everything that is not
// the next mbt_env_step_host starts from an idle stream and a state the host's bookkeeping describes.
int resident_stop(mbt_env* e) {
  if (!e->resident_active) return MBT_OK;
  e->mailbox_host->seq = mbt::kResidentExit;
  _mm_sfence();
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->resident_active = false;
  return MBT_OK;
}
#define RESIDENT_STOP(e)                  \
  do {                                    \
    const int rc_stop_ = resident_stop(e); \
    if (rc_stop_ != MBT_OK) return rc_stop_; \
  } while (0)
// ... and for every entry point that reads or advances the HOST's clock (time, episode step, Philox step), or changes what a step
// launch is given: refused while the clock lives on the device (mbt_env_device_clock_begin) - a graph captured in that mode holds
// the launches' arguments, and the host's copy of the clock is stale until mbt_env_device_clock_end brings it back.
#define HOST_CLOCK(e)                                                                                                              \
  do {                                                                                                                             \
    RESIDENT_STOP(e);                                                                                                              \
    if ((e)->device_clock)                                                                                                         \
      return fail(MBT_ERR_STATE, "%s: the environment's clock is on the device (mbt_env_device_clock_begin): call "               \
                                 "mbt_env_device_clock_end first", __func__);                                                      \
  } while (0)

// Before the host overwrites a region of the mapped callback block (h_callback_in): has every kernel that reads it in place finished?
int settle_callback_block(mbt_env* e, bool& busy) {
  if (busy) HIP_TRY(hipStreamSynchronize(e->stream));
  e->callback_inputs_busy = e->callback_scratch_busy = false;  // (one in-order stream: a wait settles both)
  busy = false;
  return MBT_OK;
}

#define SETTLE_CALLBACK_BLOCK(e, region)                            \
  do {                                                              \
    if ((e)->region) {                                              \
      const int rc_busy_ = settle_callback_block((e), (e)->region); \
      if (rc_busy_ != MBT_OK) return rc_busy_;                      \
    }                                                               \
  } while (0)

// Can the host write this device allocation (fine-grained memory through the PCIe BAR)?  Decided WITHOUT touching it - rounds 4-5
// wrote a probe word under temporary SIGSEGV / SIGBUS handlers, which are process-wide: a fault on any other thread during the
// probe would have jumped into this thread's frame.  Two facts answer the question instead: the device says its whole memory is
// visible through the BAR (hipDeviceAttributeIsLargeBar), and the kernel's own list of this process's mappings shows the block's
// address range mapped readable and writable (the runtime maps device memory into the host's address space exactly when the BAR
// reaches it) - an address that is not mapped, or mapped without access (a reservation), would fault.
bool host_can_write(const void* device_ptr, size_t bytes, int device) {
  int large_bar = 0;
  if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess || large_bar == 0) {
    (void)hipGetLastError();
    return false;
  }
  std::FILE* maps = std::fopen("/proc/self/maps", "r");
  if (maps == nullptr) return false;
  const uintptr_t lo = reinterpret_cast<uintptr_t>(device_ptr), hi = lo + bytes;
  uintptr_t covered = lo;  // [lo, covered) is known to be mapped read-write; the list is sorted by address
  char line[512];
  while (covered < hi && std::fgets(line, sizeof line, maps) != nullptr) {
    unsigned long long a = 0, b = 0;
    char perms[8] = {0};
    if (std::sscanf(line, "%llx-%llx %7s", &a, &b, perms) != 3) continue;
    if (b <= covered) continue;
    if (a > covered) break;  // a hole in front of the next mapping
    if (perms[0] != 'r' || perms[1] != 'w') break;
    covered = static_cast<uintptr_t>(b);
  }
  std::fclose(maps);
  return covered >= hi;
}

// Mailbox + action stage of the resident kernel: device memory the host can write if the platform has it (the kernel then polls and
// reads LOCAL memory: 9.6 instead of 12.1 us per step at N = 1000, profiles/r05_resident_step.txt), pinned device-mapped host memory
// otherwise.  MBT_RESIDENT_VRAM=0 forces the latter (measurement knob).
int resident_allocate(mbt_env* e) {
  const size_t action_bytes = size_t(e->n_pad) * e->act_dim * sizeof(float), bytes = 4096 + ((action_bytes + 4095) / 4096) * 4096;
  const char* knob = std::getenv("MBT_RESIDENT_VRAM");
  if (knob == nullptr || std::atoi(knob) != 0) {
    void* block = nullptr;
    if (hipExtMallocWithFlags(&block, bytes, hipDeviceMallocFinegrained) == hipSuccess && block != nullptr) {
      // (on the environment's own stream: a fill on the null stream and a device-wide wait would block behind ANOTHER environment's resident kernel)
      if (hipMemsetAsync(block, 0, bytes, e->stream) == hipSuccess && hipStreamSynchronize(e->stream) == hipSuccess && host_can_write(block, bytes, e->cfg.device)) {
        e->resident_vram_block = block;
        e->resident_vram = true;
        e->mailbox_host = e->mailbox_dev = static_cast<mbt::ResidentMailbox*>(block);  // (one address space: the same pointer on both sides)
        e->resident_action_host = e->resident_action_dev = reinterpret_cast<float*>(static_cast<char*>(block) + 4096);
        return MBT_OK;
      }
      (void)hipFree(block);
    }
    (void)hipGetLastError();
  }
  void* host = nullptr;
  HIP_TRY(hipHostMalloc(&host, bytes, hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(host, 0, bytes);
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, host, 0) != hipSuccess) {
    (void)hipHostFree(host);
    return fail(MBT_ERR_HIP, "the resident kernel's mailbox could not be mapped for the device");
  }
  e->resident_host_block = host;
  e->mailbox_host = static_cast<mbt::ResidentMailbox*>(host);
  e->mailbox_dev = static_cast<mbt::ResidentMailbox*>(dev);
  e->resident_action_host = reinterpret_cast<float*>(static_cast<char*>(host) + 4096);
  e->resident_action_dev = reinterpret_cast<float*>(static_cast<char*>(dev) + 4096);
  return MBT_OK;
}

// mirror: the launch also writes what env.step() returns into host memory - the stage, or the caller's own arrays when they
// are device-mapped (mirror_obs / mirror_reward) - and raises its flag (small-batch host API)
int launch_step(mbt_env* e, const float* action_dev, int32_t* done, bool mirror = false, float* mirror_obs = nullptr, float* mirror_reward = nullptr) {
  if (!e->was_reset) return fail(MBT_ERR_STATE, "step() before reset()");
  if (action_dev == nullptr && e->action_in_stage) {
    const int rc_file = file_staged_action(e);
    if (rc_file != MBT_OK) return rc_file;
  }
  if ((e->host_mask & mbt::kHostFill) && !e->host_fill_ready)
    return fail(MBT_ERR_STATE, "host-callback fill model: mbt_env_set_host_fill_probabilities must precede every step()");
  if ((e->host_mask & mbt::kHostImpact) && !e->host_fill_ready)
    return fail(MBT_ERR_STATE, "host-callback price impact model: mbt_env_set_host_impacts must precede every step()");
  if ((e->host_mask & mbt::kHostArrival) && !e->host_arrivals_ready)
    return fail(MBT_ERR_STATE, "host-callback arrival model: mbt_env_set_host_arrivals must precede every step()");
  if (e->host_reward_pending)
    return fail(MBT_ERR_STATE, "host-callback reward: mbt_env_set_host_rewards must follow every step() before the next one");
  const bool inject = e->cfg.noise_mode == MBT_NOISE_INJECTED;
  if (inject && !e->noise_ready) return fail(MBT_ERR_STATE, "injected-noise mode: set_noise() must precede every step()");
  if (inject && e->user_draws && !e->user_noise_ready) return fail(MBT_ERR_STATE, "injected-noise mode: the user processes' extra normals (mbt_env_set_user_noise_host) must precede every step()");
  // host clock: t += dt (TE:216); done = t >= T - dt/2 (TE:218-220).  Committed only once the launch is known to be queued.
  const double t_next = e->time + e->dt;
  const bool terminal = t_next >= e->cfg.terminal_time - e->dt / 2;
  mbt::StepParams& P = e->params;
  P.philox_step = e->philox_step;
  P.is_terminal = terminal ? 1 : 0;
  P.t_next = static_cast<float>(t_next);
  P.t_now = e->time;
  P.t_next_f64 = t_next;

  mbt::StepBuffers B;
  std::memset(&B, 0, sizeof B);
  B.state_in = e->state[e->cur];
  B.state_out = e->state[next_state(e)];
  B.action = action_dev != nullptr ? action_dev : e->action;
  B.host_fill_p = e->host_fill_p;
  B.host_arrivals = e->host_arrivals;
  e->stage_outputs_valid = false;
  if (mirror) {
    if (e->stage_state != 0) {  // host-callback plugins, order-book dynamics: rows, remainders and events for the caller's update() / calculate()
      B.host_state = e->d_stage + e->stage_state;
      B.host_resid = e->res != 0 ? reinterpret_cast<int32_t*>(e->d_stage + e->stage_resid) : nullptr;
      B.host_events = reinterpret_cast<uint8_t*>(e->d_stage + e->stage_events);
    }
    B.host_obs = mirror_obs != nullptr ? mirror_obs : e->d_stage + e->stage_obs;
    B.host_reward = mirror_reward != nullptr ? mirror_reward : e->d_stage + e->stage_reward;
    B.done_counter = e->done_counter;
    B.host_flag = reinterpret_cast<uint32_t*>(e->d_stage + e->stage_flag);
    B.flag_value = ++e->flag_seq;
  }
  B.reward = e->reward;
  B.obs = e->cfg.normalise_observation ? e->obs : nullptr;
  B.u_arr = e->u_arr;
  B.u_fill = e->u_fill;
  B.z = e->z;
  B.z_user = e->z_user;
  B.q_init = e->q_init_per_lane ? e->q_init : nullptr;
  B.resid = e->resid;
  B.events = e->record_events ? e->events : nullptr;
  B.lane_returns = e->track_returns ? e->lane_returns : nullptr;
  B.wave_sums = e->wave_sums;
  B.clip_count = e->clip_count;
  if (e->jit_step != nullptr) {
    void* args[] = {&B, &P};
    HIP_TRY(hipModuleLaunchKernel(mirror ? e->jit_step_mirror : e->jit_step, e->n_blocks, 1, 1, mbt::kBlockThreads, 1, 1, e->step_dynamic_lds, e->stream, args, nullptr));
  } else {
    hipLaunchKernelGGL(mirror ? e->kernel_mirror : e->kernel, dim3(e->n_blocks), dim3(mbt::kBlockThreads), e->step_dynamic_lds, e->stream, B, P);
    HIP_TRY(hipGetLastError());
  }
  e->time = t_next;
  e->cur = next_state(e);
  e->philox_step += 1;
  e->episode_step += 1;
  e->noise_ready = false;
  e->user_noise_ready = false;
  e->host_fill_ready = e->host_arrivals_ready = false;
  e->host_reward_pending = (e->host_mask & mbt::kHostReward) != 0;
  if (e->h_callback_in != nullptr && (e->host_mask & (mbt::kHostFill | mbt::kHostArrival | mbt::kHostImpact))) e->callback_inputs_busy = true;
  if (done != nullptr) *done = terminal ? 1 : 0;
  return MBT_OK;
}

// Starts a resident kernel whose first step carries the sequence number `first_seq` (step_kernel.hpp: resident_step_kernel); the
// mailbox must not hold that number yet.
int resident_launch(mbt_env* e, uint32_t first_seq) {
  mbt::StepParams P = e->params;
  P.philox_step = e->philox_step;
  mbt::StepBuffers B;
  std::memset(&B, 0, sizeof B);
  B.state_in = e->state[e->cur];
  B.state_out = e->state[next_state(e)];
  B.action = e->resident_action_dev;
  B.done_counter = e->done_counter;
  B.host_flag = reinterpret_cast<uint32_t*>(e->d_stage + e->stage_flag);
  B.reward = e->reward;
  B.obs = e->cfg.normalise_observation ? e->obs : nullptr;
  B.q_init = e->q_init_per_lane ? e->q_init : nullptr;
  B.resid = e->resid;
  B.events = e->record_events ? e->events : nullptr;
  B.lane_returns = e->track_returns ? e->lane_returns : nullptr;
  B.wave_sums = e->wave_sums;
  B.clip_count = e->clip_count;
  mbt::ResidentParams R;
  std::memset(&R, 0, sizeof R);
  R.mailbox = e->mailbox_dev;
  R.host_exit = reinterpret_cast<uint32_t*>(e->d_stage + e->stage_exit);
  R.control = e->done_counter + 16;  // (the second line of the counter block)
  R.generation = ++e->resident_generation;
  R.first_seq = first_seq;
  R.n_tiles = e->n_blocks;
  R.t_start = e->time;
  R.dt_f64 = e->dt;
  R.terminal_time = e->cfg.terminal_time;
  R.idle_ticks = 200000ull;        // 2 ms of the 100 MHz wall clock without a doorbell: leave (the next step starts another kernel)
  R.life_ticks = 3000000000ull;    // 30 s in any case
  if (const char* v = std::getenv("MBT_RESIDENT_IDLE_US")) R.idle_ticks = std::strtoull(v, nullptr, 10) * 100ull;
  R.ping_pong = e->ping_pong ? 1 : 0;
  const uint32_t groups = e->n_blocks < 4u ? e->n_blocks : 4u;  // (one or two tiles per workgroup: see kResidentMaxTiles)
  hipLaunchKernelGGL(e->resident_kernel, dim3(groups), dim3(mbt::kBlockThreads), 0, e->stream, B, P, R);
  HIP_TRY(hipGetLastError());
  e->resident_active = true;
  return MBT_OK;
}

// One env.step() of a small batch through the resident kernel: actions into its stage, the mailbox line (where the outputs go + the
// step's sequence number) behind them, spin on the completion flag.  The host's bookkeeping advances exactly as in launch_step.
int resident_step(mbt_env* e, const float* action_host, float* obs_host, float* reward_host, int32_t* done) {
  if (!e->was_reset) return fail(MBT_ERR_STATE, "step() before reset()");
  const size_t n_obs = size_t(e->n) * e->dim;
  float* direct_obs = static_cast<float*>(device_alias(obs_host, n_obs * sizeof(float)));
  float* direct_rew = static_cast<float*>(device_alias(reward_host, size_t(e->n) * sizeof(float)));
  if (direct_obs == nullptr || direct_rew == nullptr || reinterpret_cast<uintptr_t>(direct_obs) % 16 != 0) direct_obs = direct_rew = nullptr;
  const double t_next = e->time + e->dt;
  const bool terminal = t_next >= e->cfg.terminal_time - e->dt / 2;
  uint32_t seq = ++e->flag_seq;
  if (seq == mbt::kResidentExit) seq = ++e->flag_seq;  // (the one value that means "leave": skipped when the counter wraps, after 4e9 steps)
  if (!e->resident_active) {
    e->mailbox_host->seq = (seq - 1u == mbt::kResidentExit) ? seq - 2u : seq - 1u;  // (idle: whatever an earlier kernel was told is gone)
    _mm_sfence();
    // ... and so is what an earlier kernel said on leaving: told to leave while it waited for THIS sequence number, it wrote that
    // number - which below would read as "the kernel left before your step" and start a second one
    __atomic_store_n(reinterpret_cast<uint32_t*>(e->h_stage + e->stage_exit), 0u, __ATOMIC_RELEASE);
    const int rc = resident_launch(e, seq);
    if (rc != MBT_OK) return rc;
  }
  std::memcpy(e->resident_action_host, action_host, size_t(e->n) * e->act_dim * sizeof(float));
  _mm_sfence();  // the actions are on their way before the line that announces them (device memory is write-combining on the host side)
  // where the outputs go first, the sequence number that announces them behind a fence of its own: a write-combining buffer is
  // USUALLY flushed as one 64-byte burst, but nothing promises it, and a kernel that saw the new number with the old pointers would
  // mirror into arrays the caller has already been handed
  e->mailbox_host->host_obs = reinterpret_cast<uint64_t>(direct_obs != nullptr ? direct_obs : e->d_stage + e->stage_obs);
  e->mailbox_host->host_reward = reinterpret_cast<uint64_t>(direct_rew != nullptr ? direct_rew : e->d_stage + e->stage_reward);
  _mm_sfence();
  e->mailbox_host->seq = seq;
  _mm_sfence();
  const uint32_t* flag = reinterpret_cast<const uint32_t*>(e->h_stage + e->stage_flag);
  uint32_t* left = reinterpret_cast<uint32_t*>(e->h_stage + e->stage_exit);
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t spins = 0;
  while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
    if (__atomic_load_n(left, __ATOMIC_ACQUIRE) == seq) {  // the kernel left (idle, or its lifetime) before this step: another one takes it
      __atomic_store_n(left, 0u, __ATOMIC_RELEASE);
      const int rc = resident_launch(e, seq);
      if (rc != MBT_OK) return rc;
    }
    if (((++spins & 1023u) == 0u || e->resident_answer_ms == 0) && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(e->resident_answer_ms)) {
      e->mailbox_host->seq = mbt::kResidentExit;
      _mm_sfence();
      HIP_TRY(hipStreamSynchronize(e->stream));
      e->resident_active = false;
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
        // The kernel has left without taking the step - it takes a step whole or not at all (resident_body: the control word changes
        // only once every workgroup has finished the previous step), and its flag would say so - so the device state is the one the
        // host's bookkeeping describes.  A safety net, e.g. for a device that suspends a kernel which never ends by itself in favour
        // of other processes' queues: the mode is a latency optimisation, not a contract, and the environment goes back to one
        // launch per step - the same arithmetic, the same results - for the rest of its life.  (The soak that prompted it - eight
        // processes in resident mode, profiles/r05_soak.txt - turned out to hang on a partial step instead; with that fixed it no
        // longer comes here.)
        std::fprintf(stderr, "mbt_gym_amd: the resident step kernel did not answer step %u within %d ms: "
                             "this environment steps through one launch per step from here on\n", seq, e->resident_answer_ms);
        e->resident_kernel = nullptr;
        e->resident_vram = false;  // (mbt_env_step_host: the stage in host memory, as if the mode had never been on)
        return mbt_env_step_host(e, action_host, obs_host, reward_host, done);
      }
      break;
    }
  }
  e->time = t_next;
  e->cur = next_state(e);
  e->philox_step += 1;
  e->episode_step += 1;
  e->action_in_stage = e->action_in_resident_stage = true;
  if (terminal) {  // the kernel has left by itself (it computed the same flag); nothing is queued behind it
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->resident_active = false;
  }
  if (done != nullptr) *done = terminal ? 1 : 0;
  if (direct_obs == nullptr) {
    if (obs_host != nullptr) std::memcpy(obs_host, e->h_stage + e->stage_obs, n_obs * sizeof(float));
    if (reward_host != nullptr) std::memcpy(reward_host, e->h_stage + e->stage_reward, size_t(e->n) * sizeof(float));
  }
  return MBT_OK;
}

// ---- learned policies (policy_mlp.hpp) --------------------------------------------------------------------------------
LearnedRolloutKernel pick_learned_rollout(const mbt_config& c) {
  return mbt_table::pick_rollout_learned(arrival_family(c), c.dynamics_kind == MBT_DYN_LIMIT_AND_MARKET, c.midprice_kind == MBT_MID_BROWNIAN && c.reward_kind == MBT_REW_PNL);
}

// Device image of a learned policy: [w1: 4 x 64 half4 | w2: 8 x 64 half8 | w3: 2 x 64 half8 | b2: 64 f32 | b3: 16 f32 | lin_w: 32 f32 | lin_b: 4 f32]
constexpr size_t kLearnedW1 = 0, kLearnedW2 = kLearnedW1 + 4 * 64 * 8, kLearnedW3 = kLearnedW2 + 16 * 64 * 8, kLearnedB2 = kLearnedW3 + 4 * 64 * 8,
                 kLearnedB3 = kLearnedB2 + 64 * 4, kLearnedLinW = kLearnedB3 + 16 * 4, kLearnedLinB = kLearnedLinW + 32 * 4,
                 kLearnedBytes = kLearnedLinB + 4 * 4;

// Validates a MBT_POLICY_LINEAR / MBT_POLICY_MLP descriptor, lays its weights out in MFMA A-operand order (fp16, rounded
// to nearest even here on the host) and uploads them when they differ from what the device holds.
// The fused learned rollout is instantiated for the float32 tiers of the built-in order-book models; the exogenous-depth
// fill model, precise_state and run-time compiled plugins take the policy as a kernel of its own in front of each step
// (policy_kernel: same rows, same exploration counters - launch_rollout falls back to that loop, results identical).
float* current_obs(mbt_env* e);
bool learned_rollout_is_fused(const mbt_env* e) { return !e->speed && !exogenous_fill(e->cfg) && !e->cfg.precise_state && e->jit_step == nullptr; }
// (the policy kernel walks 512-lane tiles whatever the environment's own tile is: speed dynamics pad to 1024)
uint32_t policy_blocks(const mbt_env* e) { return e->n_pad / mbt::kTileLanes; }

int prepare_learned_policy(mbt_env* e, const mbt_policy* policy, mbt::LearnedPolicyParams& LP) {
  const mbt_config& c = e->cfg;
  if (c.dynamics_kind == MBT_DYN_AT_THE_TOUCH) return fail(MBT_ERR_INVALID, "learned policies output real-valued actions (depths, market-order scores, a trading speed): not the binary actions of at-the-touch dynamics");
  const int D = e->dim, A = e->act_dim;
  if (policy->table == nullptr) return fail(MBT_ERR_INVALID, "a learned policy needs its weights (mbt_policy.table)");
  std::vector<char> image(kLearnedBytes, 0);
  std::memset(&LP, 0, sizeof LP);
  LP.obs_dim = D;
  LP.act_dim = A;
  const bool norm = c.normalise_action != 0;
  for (int j = 0; j < 4; ++j) {  // the action space the agent acts in (TE:243-255 when normalised)
    LP.act_lo[j] = j < A ? (norm ? -1.0f : c.act_lo[j]) : 0.0f;
    LP.act_hi[j] = j < A ? (norm ? 1.0f : c.act_hi[j]) : 0.0f;
  }
  LP.clip = policy->params[1] != 0.0 ? 1 : 0;
  for (int j = 0; j < 4; ++j) {
    LP.act_std[j] = j < A ? static_cast<float>(policy->params[2 + j]) : 0.0f;
    if (!(LP.act_std[j] >= 0.0f)) return fail(MBT_ERR_INVALID, "the exploration std of action %d is %g: must be >= 0", j, policy->params[2 + j]);
    if (LP.act_std[j] != 0.0f) LP.stochastic = 1;
  }
  const float* w = policy->table;
  if (policy->kind == MBT_POLICY_LINEAR) {
    if (policy->table_cols != static_cast<uint32_t>(A * D + A)) return fail(MBT_ERR_INVALID, "a linear policy holds A*D + A = %d floats (got %u)", A * D + A, policy->table_cols);
    LP.is_linear = 1;
    float* lw = reinterpret_cast<float*>(image.data() + kLearnedLinW);
    float* lb = reinterpret_cast<float*>(image.data() + kLearnedLinB);
    for (int a = 0; a < A; ++a) {
      for (int d = 0; d < D; ++d) lw[a * 8 + d] = w[a * D + d];
      lb[a] = w[A * D + a];
    }
  } else {
    const int H = static_cast<int>(policy->table_rows);
    if (H < 1 || H > mbt::kMlpHidden) return fail(MBT_ERR_INVALID, "MLP policies have a hidden width of 1..64 (got %d)", H);
    // The matrix cores take the observation row as fp16: 11 significant bits, nothing beyond 65504.  Normalised observations
    // ([-1, 1], TE:112-118) are resolved to 5e-4 or better; a raw midprice of 100 would be quantised to 0.0625 and raw cash
    // overflows - the actions would silently differ from the network's.  Refused instead (a linear policy is float32 and
    // takes any observation; a host loop with the network in float32 takes any network).
    if (!c.normalise_observation)
      for (int j = 0; j < D; ++j) {
        const float bound = std::fmax(std::fabs(c.obs_lo[j]), std::fabs(c.obs_hi[j]));
        if (!(bound <= mbt::kMlpMaxObservationBound))
          return fail(MBT_ERR_INVALID, "MLP policies read the observation as fp16: column %d is bounded by %g (> %g) - use normalise_observation_space=True "
                      "(or evaluate the network on the host)", j, bound, mbt::kMlpMaxObservationBound);
      }
    if (D + 1 > mbt::kMlpInPad) return fail(MBT_ERR_INVALID, "MLP policies take observations of at most 15 columns");
    const size_t floats = size_t(H) * D + H + size_t(H) * H + H + size_t(A) * H + A;
    if (policy->table_cols != floats) return fail(MBT_ERR_INVALID, "an MLP policy of width %d holds %zu floats for D = %d, A = %d (got %u)", H, floats, D, A, policy->table_cols);
    const int act = static_cast<int>(policy->params[0]);
    if (act != mbt::kActTanh && act != mbt::kActRelu) return fail(MBT_ERR_INVALID, "activation %d: 0 = tanh, 1 = relu", act);
    LP.activation = act;
    const float *W1 = w, *b1 = W1 + size_t(H) * D, *W2 = b1 + H, *b2 = W2 + size_t(H) * H, *W3 = b2 + H, *b3 = W3 + size_t(A) * H;
    auto w1p = [&](int m, int k) -> float { return m >= H ? 0.0f : (k < D ? W1[m * D + k] : (k == D ? b1[m] : 0.0f)); };  // [W1 | b1 | 0]
    auto w2p = [&](int m, int k) -> float { return (m < H && k < H) ? W2[m * H + k] : 0.0f; };
    auto w3p = [&](int m, int k) -> float { return (m < A && k < H) ? W3[m * H + k] : 0.0f; };
    _Float16* f1 = reinterpret_cast<_Float16*>(image.data() + kLearnedW1);
    _Float16* f2 = reinterpret_cast<_Float16*>(image.data() + kLearnedW2);
    _Float16* f3 = reinterpret_cast<_Float16*>(image.data() + kLearnedW3);
    for (int lane = 0; lane < 64; ++lane) {
      const int row = lane % 16, group = lane / 16;
      for (int mt = 0; mt < 4; ++mt)
        for (int j = 0; j < 4; ++j) f1[(mt * 64 + lane) * 4 + j] = static_cast<_Float16>(w1p(16 * mt + row, 4 * group + j));
      for (int c = 0; c < 2; ++c)      // K = 32 operands: elements 0..3 from 16-feature chunk 2c, 4..7 from chunk 2c + 1
        for (int j = 0; j < 8; ++j) {
          const int k = 16 * (2 * c + j / 4) + 4 * group + j % 4;
          f3[(c * 64 + lane) * 8 + j] = static_cast<_Float16>(w3p(row, k));
          for (int mt = 0; mt < 4; ++mt) f2[((mt * 2 + c) * 64 + lane) * 8 + j] = static_cast<_Float16>(w2p(16 * mt + row, k));
        }
    }
    float* pb2 = reinterpret_cast<float*>(image.data() + kLearnedB2);
    float* pb3 = reinterpret_cast<float*>(image.data() + kLearnedB3);
    for (int m = 0; m < H; ++m) pb2[m] = b2[m];
    for (int a = 0; a < A; ++a) pb3[a] = b3[a];
  }
  if (e->learned_dev == nullptr) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->learned_dev), kLearnedBytes));
  if (e->learned_host != image) {
    // the previous image may still be read by a rollout in flight: order the overwrite behind it
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->learned_dev, image.data(), kLearnedBytes, hipMemcpyHostToDevice));
    e->learned_host.swap(image);
  }
  LP.w.w1 = reinterpret_cast<const mbt::half4_t*>(e->learned_dev + kLearnedW1);
  LP.w.w2 = reinterpret_cast<const mbt::half8_t*>(e->learned_dev + kLearnedW2);
  LP.w.w3 = reinterpret_cast<const mbt::half8_t*>(e->learned_dev + kLearnedW3);
  LP.w.b2 = reinterpret_cast<const float*>(e->learned_dev + kLearnedB2);
  LP.w.b3 = reinterpret_cast<const float*>(e->learned_dev + kLearnedB3);
  LP.w.lin_w = reinterpret_cast<const float*>(e->learned_dev + kLearnedLinW);
  LP.w.lin_b = reinterpret_cast<const float*>(e->learned_dev + kLearnedLinB);
  return MBT_OK;
}

// Enqueue one fused rollout from the current state; trajectory pointers are device memory (or nullptr).
int launch_rollout(mbt_env* e, const mbt_policy* policy, uint32_t max_steps, float* obs_traj, float* act_traj, float* rew_traj,
                   uint32_t* steps_done, int32_t* done) {
  if (!e->was_reset) return fail(MBT_ERR_STATE, "rollout() before reset()");
  if (e->host_mask != 0) return fail(MBT_ERR_INVALID, "host-callback plugins (NumPy-only subclasses) are consulted between launches: this environment runs step by step, not as a fused rollout");
  if (e->cfg.noise_mode != MBT_NOISE_PHILOX) return fail(MBT_ERR_STATE, "rollouts draw Philox noise; this environment is in injected-noise mode");
  if (policy == nullptr) return fail(MBT_ERR_INVALID, "null policy");
  if (policy->kind == MBT_POLICY_ACTION_BUFFER) {
    const int rc_file = file_staged_action(e);
    if (rc_file != MBT_OK) return rc_file;
  }
  mbt::RolloutParams R;
  std::memset(&R, 0, sizeof R);
  const bool learned = policy->kind == MBT_POLICY_LINEAR || policy->kind == MBT_POLICY_MLP;
  mbt::LearnedPolicyParams LP;
  if (learned) {
    int rc = prepare_learned_policy(e, policy, LP);
    if (rc != MBT_OK) return rc;
  } else if (policy->kind == MBT_POLICY_FIXED) {
    R.policy = mbt::kPolicyFixed;
    for (int j = 0; j < e->act_dim; ++j) R.action[j] = static_cast<float>(policy->params[j]);
  } else if (policy->kind == MBT_POLICY_ACTION_BUFFER) {
    R.policy = mbt::kPolicyBuffer;  // the kernel reads B.action (the library's action buffer) once per lane
  } else if (policy->kind == MBT_POLICY_TIME_TABLE) {
    if (policy->table == nullptr || policy->table_rows == 0 || policy->table_cols != static_cast<uint32_t>(e->act_dim))
      return fail(MBT_ERR_INVALID, "a time-table policy holds table_rows x action_dim (%d) floats", e->act_dim);
    const size_t floats = size_t(policy->table_rows) * policy->table_cols;
    if (floats > e->policy_table_floats) {
      if (e->policy_table != nullptr) (void)hipFree(e->policy_table);
      e->policy_table = nullptr;
      e->policy_table_floats = 0;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->policy_table), floats * sizeof(float)));
      e->policy_table_floats = floats;
    }
    HIP_TRY(hipMemcpyAsync(e->policy_table, policy->table, floats * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    R.policy = mbt::kPolicyTimeTable;
    R.table = reinterpret_cast<const float2*>(e->policy_table);
    R.table_rows = policy->table_rows;
    R.table_cols = policy->table_cols;
    R.table_row0 = static_cast<uint32_t>(std::llround(e->time / e->dt));
  } else if (e->speed) {
    return fail(MBT_ERR_INVALID, "speed dynamics support the fixed and time-table policies");
  } else if (policy->kind == MBT_POLICY_AVELLANEDA_STOIKOV) {
    if (e->cfg.normalise_action) return fail(MBT_ERR_INVALID, "the Avellaneda-Stoikov policy needs normalise_action_space=False");
    if (e->cfg.dynamics_kind != MBT_DYN_LIMIT) return fail(MBT_ERR_INVALID, "the Avellaneda-Stoikov policy quotes two depths (limit-order dynamics)");
    const double g = policy->params[0], s2 = e->cfg.volatility * e->cfg.volatility, k = e->cfg.fill_exponent;
    R.policy = mbt::kPolicyAvellanedaStoikov;
    R.as_c1 = static_cast<float>(g * s2);
    R.as_c2 = static_cast<float>(g == 0.0 ? 2.0 / k : 2.0 / g * std::log(1.0 + g / k));  // BaselineAgents.py:74-79
  } else if (policy->kind == MBT_POLICY_TIME_INVENTORY_TABLE) {
    if (e->cfg.normalise_action) return fail(MBT_ERR_INVALID, "tabulated policies hold raw depths: needs normalise_action_space=False");
    if (e->cfg.dynamics_kind != MBT_DYN_LIMIT) return fail(MBT_ERR_INVALID, "tabulated policies quote two depths (limit-order dynamics)");
    if (policy->table == nullptr || policy->table_rows == 0 || policy->table_cols == 0) return fail(MBT_ERR_INVALID, "empty policy table");
    const size_t floats = size_t(policy->table_rows) * policy->table_cols * 2;
    if (floats > e->policy_table_floats) {
      if (e->policy_table != nullptr) (void)hipFree(e->policy_table);
      e->policy_table = nullptr;
      e->policy_table_floats = 0;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->policy_table), floats * sizeof(float)));
      e->policy_table_floats = floats;
    }
    HIP_TRY(hipMemcpyAsync(e->policy_table, policy->table, floats * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));  // the caller may free its table after the call
    R.policy = mbt::kPolicyTable;
    R.table = reinterpret_cast<const float2*>(e->policy_table);
    R.table_rows = policy->table_rows;
    R.table_cols = policy->table_cols;
    R.table_q_offset = policy->table_q_offset;
    R.table_row0 = static_cast<uint32_t>(std::llround(e->time / e->dt));
  } else {
    return fail(MBT_ERR_INVALID, "unknown policy kind %d", policy->kind);
  }
  // how many steps until the episode ends, on the host clock (t += dt; done = t >= T - dt/2: TE:216-220)
  double t = e->time;
  uint32_t steps = 0;
  bool terminal = false;
  while (steps < max_steps && !terminal) {
    t += e->dt;
    ++steps;
    terminal = t >= e->cfg.terminal_time - e->dt / 2;
  }
  if (steps == 0) return fail(MBT_ERR_INVALID, "max_steps must be positive");
  if (learned && !learned_rollout_is_fused(e)) {
    // policy kernel + step kernel per step, recording by device-to-device copies in the rollout's own layout
    const size_t row_obs = size_t(e->n_pad) * e->dim, row_act = size_t(e->n_pad) * e->act_dim, row_rew = e->n_pad;
    int32_t ended = 0;
    for (uint32_t k = 0; k < steps; ++k) {
      if (obs_traj != nullptr) HIP_TRY(hipMemcpyAsync(obs_traj + k * row_obs, current_obs(e), row_obs * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
      hipLaunchKernelGGL(mbt::policy_kernel, dim3(policy_blocks(e)), dim3(mbt::kBlockThreads), 0, e->stream, current_obs(e), e->action, e->dim, e->act_dim, LP,
                         e->params.pair_offset, e->philox_step, e->params.key0, e->params.key1);
      HIP_TRY(hipGetLastError());
      e->action_in_stage = false;
      if (act_traj != nullptr) HIP_TRY(hipMemcpyAsync(act_traj + k * row_act, e->action, row_act * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
      const int rc = launch_step(e, nullptr, &ended);
      if (rc != MBT_OK) return rc;
      if (rew_traj != nullptr) HIP_TRY(hipMemcpyAsync(rew_traj + k * row_rew, e->reward, row_rew * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    }
    if (obs_traj != nullptr) HIP_TRY(hipMemcpyAsync(obs_traj + size_t(steps) * row_obs, current_obs(e), row_obs * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    if (steps_done != nullptr) *steps_done = steps;
    if (done != nullptr) *done = ended;
    return MBT_OK;
  }
  R.n_steps = steps;
  R.last_is_terminal = terminal ? 1 : 0;
  R.t_start = e->time;
  R.dt_f64 = e->dt;
  R.terminal_time = e->cfg.terminal_time;
  R.obs_traj = obs_traj;
  R.act_traj = act_traj;
  R.rew_traj = rew_traj;

  mbt::StepParams& P = e->params;
  P.philox_step = e->philox_step;
  mbt::StepBuffers B;
  std::memset(&B, 0, sizeof B);
  B.state_in = e->state[e->cur];
  B.state_out = e->state[next_state(e)];
  B.action = e->action;  // read by MBT_POLICY_ACTION_BUFFER only
  B.reward = e->reward;
  B.obs = e->cfg.normalise_observation ? e->obs : nullptr;
  B.q_init = e->q_init_per_lane ? e->q_init : nullptr;
  B.resid = e->resid;
  B.events = e->record_events ? e->events : nullptr;
  B.lane_returns = e->track_returns ? e->lane_returns : nullptr;
  B.wave_sums = e->wave_sums;
  B.clip_count = e->clip_count;
  if (learned) {
    hipLaunchKernelGGL(pick_learned_rollout(e->cfg), dim3(e->n_blocks), dim3(mbt::kBlockThreads), 0, e->stream, B, P, R, LP);
    HIP_TRY(hipGetLastError());
  } else if (e->jit_rollout != nullptr) {
    void* args[] = {&B, &P, &R};
    HIP_TRY(hipModuleLaunchKernel(e->jit_rollout, e->n_blocks, 1, 1, mbt::kBlockThreads, 1, 1, 0, e->stream, args, nullptr));
  } else {
    if (e->rollout == nullptr) return fail(MBT_ERR_INVALID, "this environment has no fused rollout kernel");
    // A RECORDING rollout is a store stream of 28 B per lane and step out of one long-running kernel: capped at five workgroups
    // per CU (32 KB of dynamic LDS, like the lightest step kernel) it wrote 5.2 instead of 4.9 TB/s at 2^18 lanes and 5.65 instead
    // of 5.5 at 2^20 in every A / B of tools/microbench/mb_rollout.hip (profiles/r05_mb_rollout.txt; the store policy itself -
    // write-back, sc1, nt, system scope - made no reproducible difference: step_kernel.hpp, MBT_RECORD_STORE_POLICY).  Order-book
    // kernels only (the speed kernels stage rows through LDS of their own).  MBT_ROLLOUT_DYNAMIC_LDS overrides (measurement knob).
    uint32_t dynamic_lds = (!e->speed && (obs_traj != nullptr || act_traj != nullptr || rew_traj != nullptr)) ? 32u * 1024u : 0u;
    if (const char* v = std::getenv("MBT_ROLLOUT_DYNAMIC_LDS")) dynamic_lds = static_cast<uint32_t>(std::strtoul(v, nullptr, 10));
    hipLaunchKernelGGL(e->rollout, dim3(e->n_blocks), dim3(mbt::kBlockThreads), dynamic_lds, e->stream, B, P, R);
    HIP_TRY(hipGetLastError());
  }
  e->cur = next_state(e);
  e->time = t;
  e->philox_step += steps;
  e->episode_step += steps;
  if (steps_done != nullptr) *steps_done = steps;
  if (done != nullptr) *done = terminal ? 1 : 0;
  return MBT_OK;
}

// The row a reset writes into every lane (TE:131-140, SP:48-53): the reference's float64 values, their float32 roundings and - for
// the columns a tier holds exactly - what float32 left of them.
mbt::ResetRow make_reset_row(const mbt_env* e, double start_time, bool per_lane_q0) {
  const mbt_config& c = e->cfg;
  mbt::ResetRow row0;
  std::memset(&row0, 0, sizeof row0);
  double* x = row0.exact;  // the row as the reference's float64 values (TE:131-140, SP:48-53), then its float32 rounding
  x[0] = c.initial_cash;
  x[1] = c.initial_inventory;
  x[2] = start_time;
  x[3] = c.initial_price;
  int col = 4;  // process columns in registry order: arrival model, fill model, price impact model (TE:303-318)
  if (e->speed) {
    if (impact_has_state(c)) x[col++] = c.impact_kind == MBT_IMPACT_TEMPORARY_AND_PERMANENT ? 0.0 : c.initial_transient_impact;  // IMP:81, IMP:121
  } else {
    if (c.arrival_kind == MBT_ARR_HAWKES) {  // ARR:103
      x[col++] = c.intensity[0];
      x[col++] = c.intensity[1];
    }
    if (exogenous_fill(c)) {  // FILL:148-154
      x[col++] = c.exogenous_depth[0];
      x[col++] = c.exogenous_depth[1];
    }
    for (int j = 0; j < e->user_state_columns; ++j) x[col++] = e->user_state_initial[j];  // SP:30-31 of the user's own processes
  }
  row0.cash0 = static_cast<float>(x[0]);
  row0.q0_scalar = static_cast<float>(x[1]);
  row0.t0 = static_cast<float>(x[2]);
  row0.s0 = static_cast<float>(x[3]);
  for (int j = 4; j < 8; ++j) row0.extra[j - 4] = static_cast<float>(x[j]);
  row0.res = e->res;
  row0.precise = c.precise_state ? 1 : 0;
  if (e->res != 0) {  // precise_state: what float32 left of each value (per-lane initial inventories are float32: no remainder)
    float hi;
    for (int j = 0; j < e->res; ++j) {
      const int column = e->res_col[j];
      exact_split_host(x[column], hi, row0.lo[j]);
      if (column >= e->dim || (column == 1 && per_lane_q0)) row0.lo[j] = 0;
    }
  }
  return row0;
}

// reuse_q0: an automatic reset (mbt_env_step_many_device) restarts from the initial inventories of the last explicit one
int do_reset(mbt_env* e, double start_time, const float* q0_host, bool reuse_q0 = false) {
  e->stage_outputs_valid = false;
  const mbt_config& c = e->cfg;
  if (!(start_time >= 0.0 && start_time < c.terminal_time))
    return fail(MBT_ERR_INVALID, "start time %g is not within [0, terminal_time)", start_time);  // TE:267
  if (!reuse_q0) {
    e->q_init_per_lane = false;
    e->q0_per_lane_reset = q0_host != nullptr;
    if (q0_host != nullptr) {
      HIP_TRY(hipMemcpyAsync(e->q_init, q0_host, e->n * sizeof(float), hipMemcpyHostToDevice, e->stream));
      e->q_init_per_lane = c.reward_kind == MBT_REW_CJ_MM || c.reward_kind == MBT_REW_CJ_OE;
    }
  }
  const bool per_lane_q0 = reuse_q0 ? e->q0_per_lane_reset : q0_host != nullptr;
  e->time = e->start_time = start_time;
  e->episode_step = 0;
  e->cur = 0;
  mbt::StepParams& P = e->params;
  fill_episode_params(e);
  const uint32_t threads = 256, blocks = (e->n_pad + threads - 1) / threads;
  const mbt::ResetRow row0 = make_reset_row(e, start_time, per_lane_q0);
  hipLaunchKernelGGL(mbt::reset_kernel, dim3(blocks > 0 ? blocks : 1), dim3(threads), 0, e->stream, e->state[0],
                     c.normalise_observation ? e->obs : nullptr, e->lane_returns, e->wave_sums,
                     per_lane_q0 ? e->q_init : nullptr, row0, e->n_pad, e->n_waves, e->dim, P, e->resid);
  HIP_TRY(hipGetLastError());
  if (q0_host != nullptr) HIP_TRY(hipStreamSynchronize(e->stream));  // q0_host may be freed by the caller
  e->was_reset = true;
  return MBT_OK;
}

float* current_obs(mbt_env* e) { return e->cfg.normalise_observation ? e->obs : e->state[e->cur]; }

// ---- run-time compilation of user plugins (mbt_env_create_jit) ----------------------------------------------------------
// The kernel source IS this library's own: build.py embeds csrc/philox.hpp and csrc/step_kernel.hpp as string literals
// (embedded_sources.inc), hiprtc compiles them with the user's two device functions in front of one instantiation of
// step_body / rollout_body.  hiprtc is bound with dlopen like RCCL (PyTorch ships its own copy).
extern const char* const kEmbeddedPhilox;
extern const char* const kEmbeddedStepKernel;
thread_local std::string g_jit_log;

typedef struct _hiprtcProgram* rtc_program;
struct HiprtcApi {
  int (*create)(rtc_program*, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*compile)(rtc_program, int, const char**) = nullptr;
  int (*log_size)(rtc_program, size_t*) = nullptr;
  int (*log)(rtc_program, char*) = nullptr;
  int (*code_size)(rtc_program, size_t*) = nullptr;
  int (*code)(rtc_program, char*) = nullptr;
  int (*destroy)(rtc_program*) = nullptr;
  bool ok = false;
  std::string why;
};

const HiprtcApi& hiprtc() {
  static const HiprtcApi api = [] {
    HiprtcApi a;
    void* h = nullptr;
    const char* override_path = std::getenv("MBT_HIPRTC_LIBRARY");
    if (override_path != nullptr && override_path[0] != 0) h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
    const char* names[] = {"libhiprtc.so.7", "libhiprtc.so"};
    for (const char* name : names)
      if (h == nullptr) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : names)
      if (h == nullptr) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (h == nullptr) h = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
    if (h == nullptr) {
      const char* err = dlerror();
      a.why = std::string("libhiprtc.so could not be loaded: ") + (err != nullptr ? err : "unknown error");
      return a;
    }
    a.create = reinterpret_cast<decltype(a.create)>(dlsym(h, "hiprtcCreateProgram"));
    a.compile = reinterpret_cast<decltype(a.compile)>(dlsym(h, "hiprtcCompileProgram"));
    a.log_size = reinterpret_cast<decltype(a.log_size)>(dlsym(h, "hiprtcGetProgramLogSize"));
    a.log = reinterpret_cast<decltype(a.log)>(dlsym(h, "hiprtcGetProgramLog"));
    a.code_size = reinterpret_cast<decltype(a.code_size)>(dlsym(h, "hiprtcGetCodeSize"));
    a.code = reinterpret_cast<decltype(a.code)>(dlsym(h, "hiprtcGetCode"));
    a.destroy = reinterpret_cast<decltype(a.destroy)>(dlsym(h, "hiprtcDestroyProgram"));
    a.ok = a.create && a.compile && a.log_size && a.log && a.code_size && a.code && a.destroy;
    if (!a.ok) a.why = "libhiprtc is loaded but lacks part of the hiprtc API";
    return a;
  }();
  return api;
}

bool valid_identifier(const std::string& name) {
  if (name.empty() || !(std::isalpha(static_cast<unsigned char>(name[0])) || name[0] == '_')) return false;
  for (char ch : name)
    if (!(std::isalnum(static_cast<unsigned char>(ch)) || ch == '_')) return false;
  return true;
}

// "k, alpha" -> "const double k = p[0]; const double alpha = p[1];"
int param_declarations(const char* names, std::string& out) {
  out.clear();
  if (names == nullptr) return MBT_OK;
  std::string token;
  int index = 0;
  const std::string all = std::string(names) + ",";
  for (char ch : all) {
    if (ch == ',') {
      if (!token.empty()) {
        if (!valid_identifier(token)) return fail(MBT_ERR_INVALID, "'%s' is not a C identifier (parameter names of a user expression)", token.c_str());
        if (index >= 8) return fail(MBT_ERR_INVALID, "a user expression takes at most 8 parameters");
        out += "  const double " + token + " = p[" + std::to_string(index++) + "];\n";
        token.clear();
      }
    } else if (!std::isspace(static_cast<unsigned char>(ch))) {
      token += ch;
    }
  }
  return MBT_OK;
}

// source -> gfx950 code object (no device needed: hiprtc cross-compiles like hipcc does)
int jit_compile(const std::string& source, std::vector<char>& code) {
  g_jit_log.clear();
  const HiprtcApi& api = hiprtc();
  if (!api.ok) return fail(MBT_ERR_HIP, "%s", api.why.c_str());
  rtc_program prog = nullptr;
  const char* headers[] = {kEmbeddedPhilox, kEmbeddedStepKernel};
  const char* header_names[] = {"philox.hpp", "step_kernel.hpp"};
  if (api.create(&prog, source.c_str(), "mbt_user_plugins.hip", 2, headers, header_names) != 0) return fail(MBT_ERR_HIP, "hiprtcCreateProgram failed");
  // the flags of the ahead-of-time build (mbt_gym_amd/build.py): in particular -ffp-contract=off, which keeps the step and
  // the rollout instantiation bit-identical
  const char* options[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function"};
  const int rc = api.compile(prog, 5, options);
  size_t log_bytes = 0;
  if (api.log_size(prog, &log_bytes) == 0 && log_bytes > 1) {
    g_jit_log.resize(log_bytes);
    (void)api.log(prog, &g_jit_log[0]);
    while (!g_jit_log.empty() && g_jit_log.back() == 0) g_jit_log.pop_back();
  }
  if (rc != 0) {
    (void)api.destroy(&prog);
    return fail(MBT_ERR_INVALID, "the user expression does not compile (hiprtc status %d): see mbt_jit_log()", rc);
  }
  size_t code_bytes = 0;
  if (api.code_size(prog, &code_bytes) != 0 || code_bytes == 0) {
    (void)api.destroy(&prog);
    return fail(MBT_ERR_HIP, "hiprtcGetCodeSize failed");
  }
  code.resize(code_bytes);
  const int got = api.code(prog, code.data());
  (void)api.destroy(&prog);
  if (got != 0) return fail(MBT_ERR_HIP, "hiprtcGetCode failed");
  return MBT_OK;
}

struct JitKernels {
  hipModule_t module = nullptr;
  hipFunction_t step = nullptr, rollout = nullptr, step_mirror = nullptr, step_captured = nullptr;
};

// One module per distinct (device, generated source): environments that share plugins share the compiled code.  Modules
// live until the process exits (they are small and a handle may outlive the environment that first asked for them).
int jit_build(int device, const std::string& source, bool with_rollout, JitKernels& out) {
  static std::mutex mutex;
  static std::map<std::pair<int, std::string>, JitKernels> cache;
  std::lock_guard<std::mutex> guard(mutex);
  g_jit_log.clear();  // a cache hit has nothing to say
  const auto key = std::make_pair(device, source);
  const auto hit = cache.find(key);
  if (hit != cache.end()) {
    out = hit->second;
    return MBT_OK;
  }
  std::vector<char> code;
  const int rc_compile = jit_compile(source, code);
  if (rc_compile != MBT_OK) return rc_compile;
  JitKernels k;
  HIP_TRY(hipModuleLoadData(&k.module, code.data()));
  HIP_TRY(hipModuleGetFunction(&k.step, k.module, "mbt_user_step"));
  if (with_rollout) HIP_TRY(hipModuleGetFunction(&k.rollout, k.module, "mbt_user_rollout"));
  if (hipModuleGetFunction(&k.step_mirror, k.module, "mbt_user_step_mirror") != hipSuccess) {  // (injected-noise units have none)
    (void)hipGetLastError();
    k.step_mirror = nullptr;
  }
  if (hipModuleGetFunction(&k.step_captured, k.module, "mbt_user_step_captured") != hipSuccess) {  // (nor do units with host callbacks)
    (void)hipGetLastError();
    k.step_captured = nullptr;
  }
  cache.emplace(key, k);
  out = k;
  return MBT_OK;
}

// The translation unit: the library's own kernel source with the user's functions and ONE instantiation.
int jit_source(const mbt_config& c, const mbt_user_code& u, std::string& src) {
  const bool user_fill = c.fill_kind == MBT_FILL_USER, user_reward = c.reward_kind == MBT_REW_USER, user_arrival = c.arrival_kind == MBT_ARR_USER;
  const bool user_mid = c.midprice_kind == MBT_MID_USER;
  const int host_mask = (c.fill_kind == MBT_FILL_HOST ? mbt::kHostFill : 0) | (c.arrival_kind == MBT_ARR_HOST ? mbt::kHostArrival : 0) |
                        (c.reward_kind == MBT_REW_HOST ? mbt::kHostReward : 0);
  std::string fill_decl, reward_decl, arrival_decl, mid_decl, state_decl;
  if (int rc_state = param_declarations(u.state_param_names, state_decl); rc_state != MBT_OK) return rc_state;
  const int user_state = u.state_columns;
  // the symbols every process expression may read beyond its own arguments
  const std::string process_symbols = "  const double x0 = u_.x0, x1 = u_.x1, z1 = u_.z1, z2 = u_.z2;\n  (void)x0; (void)x1; (void)z1; (void)z2;\n";
  if (int rc_mid = param_declarations(u.midprice_param_names, mid_decl); rc_mid != MBT_OK) return rc_mid;
  int rc = param_declarations(u.fill_param_names, fill_decl);
  if (rc != MBT_OK) return rc;
  rc = param_declarations(u.reward_param_names, reward_decl);
  if (rc != MBT_OK) return rc;
  rc = param_declarations(u.arrival_param_names, arrival_decl);
  if (rc != MBT_OK) return rc;
  const int arr = c.arrival_kind == MBT_ARR_HAWKES ? mbt::kArrHawkes : mbt::kArrPoisson;
  const int dyn = c.dynamics_kind == MBT_DYN_LIMIT_AND_MARKET ? mbt::kDynLimitAndMarket : c.dynamics_kind == MBT_DYN_AT_THE_TOUCH ? mbt::kDynTouch : mbt::kDynLimit;
  const bool inject = c.noise_mode == MBT_NOISE_INJECTED;
  src = "#define MBT_JIT_USER_CODE 1\n#include \"step_kernel.hpp\"\nnamespace mbt {\n";
  src += "__device__ double mbt_user_fill_probability(double depth, int side, const double* p) {\n" + fill_decl + "  return static_cast<double>(" +
         std::string(user_fill ? u.fill_probability : "0.0") + ");\n}\n";
  src += "__device__ double mbt_user_reward(const UserRewardArgs& s_, const double* p) {\n"
         "  const double cash = s_.cash, q = s_.q, t = s_.t, mid = s_.mid, cash_next = s_.cash_next, q_next = s_.q_next, t_next = s_.t_next,\n"
         "               mid_next = s_.mid_next, a0 = s_.a0, a1 = s_.a1, a2 = s_.a2, a3 = s_.a3, pnl = s_.pnl, dt = s_.dt,\n"
         "               is_terminal = s_.is_terminal, q0 = s_.q0, episode_length = s_.episode_length;\n"
         "  (void)cash; (void)q; (void)t; (void)mid; (void)cash_next; (void)q_next; (void)t_next; (void)mid_next; (void)a0; (void)a1; (void)a2;\n"
         "  (void)a3; (void)pnl; (void)dt; (void)is_terminal; (void)q0; (void)episode_length; (void)p;\n" +
         reward_decl + "  return static_cast<double>(" + std::string(user_reward ? u.reward : "0.0") + ");\n}\n";
  src += "__device__ double mbt_user_arrival_probability(double t, int side, double dt, const UserProcessState& u_, const double* p) {\n  (void)t; (void)side; (void)dt; (void)p;\n" +
         process_symbols + arrival_decl + "  return static_cast<double>(" + std::string(user_arrival ? u.arrival_probability : "0.0") + ");\n}\n";
  src += "__device__ double mbt_user_midprice_increment(double S, double t, double z, double dt, double fills_bid, double fills_ask, const UserProcessState& u_, const double* p) {\n"
         "  (void)S; (void)t; (void)z; (void)dt; (void)fills_bid; (void)fills_ask; (void)p;\n" + process_symbols + mid_decl +
         "  return static_cast<double>(" + std::string(user_mid ? u.midprice_increment : "0.0") + ");\n}\n";
  const auto owner_dt = [&](int j) { return std::string(u.state_owner[j] == 1 ? "dt_arr" : "dt_mid"); };  // each column advances with its owner's step size (SP:21)
  src += "__device__ double mbt_user_state_next(int which, double S, double t, double dt_mid, double dt_arr, double z, double arr_bid, double arr_ask, double fills_bid, "
         "double fills_ask, const UserProcessState& u_, const double* p) {\n  (void)which; (void)S; (void)t; (void)dt_mid; (void)dt_arr; (void)z; (void)arr_bid; (void)arr_ask; "
         "(void)fills_bid; (void)fills_ask; (void)p;\n" + process_symbols +
         "  const double S_next = u_.S_next, t_next = u_.t_next, q_next = u_.q_next, cash_next = u_.cash_next;\n  (void)S_next; (void)t_next; (void)q_next; (void)cash_next;\n" + state_decl +
         // (a column the HOST advances - a host-callback arrival model's own state - passes through the kernel unchanged: "x0" / "x1")
         "  if (which == 0) { const double dt = " + owner_dt(0) + "; (void)dt; return static_cast<double>(" + std::string(user_state > 0 ? (u.state_update[0] != nullptr && u.state_update[0][0] != 0 ? u.state_update[0] : "x0") : "0.0") + "); }\n"
         "  { const double dt = " + owner_dt(1) + "; (void)dt; return static_cast<double>(" + std::string(user_state > 1 ? (u.state_update[1] != nullptr && u.state_update[1][0] != 0 ? u.state_update[1] : "x1") : "0.0") + "); }\n}\n}  // namespace mbt\n";
  {  // the kernel's shape as its list of named tags (step_kernel.hpp: Variant) - the general tier: any midprice, every reward, normalisation flags at run time
    std::string tags;
    const auto tag = [&tags](bool on, const std::string& name) {
      if (on) tags += (tags.empty() ? "" : ", ") + ("mbt::shape::" + name);
    };
    tag(arr == mbt::kArrHawkes && !exact_intensities(c), "hawkes");
    tag(arr == mbt::kArrHawkes && exact_intensities(c), "hawkes_exact");
    tag(dyn == mbt::kDynLimitAndMarket, "limit_and_market");
    tag(dyn == mbt::kDynTouch, "touch");
    tag(true, "normalised");
    tag(inject, "injected");
    tag(exogenous_fill(c), "exogenous");
    tag(c.precise_state != 0, "precise");
    tag(user_fill, "user_fill");
    tag(user_reward, "user_reward");
    tag(user_arrival, "user_arrival");
    tag(user_mid, "user_mid");
    tag(user_state != 0, "user_state<" + std::to_string(user_state) + ">");
    tag(u.extra_normals != 0, "user_draws");
    tag(host_mask != 0, "host<" + std::to_string(host_mask) + ">");
    src += "using V = mbt::Variant<" + tags + ">;\n";
  }
  src += "extern \"C\" __global__ __launch_bounds__(256) void mbt_user_step(const mbt::StepBuffers B, const mbt::StepParams P) { mbt::step_body<V, false>(B, P); }\n";
  if (!inject)  // the small-batch host-API instantiation (step_kernel.hpp: signal_host)
    src += "extern \"C\" __global__ __launch_bounds__(256) void mbt_user_step_mirror(const mbt::StepBuffers B, const mbt::StepParams P) { mbt::step_body<V, false, true>(B, P); }\n";
  if (!inject && host_mask == 0)  // the graph-capturable instantiation (step_kernel.hpp: captured_step_kernel)
    src += "extern \"C\" __global__ __launch_bounds__(256) void mbt_user_step_captured(const mbt::StepBuffers B, const mbt::StepParams P, const mbt::CapturedParams C) { "
           "mbt::captured_step_body<V, false>(B, P, C); }\n";
  if (!inject && host_mask == 0)
    src += "extern \"C\" __global__ __launch_bounds__(256) void mbt_user_rollout(const mbt::StepBuffers B, const mbt::StepParams P, const mbt::RolloutParams R) { "
           "mbt::rollout_body<V>(B, P, R); }\n";
  return MBT_OK;
}

// ---- RCCL, bound at run time ------------------------------------------------------------------------------------------
// The trajectory axis shards with no data-path collective; the one exchange is 24 bytes per episode.  libmbtenv therefore
// does not link librccl: the symbols are looked up in the copy the process already has (PyTorch bundles its own; two
// copies of RCCL - like two copies of the HIP runtime - must not meet in one process) or, failing that, in the system's.
struct RcclApi {
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclCommCount) comm_count = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  bool ok = false;
  std::string why;
};

// RTLD_LOCAL on purpose: librccl brings librocm_smi64 along, and /opt/rocm/lib/libamd_smi.so (which RCCL opens at
// initialisation) carries its own copy of the same amd::smi globals - with the first copy in the global scope the second
// binds to it and both destructors free it at exit ("double free or corruption", measured: tools/dbg/exit_abort.sh).
const RcclApi& rccl() {
  static const RcclApi api = [] {
    RcclApi a;
    void* h = nullptr;
    const char* override_path = std::getenv("MBT_RCCL_LIBRARY");
    if (override_path != nullptr && override_path[0] != 0) h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* name : names)  // already in the process?
      if (h == nullptr) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : names)
      if (h == nullptr) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (h == nullptr) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (h == nullptr) {
      const char* err = dlerror();
      a.why = std::string("librccl.so.1 could not be loaded: ") + (err != nullptr ? err : "unknown error");
      return a;
    }
    a.all_reduce = reinterpret_cast<decltype(a.all_reduce)>(dlsym(h, "ncclAllReduce"));
    a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    a.comm_count = reinterpret_cast<decltype(a.comm_count)>(dlsym(h, "ncclCommCount"));
    a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(h, "ncclGetErrorString"));
    a.ok = a.all_reduce != nullptr && a.get_unique_id != nullptr && a.comm_init_rank != nullptr && a.comm_destroy != nullptr;
    if (!a.ok) a.why = "librccl is loaded but lacks ncclAllReduce / ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy";
    return a;
  }();
  return api;
}

int rccl_fail(ncclResult_t r, const char* what) {
  const RcclApi& api = rccl();
  return fail(MBT_ERR_HIP, "%s failed: %s", what, api.error_string != nullptr ? api.error_string(r) : "RCCL error");
}
static_assert(sizeof(ncclUniqueId) == MBT_COMM_ID_BYTES, "ncclUniqueId size");

// Enqueue "this episode's [sum R, sum R^2, lanes], over all ranks, into the next slot of the episode log": one reduction
// launch, one in-place 24-byte all-reduce when a communicator is set, one copy to pinned memory, one event - all on the
// environment's stream, nothing waits.
int log_wait_oldest(mbt_env* e, double sums[3]);
int log_push(mbt_env* e) {
  if (e->log_count == mbt_env::kLogSlots) {
    double dropped[3];
    int rc = log_wait_oldest(e, dropped);  // the ring is full: the oldest entry is lost to the log (documented in the header)
    if (rc < 0) return rc;
  }
  const uint32_t slot = (e->log_head + e->log_count) % mbt_env::kLogSlots;
  double* dev = e->log_dev + 3 * slot;
  hipLaunchKernelGGL(mbt::reduce_returns_kernel, dim3(1), dim3(256), 0, e->stream, e->wave_sums, e->n_waves,
                     e->track_returns ? e->lane_returns : nullptr, e->n, dev);
  HIP_TRY(hipGetLastError());
  if (e->comm != nullptr) {
    const RcclApi& api = rccl();
    if (!api.ok) return fail(MBT_ERR_HIP, "%s", api.why.c_str());
    ncclResult_t r = api.all_reduce(dev, dev, 3, ncclDouble, ncclSum, static_cast<ncclComm_t>(e->comm), e->stream);
    if (r != ncclSuccess) return rccl_fail(r, "ncclAllReduce");
  }
  HIP_TRY(hipMemcpyAsync(e->log_host + 3 * slot, dev, 3 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipEventRecord(e->log_event[slot], e->stream));
  e->log_count += 1;
  return MBT_OK;
}

int log_wait_oldest(mbt_env* e, double sums[3]) {
  const uint32_t slot = e->log_head;
  gate_open(e);  // (never block on work that sits behind a closed launch gate)
  HIP_TRY(hipEventSynchronize(e->log_event[slot]));
  for (int j = 0; j < 3; ++j) sums[j] = e->log_host[3 * slot + j];
  e->log_head = (e->log_head + 1) % mbt_env::kLogSlots;
  e->log_count -= 1;
  return 1;
}

}  // namespace

extern "C" {

#ifndef MBT_SOURCE_HASH
#define MBT_SOURCE_HASH "unknown"
#endif
// the marker lets build.py read the hash out of the file without loading the library
static const char kSourceHash[] = "mbt-source-hash:" MBT_SOURCE_HASH;
const char* mbt_source_hash(void) { return kSourceHash + 16; }
uint32_t mbt_abi_version(void) { return MBT_ABI_VERSION; }
size_t mbt_config_sizeof(void) { return sizeof(mbt_config); }
const char* mbt_last_error(void) { return g_error.c_str(); }

int mbt_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

int mbt_device_name(int device, char* buf, size_t buf_len) {
  if (buf == nullptr || buf_len == 0) return fail(MBT_ERR_INVALID, "null buffer");
  int count = mbt_device_count();
  if (device < 0 || device >= count) return fail(MBT_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, count);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  std::snprintf(buf, buf_len, "%s", prop.gcnArchName);
  return MBT_OK;
}

// The host-callback kinds the kernels never see, lowered to what the kernels run:
//   MBT_MID_HOST -> a midprice that stands still during the launch (the caller's update() moves it between launches,
//                   mbt_env_set_host_state_columns);
//   MBT_REW_HOST with SPEED dynamics -> the kernel's PnL (the speed kernels are built ahead of time, without a host variant):
//                   mbt_env_set_host_rewards then REPLACES what the kernel filed - reward buffer and return sums.
// `reason` is set when the configuration cannot run.
struct HostLowering {
  bool midprice = false, speed_reward = false;
};
static HostLowering lower_host_kinds(const mbt_config& in, mbt_config& out, const char** reason) {
  HostLowering h;
  out = in;
  *reason = nullptr;
  if (in.midprice_kind == MBT_MID_HOST) {
    if (in.reward_kind != MBT_REW_HOST)
      *reason = "MBT_MID_HOST: the midprice moves after the launch, so the step's reward is formed by the caller - reward_kind must be MBT_REW_HOST";
    out.midprice_kind = MBT_MID_CONSTANT;
    h.midprice = true;
  }
  if (in.dynamics_kind == MBT_DYN_SPEED && in.reward_kind == MBT_REW_HOST) {
    out.reward_kind = MBT_REW_PNL;
    h.speed_reward = true;
  }
  return h;
}

static int create_env(const mbt_config* cfg, const mbt_user_code* code, mbt_env** out) {
  if (cfg == nullptr || out == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  *out = nullptr;
  // first of all: a caller built against another ABI passes structs of another SIZE - nothing of *cfg beyond its first word,
  // and nothing of *code, may be read before the versions are known to agree
  if (cfg->abi_version != MBT_ABI_VERSION)
    return fail(MBT_ERR_ABI, "mbt_config.abi_version %u != library %u", cfg->abi_version, MBT_ABI_VERSION);
  mbt_config lowered;
  const char* refused = nullptr;
  const HostLowering host_lowered = lower_host_kinds(*cfg, lowered, &refused);
  if (refused != nullptr) return fail(MBT_ERR_INVALID, "%s", refused);
  cfg = &lowered;
  const bool host_mid = host_lowered.midprice;
  const bool user_arrival = cfg->arrival_kind == MBT_ARR_USER;
  const bool user_fill = cfg->fill_kind == MBT_FILL_USER, user_reward = cfg->reward_kind == MBT_REW_USER;
  const bool user_mid = cfg->midprice_kind == MBT_MID_USER;
  const bool host_fill = cfg->fill_kind == MBT_FILL_HOST, host_arrival = cfg->arrival_kind == MBT_ARR_HOST, host_reward = cfg->reward_kind == MBT_REW_HOST;
  const bool any_host = host_fill || host_arrival || host_reward;
  const bool any_user = user_fill || user_reward || user_arrival || user_mid;
  const bool needs_jit = any_user || any_host;
  static const mbt_user_code kNoUserCode = {};  // host-callback kinds alone need no expressions: mbt_env_create serves them
  if (code == nullptr && any_host && !any_user) code = &kNoUserCode;
  if (user_mid && (code == nullptr || code->midprice_increment == nullptr || code->midprice_increment[0] == 0))
    return fail(MBT_ERR_INVALID, "MBT_MID_USER without a midprice_increment expression (mbt_env_create_jit)");
  if (needs_jit && code == nullptr)
    return fail(MBT_ERR_INVALID, "user-defined plugin kinds (MBT_ARR_USER / MBT_FILL_USER / MBT_REW_USER) need their device expressions: use mbt_env_create_jit");
  if (user_arrival && (code->arrival_probability == nullptr || code->arrival_probability[0] == 0)) return fail(MBT_ERR_INVALID, "MBT_ARR_USER without an arrival_probability expression");
  if (user_fill && (code->fill_probability == nullptr || code->fill_probability[0] == 0)) return fail(MBT_ERR_INVALID, "MBT_FILL_USER without a fill_probability expression");
  if (user_reward && (code->reward == nullptr || code->reward[0] == 0)) return fail(MBT_ERR_INVALID, "MBT_REW_USER without a reward expression");
  if (code != nullptr && !needs_jit) return fail(MBT_ERR_INVALID, "mbt_env_create_jit: no plugin kind of the configuration names a user-defined plugin");
  if (needs_jit) {
    if (cfg->dynamics_kind != MBT_DYN_LIMIT && cfg->dynamics_kind != MBT_DYN_LIMIT_AND_MARKET && !(cfg->dynamics_kind == MBT_DYN_AT_THE_TOUCH && !user_fill && !host_fill))
      return fail(MBT_ERR_INVALID, "user-defined plugins run on the order-book kernels (a fill model needs limit or limit + market dynamics)");
    if (any_host && cfg->fill_kind == MBT_FILL_EXOGENOUS_MM && host_fill) return fail(MBT_ERR_INVALID, "a fill model is either built in or a host callback");
    if (code->state_columns < 0 || code->state_columns > 2) return fail(MBT_ERR_INVALID, "user processes own at most two state columns (got %d)", code->state_columns);
    if (code->state_columns > 0) {
      if (!(user_mid || user_arrival || host_arrival || host_mid || host_fill)) return fail(MBT_ERR_INVALID, "user state columns belong to a user-defined midprice, arrival or host-callback fill model (MBT_MID_USER / MBT_MID_HOST / MBT_ARR_USER / MBT_ARR_HOST / MBT_FILL_HOST)");
      if (cfg->arrival_kind == MBT_ARR_HAWKES || exogenous_fill(*cfg))
        return fail(MBT_ERR_INVALID, "user state columns take the place of the Hawkes intensities / exogenous depths: Poisson-type or user arrivals, exponential or user fills");
      for (int j = 0; j < code->state_columns; ++j) {
        const bool host_owned = (host_arrival && code->state_owner[j] == 1) || (host_mid && code->state_owner[j] == 0) || (host_fill && code->state_owner[j] == 2);  // advanced by the caller's update() on the host (mbt_env_set_host_state_columns)
        if (!host_owned && (code->state_update[j] == nullptr || code->state_update[j][0] == 0)) return fail(MBT_ERR_INVALID, "user state column %d has no update expression", j);
      }
    }
    if (code->extra_normals && !(user_mid || user_arrival)) return fail(MBT_ERR_INVALID, "extra normals are drawn for user-defined midprice / arrival models");
  }
  if (cfg->num_trajectories == 0 || cfg->num_trajectories > 0x7FFFF000ull)
    return fail(MBT_ERR_INVALID, "num_trajectories %llu out of range", (unsigned long long)cfg->num_trajectories);
  if (cfg->n_steps == 0 || !(cfg->terminal_time > 0.0)) return fail(MBT_ERR_INVALID, "n_steps and terminal_time must be positive");
  const bool speed = cfg->dynamics_kind == MBT_DYN_SPEED;
  if (cfg->midprice_kind < MBT_MID_BROWNIAN || cfg->midprice_kind > MBT_MID_USER)
    return fail(MBT_ERR_INVALID, "midprice kind %d has no device implementation", cfg->midprice_kind);
  if (cfg->reward_terminal_time != 0.0 && !(cfg->reward_terminal_time > 0.0)) return fail(MBT_ERR_INVALID, "reward_terminal_time must be positive (or 0 = terminal_time)");
  if (cfg->dynamics_kind < MBT_DYN_LIMIT || cfg->dynamics_kind > MBT_DYN_SPEED)
    return fail(MBT_ERR_INVALID, "dynamics kind %d has no device implementation", cfg->dynamics_kind);
  if (cfg->reward_kind < MBT_REW_PNL || cfg->reward_kind > MBT_REW_HOST)
    return fail(MBT_ERR_INVALID, "reward kind %d has no device implementation", cfg->reward_kind);
  if (speed) {
    if (cfg->arrival_kind != MBT_ARR_NONE || cfg->fill_kind != MBT_FILL_NONE)
      return fail(MBT_ERR_INVALID, "speed dynamics take no arrival or fill model (MD:273-275)");
    if (cfg->impact_kind < MBT_IMPACT_TEMPORARY_POWER || cfg->impact_kind > MBT_IMPACT_HOST_STATE)
      return fail(MBT_ERR_INVALID, "speed dynamics need a price impact model (impact kind %d)", cfg->impact_kind);
    if (host_impact(*cfg) && !cfg->precise_state)
      return fail(MBT_ERR_INVALID, "MBT_IMPACT_HOST: the caller's get_impact() returns float64 values - set precise_state");
    if (cfg->midprice_kind == MBT_MID_BROWNIAN_JUMP || cfg->midprice_kind == MBT_MID_OU_JUMP ||
        (cfg->midprice_kind == MBT_MID_LINEAR_SDE && cfg->jump_size != 0.0))
      return fail(MBT_ERR_INVALID, "jump midprice models move on the agent's fills; speed dynamics have none");
    if (cfg->midprice_kind == MBT_MID_USER) return fail(MBT_ERR_INVALID, "user-defined midprice expressions run on the order-book kernels");
    if (cfg->trajectory_offset % mbt::kSpeedTileLanes != 0)
      return fail(MBT_ERR_INVALID, "speed dynamics draw noise per 1024-lane tile: trajectory_offset must be a multiple of 1024");
  } else {
    if (cfg->arrival_kind != MBT_ARR_POISSON && cfg->arrival_kind != MBT_ARR_HAWKES && cfg->arrival_kind != MBT_ARR_POISSON_NONLINEAR && !user_arrival && !host_arrival)
      return fail(MBT_ERR_INVALID, "arrival kind %d has no device implementation for order-book dynamics", cfg->arrival_kind);
    if (cfg->dynamics_kind == MBT_DYN_AT_THE_TOUCH) {
      if (cfg->normalise_action) return fail(MBT_ERR_INVALID, "at-the-touch actions are binary: normalise_action_space must be False");
    } else if (user_fill || host_fill) {
      // the expression (or the caller's own code) is the model: no built-in parameter to validate
    } else if (!(cfg->fill_exponent > 0.0)) {
      return fail(MBT_ERR_INVALID, "fill_exponent must be positive (got %g)", cfg->fill_exponent);
    } else if (cfg->fill_kind == MBT_FILL_EXOGENOUS_MM) {
      if (!(cfg->base_fill_probability >= 0.0 && cfg->base_fill_probability <= 1.0))
        return fail(MBT_ERR_INVALID, "base_fill_probability %g is not a probability", cfg->base_fill_probability);
    } else if (cfg->fill_kind != MBT_FILL_EXPONENTIAL) {
      return fail(MBT_ERR_INVALID, "fill kind %d has no device implementation", cfg->fill_kind);
    }
    if (cfg->arrival_kind == MBT_ARR_HAWKES && !cfg->allow_stiff_hawkes) {
      const double dt_arr = cfg->arrival_step_size > 0.0 ? cfg->arrival_step_size : cfg->terminal_time / cfg->n_steps;
      if (!(cfg->hawkes_speed * dt_arr < 1.0))
        return fail(MBT_ERR_INVALID, "Hawkes mean_reversion_speed * step_size = %g >= 1: the intensity recursion (ARR:110-119) oscillates (>= 2: diverges) "
                    "and float32 state stops tracking the float64 reference; set allow_stiff_hawkes to run it anyway", cfg->hawkes_speed * dt_arr);
    }
    if (cfg->reward_kind == MBT_REW_CJ_OE) return fail(MBT_ERR_INVALID, "CjOeCriterion needs the one-dimensional action of speed dynamics (RW:65)");
    if (cfg->impact_kind != MBT_IMPACT_NONE) return fail(MBT_ERR_INVALID, "price impact models belong to speed dynamics");
    if (cfg->trajectory_offset % mbt::kTileLanes != 0)
      return fail(MBT_ERR_INVALID, "order-book dynamics draw noise per 512-lane tile: trajectory_offset must be a multiple of 512");
  }
  if (cfg->noise_mode != MBT_NOISE_PHILOX && cfg->noise_mode != MBT_NOISE_INJECTED)
    return fail(MBT_ERR_INVALID, "unknown noise mode %d", cfg->noise_mode);
  int rc = check_device(cfg->device);
  if (rc != MBT_OK) return rc;
  HIP_TRY(hipSetDevice(cfg->device));

  mbt_env* e = new (std::nothrow) mbt_env();
  if (e == nullptr) return fail(MBT_ERR_INVALID, "out of host memory");
  e->cfg = *cfg;
  e->speed = speed;
  e->user_state_columns = needs_jit ? code->state_columns : 0;
  e->user_draws = needs_jit && code->extra_normals != 0;
  // the columns host-callback processes own: one contiguous block in registry order (midprice column 3, the midprice model's
  // further columns, then a host arrival model's)
  if (host_mid) e->host_state_first = 3, e->host_state_count = 1;
  if (needs_jit)
    for (int j = 0; j < code->state_columns; ++j)
      if ((host_arrival && code->state_owner[j] == 1) || (host_mid && code->state_owner[j] == 0) || (host_fill && code->state_owner[j] == 2)) {
        if (e->host_state_count == 0) e->host_state_first = 4 + j;
        e->host_state_count += 1;
      }
  if (cfg->impact_kind == MBT_IMPACT_HOST_STATE) {  // the impact-state column, behind the midprice column of speed rows
    if (e->host_state_count == 0) e->host_state_first = 4;
    e->host_state_count += 1;
  }
  e->host_mid = host_mid;
  for (int j = 0; j < 2; ++j) e->user_state_initial[j] = needs_jit ? code->state_initial[j] : 0.0;
  e->dim = speed ? (impact_has_state(*cfg) ? 5 : 4) : 4 + (cfg->arrival_kind == MBT_ARR_HAWKES ? 2 : 0) + (exogenous_fill(*cfg) ? 2 : 0) + e->user_state_columns;
  e->act_dim = speed ? 1 : (cfg->dynamics_kind == MBT_DYN_LIMIT_AND_MARKET ? 4 : 2);
  e->n = static_cast<uint32_t>(cfg->num_trajectories);
  // a thread owns four lanes of a 1024-lane tile (speed) or two lanes of a 512-lane tile (order book): pad to whole tiles
  const uint32_t tile = speed ? mbt::kSpeedTileLanes : mbt::kTileLanes;
  e->n_pad = ((e->n + tile - 1u) / tile) * tile;
  e->n_pairs = e->n_pad / 2;
  const uint32_t n_threads = speed ? e->n_pad / 4 : e->n_pairs;
  e->n_blocks = (n_threads + mbt::kBlockThreads - 1) / mbt::kBlockThreads;
  e->n_waves = e->n_blocks * (mbt::kBlockThreads / 64);
  e->dt = cfg->terminal_time / cfg->n_steps;  // TE:49
  e->mid_dt = cfg->midprice_step_size > 0.0 ? cfg->midprice_step_size : e->dt;
  e->arr_dt = cfg->arrival_step_size > 0.0 ? cfg->arrival_step_size : e->dt;
  e->imp_dt = cfg->impact_step_size > 0.0 ? cfg->impact_step_size : e->dt;
  e->seed = cfg->seed;
  e->host_mask = (host_fill ? mbt::kHostFill : 0) | (host_arrival ? mbt::kHostArrival : 0) | ((host_reward || host_lowered.speed_reward) ? mbt::kHostReward : 0) |
                 (host_impact(*cfg) ? mbt::kHostImpact : 0);
  e->host_reward_replaces = host_lowered.speed_reward;
  e->res = !cfg->precise_state ? (exact_intensities(*cfg) ? 2 : 0) : speed ? 4 : ((cfg->arrival_kind == MBT_ARR_HAWKES || e->user_state_columns > 0) ? 4 : 2);
  {
    const int order_book[4] = {0, 3, 4, 5}, speed_cols[4] = {0, 1, 3, 4}, intensities[4] = {4, 5, 4, 5};  // state columns behind the remainder columns
    for (int j = 0; j < 4; ++j) e->res_col[j] = !cfg->precise_state ? intensities[j] : speed ? speed_cols[j] : order_book[j];
  }
  tune_for_size(e);
  if (needs_jit) {
    for (int j = 0; j < 8; ++j) {
      e->user_fill_p[j] = code->fill_params[j];
      e->user_reward_p[j] = code->reward_params[j];
      e->user_arrival_p[j] = code->arrival_params[j];
      e->user_mid_p[j] = code->midprice_params[j];
      e->user_state_p[j] = code->state_params[j];
    }
    std::string source;
    JitKernels kernels;
    int jit_rc = jit_source(*cfg, *code, source);
    if (jit_rc == MBT_OK) jit_rc = jit_build(cfg->device, source, cfg->noise_mode == MBT_NOISE_PHILOX && e->host_mask == 0, kernels);
    if (jit_rc != MBT_OK) {
      delete e;
      return jit_rc;
    }
    e->jit_step = kernels.step;
    e->jit_rollout = kernels.rollout;
    e->jit_step_mirror = kernels.step_mirror;
    e->jit_step_captured = kernels.step_captured;
    e->stream_loads = false;  // one instantiation is compiled: default-policy loads
  } else {
    e->kernel = pick_kernel(*cfg, e->stream_loads ? kStream : kPlain);
    e->kernel_mirror = pick_kernel(*cfg, kMirror);
    e->kernel_captured = pick_kernel(*cfg, e->stream_loads ? mbt_table::kCapturedStream : mbt_table::kCaptured);  // (nullptr: injected noise, host callbacks)
    e->rollout = pick_rollout_kernel(*cfg);
  }
  fill_static_params(e);
  key_from_seed(e);

#define ENV_TRY(expr)          \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != MBT_OK) {       \
      mbt_env_destroy(e);      \
      return rc_;              \
    }                          \
  } while (0)
  hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  if (he != hipSuccess) {
    delete e;
    return fail(MBT_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(he));
  }
  e->own_stream = true;
  if (hipEventCreate(&e->ev_begin) != hipSuccess || hipEventCreate(&e->ev_end) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_sums, hipEventDisableTiming) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&e->h_sums), 3 * sizeof(double), hipHostMallocDefault) != hipSuccess) {
    mbt_env_destroy(e);
    return fail(MBT_ERR_HIP, "hipEventCreate failed");
  }
  const size_t np = e->n_pad;
  ENV_TRY(dev_alloc(&e->state[0], np * e->dim, e->stream));
  if (e->ping_pong) ENV_TRY(dev_alloc(&e->state[1], np * e->dim, e->stream));
  if (cfg->normalise_observation) ENV_TRY(dev_alloc(&e->obs, np * e->dim, e->stream));
  ENV_TRY(dev_alloc(&e->action, np * e->act_dim, e->stream));
  ENV_TRY(dev_alloc(&e->reward, np, e->stream));
  if (cfg->noise_mode == MBT_NOISE_INJECTED) {
    if (!speed) {
      ENV_TRY(dev_alloc(&e->u_arr, np * 2, e->stream));
      ENV_TRY(dev_alloc(&e->u_fill, np * 2, e->stream));
    }
    ENV_TRY(dev_alloc(&e->z, np, e->stream));
    if (e->user_draws) ENV_TRY(dev_alloc(&e->z_user, np * 2, e->stream));
  }
  ENV_TRY(dev_alloc(&e->q_init, np, e->stream));
  if (e->res != 0) ENV_TRY(dev_alloc(&e->resid, np * e->res, e->stream));
  ENV_TRY(dev_alloc(&e->wave_sums, e->n_waves, e->stream));
  ENV_TRY(dev_alloc(&e->clip_count, mbt::kClipSlots, e->stream));
  ENV_TRY(dev_alloc(&e->reduce_out, 3, e->stream));
  ENV_TRY(dev_alloc(&e->log_dev, 3 * mbt_env::kLogSlots, e->stream));
  ENV_TRY(dev_alloc(&e->done_counter, 32, e->stream));  // [0]: workgroups finished (signal_host); [16]: the resident kernel's control word
  // graph-capturable stepping: the clock block and the arrival counters of its launches (a 64-byte line for the top level + one per group of 32 workgroups)
  ENV_TRY(dev_alloc(&e->clock_dev, 1, e->stream));
  ENV_TRY(dev_alloc(&e->clock_counters, 16u * (1u + (e->n_blocks + 31u) / 32u), e->stream));
  if (hipHostMalloc(reinterpret_cast<void**>(&e->clock_host), sizeof(mbt::DeviceClock), hipHostMallocDefault) != hipSuccess) {
    mbt_env_destroy(e);
    return fail(MBT_ERR_HIP, "hipHostMalloc failed");
  }
  if (e->host_mask != 0) {
    if (host_fill || host_impact(*cfg)) ENV_TRY(dev_alloc(&e->host_fill_p, np * 2, e->stream));  // (N, 2) fill probabilities / (N) price impacts
    if (host_arrival) ENV_TRY(dev_alloc(&e->host_arrivals, np * 2, e->stream));
    ENV_TRY(dev_alloc(&e->host_scratch, np * 4, e->stream));  // (N) rewards or (N, d <= 3) state columns, float64
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&e->log_host), 3 * mbt_env::kLogSlots * sizeof(double), hipHostMallocDefault) != hipSuccess) {
    mbt_env_destroy(e);
    return fail(MBT_ERR_HIP, "hipHostMalloc failed");
  }
  for (uint32_t k = 0; k < mbt_env::kLogSlots; ++k)
    if (hipEventCreateWithFlags(&e->log_event[k], hipEventDisableTiming) != hipSuccess) {
      mbt_env_destroy(e);
      return fail(MBT_ERR_HIP, "hipEventCreate failed");
    }
  uint32_t fast_path_lanes = kHostFastPathLanes;
  if (const char* env_override = std::getenv("MBT_HOST_FAST_PATH_LANES")) fast_path_lanes = static_cast<uint32_t>(std::strtoul(env_override, nullptr, 10));
  if (e->n <= fast_path_lanes) {  // see step_host: zero-copy staging instead of pageable DMA copies
    e->stage_action = 0;
    e->stage_obs = np * e->act_dim;
    e->stage_reward = e->stage_obs + size_t(e->n) * e->dim;
    e->stage_flag = ((e->stage_reward + e->n + 15u) / 16u) * 16u;  // the completion flag of signal_host, on a cache line of its own
    e->stage_exit = e->stage_flag + 16u;                           // where a resident kernel says it left before a step, on the next line
    size_t floats = e->stage_exit + 16u;
    if (e->host_mask != 0 && !speed) {  // host-callback plugins: the step also mirrors rows, remainders and event bytes (launch_step)
      e->stage_state = floats;
      e->stage_resid = e->stage_state + ((size_t(e->n) * e->dim + 3u) / 4u) * 4u;
      e->stage_events = e->stage_resid + ((size_t(e->n) * e->res + 3u) / 4u) * 4u;
      floats = e->stage_events + (e->n + 3u) / 4u + 4u;
    }
    // (coherent: the host reads the flag, and then the mirror, while the GPU context is live - not after a synchronisation)
    if (hipHostMalloc(reinterpret_cast<void**>(&e->h_stage), floats * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      std::memset(e->h_stage, 0, floats * sizeof(float));
      if (hipHostGetDevicePointer(reinterpret_cast<void**>(&e->d_stage), e->h_stage, 0) != hipSuccess) {
        (void)hipHostFree(e->h_stage);
        e->h_stage = e->d_stage = nullptr;
      }
    } else {
      e->h_stage = nullptr;  // not fatal: the DMA path below serves every size
      (void)hipGetLastError();
    }
    if (e->h_stage == nullptr) e->stage_state = e->stage_resid = e->stage_events = 0;
    if (e->h_stage != nullptr && e->host_mask != 0) {
      // ... and what the caller's NumPy code computed goes DOWN through one mapped block the kernels read in place (no copy, no wait)
      const size_t fill_bytes = np * 2 * sizeof(double), arrival_bytes = np * 2 * sizeof(float), scratch_bytes = np * 4 * sizeof(double);
      char* host = nullptr;
      char* dev = nullptr;
      if (hipHostMalloc(reinterpret_cast<void**>(&host), fill_bytes + arrival_bytes + scratch_bytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
          hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0) == hipSuccess) {
        std::memset(host, 0, fill_bytes + arrival_bytes + scratch_bytes);
        for (void* b : {static_cast<void*>(e->host_fill_p), static_cast<void*>(e->host_arrivals), static_cast<void*>(e->host_scratch)})
          if (b != nullptr) (void)hipFree(b);
        e->h_callback_in = host;
        e->host_fill_p = reinterpret_cast<double*>(dev);
        e->host_arrivals = reinterpret_cast<float*>(dev + fill_bytes);
        e->host_scratch = reinterpret_cast<double*>(dev + fill_bytes + arrival_bytes);
      } else {
        if (host != nullptr) (void)hipHostFree(host);
        (void)hipGetLastError();
      }
    }
  }
  {  // opt-in: MBT_RESIDENT_STEP=1 - small batches of the float32 tier's built-in order-book models step through a kernel that stays on the device
    const char* resident = std::getenv("MBT_RESIDENT_STEP");
    const mbt_config& c = *cfg;
    const bool resident_wanted = c.resident_step != 0 || (resident != nullptr && std::atoi(resident) != 0);
    // up to 8 tiles (4096 lanes): beyond that four workgroups walking the tiles one after the other lose to a launch that steps them all
    // at once (N = 65536: 72 vs 62 us per env.step, profiles/r05_resident_step.txt)
    constexpr uint32_t kResidentMaxTiles = 8;
    if (resident_wanted && e->h_stage != nullptr && !needs_jit && !speed && !c.precise_state && !exogenous_fill(c) &&
        e->host_mask == 0 && c.noise_mode == MBT_NOISE_PHILOX && e->n_blocks <= kResidentMaxTiles) {
      const bool norm = c.normalise_action != 0 || c.normalise_observation != 0;
      e->resident_kernel = mbt_table::pick_resident(arrival_family(c), c.dynamics_kind, c.midprice_kind == MBT_MID_BROWNIAN, reward_weight(c), norm);
      if (e->resident_kernel != nullptr) ENV_TRY(resident_allocate(e));
      if (const char* v = std::getenv("MBT_RESIDENT_ANSWER_MS")) e->resident_answer_ms = std::atoi(v) > 0 ? std::atoi(v) : 0;
    }
    if (const char* v = std::getenv("MBT_TEST_FLAG_SEQ")) e->flag_seq = static_cast<uint32_t>(std::strtoul(v, nullptr, 0));  // (test hook: start the sequence numbers of the completion flag near their wrap)
    const char* vram_stage = std::getenv("MBT_VRAM_ACTION_STAGE");  // (measurement knob, see mbt_env_step_host)
    if (e->resident_kernel == nullptr && vram_stage != nullptr && std::atoi(vram_stage) != 0 && e->h_stage != nullptr) {
      ENV_TRY(resident_allocate(e));
      if (!e->resident_vram) {  // (host memory would be the existing stage again)
        (void)hipHostFree(e->resident_host_block);
        e->resident_host_block = nullptr;
      }
    }
  }
#undef ENV_TRY
  // the zero fills above are ordered on e->stream; callers may touch the buffers from other streams (action_device,
  // obs_device under PyTorch) as soon as this returns, so they must have landed
  he = hipStreamSynchronize(e->stream);
  if (he != hipSuccess) {
    mbt_env_destroy(e);
    return fail(MBT_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(he));
  }
  *out = e;
  return MBT_OK;
}

int mbt_env_create(const mbt_config* cfg, mbt_env** out) { return create_env(cfg, nullptr, out); }

int mbt_env_create_jit(const mbt_config* cfg, const mbt_user_code* code, mbt_env** out) {
  if (code == nullptr) return fail(MBT_ERR_INVALID, "null user code");
  return create_env(cfg, code, out);
}

const char* mbt_jit_log(void) { return g_jit_log.c_str(); }

int mbt_jit_check(const mbt_config* cfg, const mbt_user_code* code) {
  if (cfg == nullptr || code == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (cfg->abi_version != MBT_ABI_VERSION) return fail(MBT_ERR_ABI, "mbt_config.abi_version %u != library %u", cfg->abi_version, MBT_ABI_VERSION);
  mbt_config lowered;
  const char* refused = nullptr;
  const HostLowering host_lowered = lower_host_kinds(*cfg, lowered, &refused);
  if (refused != nullptr) return fail(MBT_ERR_INVALID, "%s", refused);
  cfg = &lowered;
  const bool user_fill = cfg->fill_kind == MBT_FILL_USER, user_reward = cfg->reward_kind == MBT_REW_USER, user_arrival = cfg->arrival_kind == MBT_ARR_USER;
  const bool user_mid = cfg->midprice_kind == MBT_MID_USER;
  const bool any_host = cfg->fill_kind == MBT_FILL_HOST || cfg->arrival_kind == MBT_ARR_HOST || cfg->reward_kind == MBT_REW_HOST;
  if (!user_fill && !user_reward && !user_arrival && !user_mid && !any_host) {
    if (host_lowered.speed_reward || host_lowered.midprice || host_impact(*cfg)) return MBT_OK;  // host callbacks on the ahead-of-time speed kernels: nothing to compile
    return fail(MBT_ERR_INVALID, "no plugin kind of the configuration names a user-defined plugin");
  }
  if (user_mid && (code->midprice_increment == nullptr || code->midprice_increment[0] == 0)) return fail(MBT_ERR_INVALID, "MBT_MID_USER without a midprice_increment expression");
  if (user_fill && (code->fill_probability == nullptr || code->fill_probability[0] == 0)) return fail(MBT_ERR_INVALID, "MBT_FILL_USER without a fill_probability expression");
  if (user_reward && (code->reward == nullptr || code->reward[0] == 0)) return fail(MBT_ERR_INVALID, "MBT_REW_USER without a reward expression");
  if (user_arrival && (code->arrival_probability == nullptr || code->arrival_probability[0] == 0)) return fail(MBT_ERR_INVALID, "MBT_ARR_USER without an arrival_probability expression");
  std::string source;
  std::vector<char> object;
  int rc = jit_source(*cfg, *code, source);
  if (rc == MBT_OK) rc = jit_compile(source, object);
  return rc;
}

void mbt_env_destroy(mbt_env* e) {
  if (e == nullptr) return;
  (void)hipSetDevice(e->cfg.device);
  if (e->stream != nullptr) (void)resident_stop(e);
  if (e->stream != nullptr) (void)hipStreamSynchronize(e->stream);
  if (e->resident_vram_block != nullptr) (void)hipFree(e->resident_vram_block);
  if (e->resident_host_block != nullptr) (void)hipHostFree(e->resident_host_block);
  void* bufs[] = {e->z_user, e->resid, e->state[0], e->state[1], e->obs,    e->action,       e->reward,    e->u_arr,      e->u_fill,
                  e->z,        e->q_init,   e->events, e->lane_returns, e->wave_sums, e->clip_count, e->reduce_out,
                  e->policy_table, e->log_dev, e->traj_stage[0], e->traj_stage[1], e->traj_stage[2], e->learned_dev};
  for (void* b : bufs)
    if (b != nullptr) (void)hipFree(b);
  if (e->log_host != nullptr) (void)hipHostFree(e->log_host);
  for (hipEvent_t ev : e->log_event)
    if (ev != nullptr) (void)hipEventDestroy(ev);
  if (e->h_callback_in != nullptr) {  // (the three callback buffers point into this one mapped block)
    (void)hipHostFree(e->h_callback_in);
    e->host_fill_p = nullptr;
    e->host_arrivals = nullptr;
    e->host_scratch = nullptr;
  }
  for (void* b : {static_cast<void*>(e->done_counter), static_cast<void*>(e->host_fill_p), static_cast<void*>(e->host_arrivals), static_cast<void*>(e->host_scratch),
                  static_cast<void*>(e->clock_dev), static_cast<void*>(e->clock_counters), static_cast<void*>(e->terminal_obs)})
    if (b != nullptr) (void)hipFree(b);
  if (e->clock_host != nullptr) (void)hipHostFree(e->clock_host);
  if (e->h_gate != nullptr) (void)hipHostFree(e->h_gate);
  if (e->h_stage != nullptr) (void)hipHostFree(e->h_stage);
  if (e->h_bounce != nullptr) (void)hipHostFree(e->h_bounce);
  if (e->ev_begin != nullptr) (void)hipEventDestroy(e->ev_begin);
  if (e->ev_end != nullptr) (void)hipEventDestroy(e->ev_end);
  if (e->ev_sums != nullptr) (void)hipEventDestroy(e->ev_sums);
  if (e->h_sums != nullptr) (void)hipHostFree(e->h_sums);
  if (e->own_stream && e->stream != nullptr) (void)hipStreamDestroy(e->stream);
  delete e;
}

int mbt_env_set_stream(mbt_env* e, void* hip_stream) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->own_stream && e->stream != nullptr) (void)hipStreamDestroy(e->stream);
  e->stream = static_cast<hipStream_t>(hip_stream);
  e->own_stream = false;
  return MBT_OK;
}

int mbt_env_synchronize(mbt_env* e) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  // A blocking wait costs a wake-up (an interrupt round trip, tens of microseconds) - more than a step at 2^20 lanes.
  // Poll first: short waits, the common case between a consumer's launches, end within a microsecond of the stream.
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(e->stream);
    if (q == hipSuccess) {
      e->callback_inputs_busy = e->callback_scratch_busy = false;
      return MBT_OK;
    }
    if (q != hipErrorNotReady) return fail(MBT_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->callback_inputs_busy = e->callback_scratch_busy = false;
  return MBT_OK;
}

int mbt_env_set_step_size(mbt_env* e, double step_size) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  if (!(step_size > 0.0)) return fail(MBT_ERR_INVALID, "step_size must be positive");
  if (e->resident_active) HIP_TRY(hipSetDevice(e->cfg.device));  // (a resident kernel holds the old step size)
  HOST_CLOCK(e);
  if (e->cfg.arrival_kind == MBT_ARR_HAWKES && !e->cfg.allow_stiff_hawkes && !(e->cfg.hawkes_speed * step_size < 1.0))  // the same domain mbt_env_create enforces
    return fail(MBT_ERR_INVALID, "Hawkes mean_reversion_speed * step_size = %g >= 1 with the new step size: the intensity recursion (ARR:110-119) "
                "oscillates (>= 2: diverges); set allow_stiff_hawkes to run it anyway", e->cfg.hawkes_speed * step_size);
  // TE:158-167: the environment's clock and every process continue with the new value; like the reference, nothing else
  // (n_steps, terminal_time, max_cash, Box bounds) is re-derived.  The kernel parameters are host-side data.
  e->dt = e->mid_dt = e->arr_dt = e->imp_dt = step_size;
  e->cfg.midprice_step_size = e->cfg.arrival_step_size = e->cfg.impact_step_size = step_size;
  fill_static_params(e);
  key_from_seed(e);
  fill_episode_params(e);
  return MBT_OK;
}

int mbt_env_seed(mbt_env* e, uint64_t seed) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  if (e->resident_active) HIP_TRY(hipSetDevice(e->cfg.device));  // (a resident kernel holds the old key)
  HOST_CLOCK(e);
  e->seed = seed;
  e->philox_step = 0;
  key_from_seed(e);
  return MBT_OK;
}

int mbt_env_reset(mbt_env* e, double start_time, const float* q0_host) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  return do_reset(e, start_time, q0_host);
}

int mbt_env_reset_host(mbt_env* e, double start_time, const float* q0_host, float* obs_host) {
  int rc = mbt_env_reset(e, start_time, q0_host);
  if (rc != MBT_OK) return rc;
  if (obs_host != nullptr)
    HIP_TRY(hipMemcpyAsync(obs_host, current_obs(e), size_t(e->n) * e->dim * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MBT_OK;
}

int mbt_env_step_host(mbt_env* e, const float* action_host, float* obs_host, float* reward_host, int32_t* done) {
  if (e == nullptr || action_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (e->device_clock) return fail(MBT_ERR_STATE, "mbt_env_step_host: the environment's clock is on the device (mbt_env_device_clock_begin): call mbt_env_device_clock_end first");
  if (e->resident_kernel != nullptr && e->h_stage != nullptr && e->gate_chunk == 0) return resident_step(e, action_host, obs_host, reward_host, done);
  if (e->h_stage != nullptr) {
    // Small batch (the reference's own regime, N ~ 1000): a DMA copy costs ~15-25 us per call whatever its size, a second
    // launch ~5 us, a blocking wait an interrupt round trip.  None of them here: the actions are read by the step kernel
    // from, and its observation rows and rewards written by it into, pinned device-mapped host memory; the last workgroup
    // to finish raises a flag there and this thread spins on it (step_kernel.hpp: signal_host).  ONE launch per env.step().
    const size_t n_obs = size_t(e->n) * e->dim;
    // (MBT_VRAM_ACTION_STAGE=1, measurement knob: the actions go into device memory through the PCIe BAR instead - the kernel then reads
    // them locally rather than across the link; profiles/r05_resident_step.txt)
    const bool vram_actions = e->resident_vram && e->resident_kernel == nullptr;
    if (vram_actions) {
      std::memcpy(e->resident_action_host, action_host, size_t(e->n) * e->act_dim * sizeof(float));
      _mm_sfence();
    } else {
      std::memcpy(e->h_stage + e->stage_action, action_host, size_t(e->n) * e->act_dim * sizeof(float));
    }
    e->action_in_stage = false;  // (set below: launch_step must not file the PREVIOUS stage contents first)
    const bool mirror = e->jit_step != nullptr ? e->jit_step_mirror != nullptr : e->kernel_mirror != nullptr;
    // the caller's arrays are blocks of mbt_host_alloc (the Python layer's output pools are): the kernel writes them directly
    float* direct_obs = mirror ? static_cast<float*>(device_alias(obs_host, n_obs * sizeof(float))) : nullptr;
    float* direct_rew = mirror ? static_cast<float*>(device_alias(reward_host, size_t(e->n) * sizeof(float))) : nullptr;
    if (direct_obs == nullptr || direct_rew == nullptr || reinterpret_cast<uintptr_t>(direct_obs) % 16 != 0) direct_obs = direct_rew = nullptr;  // (rows leave as 16-byte vectors)
    int rc = launch_step(e, vram_actions ? e->resident_action_dev : e->d_stage + e->stage_action, done, mirror, direct_obs, direct_rew);
    if (rc != MBT_OK) return rc;
    e->action_in_stage = true;
    e->action_in_resident_stage = vram_actions;
    if (!mirror) {  // (injected-noise kernels have no mirror instantiation: a second launch exports, the stream is waited for)
      const uint32_t threads = 256, blocks = static_cast<uint32_t>((n_obs + threads - 1) / threads);
      hipLaunchKernelGGL(mbt::export_step_kernel, dim3(blocks), dim3(threads), 0, e->stream, current_obs(e), e->reward, e->d_stage + e->stage_obs,
                         e->d_stage + e->stage_reward, static_cast<uint32_t>(n_obs), e->n);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(e->stream));
    } else {
      const uint32_t* flag = reinterpret_cast<const uint32_t*>(e->h_stage + e->stage_flag);
      const uint32_t want = e->flag_seq;
      const auto t0 = std::chrono::steady_clock::now();
      uint32_t spins = 0;
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) {
        if ((++spins & 1023u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
          // not a latency path any more (a clock ramp, a page migration - or a kernel that died): let the runtime wait and report
          HIP_TRY(hipStreamSynchronize(e->stream));
          if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) return fail(MBT_ERR_HIP, "the step kernel finished without raising its completion flag");
          break;
        }
      }
    }
    e->callback_inputs_busy = e->callback_scratch_busy = false;  // (the flag / the wait above: everything queued before it has run)
    e->stage_outputs_valid = mirror && e->stage_state != 0;
    if (direct_obs != nullptr) return MBT_OK;  // already where the caller wants them
    if (obs_host != nullptr) std::memcpy(obs_host, e->h_stage + e->stage_obs, n_obs * sizeof(float));
    if (reward_host != nullptr) std::memcpy(reward_host, e->h_stage + e->stage_reward, size_t(e->n) * sizeof(float));
    return MBT_OK;
  }
  // Larger batches: three DMA copies around the launch, bandwidth-bound (28 B per lane over PCIe for the AS workload).  A
  // copy engine can only read / write PINNED host memory directly; handed pageable memory the runtime stages it piecewise
  // through its own small pinned buffers (2.2 GB/s effective at 2^20 lanes, profiles/r02_host_path.json).  So: buffers from
  // mbt_host_alloc (or any pinned memory) are used as they are - the Python binding keeps its output arrays in such
  // memory and re-uses them - and pageable ones bounce through a pinned buffer of the environment's own.
  const size_t act_floats = size_t(e->n) * e->act_dim, obs_floats = size_t(e->n) * e->dim, rew_floats = e->n;
  const bool act_direct = e->memo_action.lookup(action_host, act_floats * sizeof(float));
  const bool obs_direct = obs_host == nullptr || e->memo_obs.lookup(obs_host, obs_floats * sizeof(float));
  const bool rew_direct = reward_host == nullptr || e->memo_reward.lookup(reward_host, rew_floats * sizeof(float));
  if (!(act_direct && obs_direct && rew_direct)) {
    const size_t want = act_floats + obs_floats + rew_floats;
    if (want > e->bounce_floats) {
      if (e->h_bounce != nullptr) (void)hipHostFree(e->h_bounce);
      e->h_bounce = nullptr;
      e->bounce_floats = 0;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_bounce), want * sizeof(float), hipHostMallocDefault));
      e->bounce_floats = want;
    }
  }
  float* bounce_act = e->h_bounce;
  float* bounce_obs = e->h_bounce != nullptr ? e->h_bounce + act_floats : nullptr;
  float* bounce_rew = e->h_bounce != nullptr ? e->h_bounce + act_floats + obs_floats : nullptr;
  if (!act_direct) std::memcpy(bounce_act, action_host, act_floats * sizeof(float));
  HIP_TRY(hipMemcpyAsync(e->action, act_direct ? action_host : bounce_act, act_floats * sizeof(float), hipMemcpyHostToDevice, e->stream));
  int rc = launch_step(e, nullptr, done);
  if (rc != MBT_OK) return rc;
  if (obs_host != nullptr)
    HIP_TRY(hipMemcpyAsync(obs_direct ? obs_host : bounce_obs, current_obs(e), obs_floats * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  if (reward_host != nullptr)
    HIP_TRY(hipMemcpyAsync(rew_direct ? reward_host : bounce_rew, e->reward, rew_floats * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (obs_host != nullptr && !obs_direct) std::memcpy(obs_host, bounce_obs, obs_floats * sizeof(float));
  if (reward_host != nullptr && !rew_direct) std::memcpy(reward_host, bounce_rew, rew_floats * sizeof(float));
  return MBT_OK;
}

int mbt_env_step_many_device(mbt_env* e, uint32_t k, const float* action_device, int32_t auto_reset, uint32_t* steps_done,
                             uint32_t* episodes_ended) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  if (e->cfg.noise_mode != MBT_NOISE_PHILOX) return fail(MBT_ERR_STATE, "injected noise is consumed one step at a time: use mbt_env_step_device");
  if (action_device != nullptr && e->n != e->n_pad) {  // see mbt_env_step_device: stage a caller buffer that has no pad rows, once
    HIP_TRY(hipMemcpyAsync(e->action, action_device, size_t(e->n) * e->act_dim * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    e->action_in_stage = false;  // the caller's actions are the newest now: what an earlier host step left in the stage must not be filed over them
    action_device = nullptr;
  }
  if (action_device == nullptr && e->action_in_stage) {
    const int rc_file = file_staged_action(e);
    if (rc_file != MBT_OK) return rc_file;
  }
  uint32_t steps = 0, episodes = 0, in_burst = 0;
  int rc = MBT_OK;
  while (steps < k) {
    if (e->gate_chunk != 0 && !e->gate_closed) {  // a new burst: everything up to gate_open() is enqueued behind this kernel
      e->gate_seq += 1;
      hipLaunchKernelGGL(mbt::gate_kernel, dim3(1), dim3(1), 0, e->stream, e->d_gate, e->gate_seq, 100000000ull /* 1 s of the 100 MHz clock */);
      HIP_TRY(hipGetLastError());
      e->gate_closed = true;
      in_burst = 0;
    }
    int32_t done = 0;
    rc = launch_step(e, action_device, &done);
    if (rc != MBT_OK) break;
    ++steps;
    if (done) {
      ++episodes;
      if (!auto_reset) break;
      rc = log_push(e);
      if (rc != MBT_OK) break;
      rc = do_reset(e, e->start_time, nullptr, /*reuse_q0=*/true);
      if (rc != MBT_OK) break;
    }
    if (e->gate_closed && ++in_burst >= e->gate_chunk) gate_open(e);
  }
  gate_open(e);
  if (steps_done != nullptr) *steps_done = steps;
  if (episodes_ended != nullptr) *episodes_ended = episodes;
  return rc;
}

int mbt_env_episode_log_pop(mbt_env* e, double sums[3], int32_t wait) {
  if (e == nullptr || sums == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (e->log_count == 0) return 0;
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  if (!wait) {
    const hipError_t q = hipEventQuery(e->log_event[e->log_head]);
    if (q == hipErrorNotReady) return 0;
    if (q != hipSuccess) return fail(MBT_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
  }
  return log_wait_oldest(e, sums);
}

int mbt_env_set_communicator(mbt_env* e, void* nccl_comm) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  if (nccl_comm != nullptr && !rccl().ok) return fail(MBT_ERR_HIP, "%s", rccl().why.c_str());
  e->comm = nccl_comm;
  return MBT_OK;
}

int mbt_env_allreduce_returns(mbt_env* e, void* nccl_comm, double sums[3]) {
  if (e == nullptr || nccl_comm == nullptr || sums == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  const RcclApi& api = rccl();
  if (!api.ok) return fail(MBT_ERR_HIP, "%s", api.why.c_str());
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  // h_sums (pinned) -> reduce_out (device) -> all-reduce in place -> h_sums; ordered on the environment's stream
  if (e->sums_pending) return fail(MBT_ERR_STATE, "a return-sums request is in flight: call mbt_env_return_sums_end first");
  for (int j = 0; j < 3; ++j) e->h_sums[j] = sums[j];
  HIP_TRY(hipMemcpyAsync(e->reduce_out, e->h_sums, 3 * sizeof(double), hipMemcpyHostToDevice, e->stream));
  ncclResult_t r = api.all_reduce(e->reduce_out, e->reduce_out, 3, ncclDouble, ncclSum, static_cast<ncclComm_t>(nccl_comm), e->stream);
  if (r != ncclSuccess) return rccl_fail(r, "ncclAllReduce");
  HIP_TRY(hipMemcpyAsync(e->h_sums, e->reduce_out, 3 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (int j = 0; j < 3; ++j) sums[j] = e->h_sums[j];  // an untracked second moment is NaN on its rank and stays NaN in the sum
  return MBT_OK;
}

int mbt_comm_unique_id(void* id_out) {
  if (id_out == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  const RcclApi& api = rccl();
  if (!api.ok) return fail(MBT_ERR_HIP, "%s", api.why.c_str());
  ncclUniqueId id;
  ncclResult_t r = api.get_unique_id(&id);
  if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
  std::memcpy(id_out, &id, sizeof id);
  return MBT_OK;
}

int mbt_comm_init_rank(int device, int n_ranks, const void* id_bytes, int rank, void** comm_out) {
  if (id_bytes == nullptr || comm_out == nullptr || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(MBT_ERR_INVALID, "bad argument");
  const RcclApi& api = rccl();
  if (!api.ok) return fail(MBT_ERR_HIP, "%s", api.why.c_str());
  int rc = check_device(device);
  if (rc != MBT_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof id);
  ncclComm_t comm = nullptr;
  ncclResult_t r = api.comm_init_rank(&comm, n_ranks, id, rank);
  if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
  *comm_out = comm;
  return MBT_OK;
}

int mbt_comm_count(void* comm, int* n_ranks) {
  if (comm == nullptr || n_ranks == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  const RcclApi& api = rccl();
  if (!api.ok) return fail(MBT_ERR_HIP, "%s", api.why.c_str());
  if (api.comm_count == nullptr) return fail(MBT_ERR_HIP, "librccl lacks ncclCommCount");
  ncclResult_t r = api.comm_count(static_cast<ncclComm_t>(comm), n_ranks);
  if (r != ncclSuccess) return rccl_fail(r, "ncclCommCount");
  return MBT_OK;
}

int mbt_comm_destroy(void* comm) {
  if (comm == nullptr) return MBT_OK;
  const RcclApi& api = rccl();
  if (!api.ok) return fail(MBT_ERR_HIP, "%s", api.why.c_str());
  ncclResult_t r = api.comm_destroy(static_cast<ncclComm_t>(comm));
  if (r != ncclSuccess) return rccl_fail(r, "ncclCommDestroy");
  return MBT_OK;
}

int mbt_env_set_launch_gate(mbt_env* e, uint32_t burst) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  if (burst > 4096u) return fail(MBT_ERR_INVALID, "a burst of more than 4096 launches may not fit the hardware queue behind a closed gate");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  if (burst != 0 && e->h_gate == nullptr) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_gate), 64, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(e->h_gate, 0, 64);
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->d_gate), e->h_gate, 0));
  }
  gate_open(e);
  e->gate_chunk = burst;
  return MBT_OK;
}

// ---- host-callback plugins: the device side of what surrounds the caller's NumPy code (Variant::HOST) -------------------
int mbt_env_host_depths(mbt_env* e, const float* action_host, double* depths_host) {
  if (e == nullptr || action_host == nullptr || depths_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (e->speed || e->cfg.dynamics_kind == MBT_DYN_AT_THE_TOUCH) return fail(MBT_ERR_INVALID, "only limit-order dynamics quote depths (MD:104-106)");
  if (e->host_scratch == nullptr) return fail(MBT_ERR_STATE, "this environment has no host-callback plugin (MBT_FILL_HOST / MBT_ARR_HOST / MBT_REW_HOST)");
  // The quotes the caller's _get_fill_probabilities(depths) is asked about are an affine map of the ACTION the caller has just handed
  // over (TE:124) - host data in, host data out.  Rounds 4's version sent it through the device and back (two copies, a launch, a
  // wait: ~45 us at N = 1000); this is the same arithmetic - float32 action and float32 Box bounds promoted, (a + 1) * gradient + low
  // in double, one rounding per operation (the library is built with -ffp-contract=off on both sides) - as decide<V>() evaluates it
  // in the kernel, so the depths are the kernel's to the bit (tests/test_gpu_host_callbacks.py compares them with mbt_env_action... ).
  const mbt::StepParams& P = e->params;
  const int a = e->act_dim;
  for (size_t i = 0; i < e->n; ++i)
    for (int side = 0; side < 2; ++side) {
      const double x = action_host[i * a + side];
      depths_host[i * 2 + side] = P.norm_act ? (x + 1.0) * static_cast<double>(P.act_grad[side]) + static_cast<double>(P.act_lo[side]) : x;
    }
  return MBT_OK;
}

int mbt_env_set_host_fill_probabilities(mbt_env* e, const double* probabilities_host) {
  if (e == nullptr || probabilities_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (!(e->host_mask & mbt::kHostFill)) return fail(MBT_ERR_STATE, "the fill model of this environment is not a host callback (MBT_FILL_HOST)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  if (e->h_callback_in != nullptr) {  // small batches: the kernel reads the block in place (the previous step has finished: its flag was waited for)
    SETTLE_CALLBACK_BLOCK(e, callback_inputs_busy);  // (... unless it was a device-API step)
    std::memcpy(e->h_callback_in, probabilities_host, size_t(e->n) * 2 * sizeof(double));
  } else {
    HIP_TRY(hipMemcpyAsync(e->host_fill_p, probabilities_host, size_t(e->n) * 2 * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  e->host_fill_ready = true;
  return MBT_OK;
}

int mbt_env_set_host_impacts(mbt_env* e, const double* impacts_host) {
  if (e == nullptr || impacts_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (!(e->host_mask & mbt::kHostImpact)) return fail(MBT_ERR_STATE, "the price impact model of this environment is not a host callback (MBT_IMPACT_HOST)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  if (e->h_callback_in != nullptr) {
    SETTLE_CALLBACK_BLOCK(e, callback_inputs_busy);
    std::memcpy(e->h_callback_in, impacts_host, size_t(e->n) * sizeof(double));
  } else {
    HIP_TRY(hipMemcpyAsync(e->host_fill_p, impacts_host, size_t(e->n) * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));  // the caller's array may go away
  }
  e->host_fill_ready = true;
  return MBT_OK;
}

int mbt_env_set_host_arrivals(mbt_env* e, const float* arrivals_host) {
  if (e == nullptr || arrivals_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (!(e->host_mask & mbt::kHostArrival)) return fail(MBT_ERR_STATE, "the arrival model of this environment is not a host callback (MBT_ARR_HOST)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  if (e->h_callback_in != nullptr) {
    SETTLE_CALLBACK_BLOCK(e, callback_inputs_busy);
    std::memcpy(e->h_callback_in + size_t(e->n_pad) * 2 * sizeof(double), arrivals_host, size_t(e->n) * 2 * sizeof(float));
  } else {
    HIP_TRY(hipMemcpyAsync(e->host_arrivals, arrivals_host, size_t(e->n) * 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  e->host_arrivals_ready = true;
  return MBT_OK;
}

int mbt_env_set_host_state_columns(mbt_env* e, const double* columns_host) {
  if (e == nullptr || columns_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (e->host_state_count == 0)
    return fail(MBT_ERR_STATE, "this environment has no host-callback process that owns state columns (MBT_MID_HOST, or MBT_ARR_HOST with state)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  const int d = e->host_state_count;
  e->stage_outputs_valid = false;  // (the rows change under the mirror)
  if (e->h_callback_in != nullptr) {  // (the scratch block is mapped host memory: the kernel reads the values in place)
    HIP_TRY(hipStreamSynchronize(e->stream));  // a reward-filing kernel of this step may still be reading the block
    std::memcpy(e->h_callback_in + size_t(e->n_pad) * (2 * sizeof(double) + 2 * sizeof(float)), columns_host, size_t(e->n) * d * sizeof(double));
  } else {
    HIP_TRY(hipMemcpyAsync(e->host_scratch, columns_host, size_t(e->n) * d * sizeof(double), hipMemcpyHostToDevice, e->stream));
  }
  hipLaunchKernelGGL(mbt::host_columns_kernel, dim3((e->n + 255u) / 256u), dim3(256), 0, e->stream, e->host_scratch, e->n, d, e->dim, e->host_state_first,
                     e->state[e->cur], e->cfg.precise_state ? e->resid : nullptr, e->res, e->speed ? 1 : 0, e->cfg.normalise_observation ? e->obs : nullptr, e->params);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MBT_OK;
}

int mbt_env_set_host_rewards(mbt_env* e, const double* rewards_host, float* reward_out_host) {
  if (e == nullptr || rewards_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (!(e->host_mask & mbt::kHostReward)) return fail(MBT_ERR_STATE, "the reward function of this environment is not a host callback (MBT_REW_HOST)");
  if (!e->host_reward_pending) return fail(MBT_ERR_STATE, "no step is waiting for its host-computed rewards");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  const uint32_t blocks = (e->n + 255u) / 256u;
  if (e->h_callback_in != nullptr) {
    // small batches: the filing kernel reads the caller's values in place and nobody waits for it - what it files is a function of
    // values the host holds (float32(scale * r), exactly host_reward_kernel's expression), so the caller's copy is formed here
    double* scratch = reinterpret_cast<double*>(e->h_callback_in + size_t(e->n_pad) * (2 * sizeof(double) + 2 * sizeof(float)));
    SETTLE_CALLBACK_BLOCK(e, callback_scratch_busy);  // (the previous step's filing kernel, if nothing waited since)
    std::memcpy(scratch, rewards_host, size_t(e->n) * sizeof(double));
    hipLaunchKernelGGL(mbt::host_reward_kernel, dim3(blocks), dim3(256), 0, e->stream, e->host_scratch, e->cfg.reward_scale, e->n, e->reward,
                       e->track_returns ? e->lane_returns : nullptr, e->wave_sums, e->n_waves, e->host_reward_replaces ? 1 : 0);
    HIP_TRY(hipGetLastError());
    if (reward_out_host != nullptr)
      for (size_t i = 0; i < e->n; ++i) reward_out_host[i] = static_cast<float>(e->cfg.reward_scale * rewards_host[i]);
    e->callback_scratch_busy = true;
    e->host_reward_pending = false;
    return MBT_OK;
  }
  HIP_TRY(hipMemcpyAsync(e->host_scratch, rewards_host, size_t(e->n) * sizeof(double), hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(mbt::host_reward_kernel, dim3(blocks), dim3(256), 0, e->stream, e->host_scratch, e->cfg.reward_scale, e->n, e->reward,
                     e->track_returns ? e->lane_returns : nullptr, e->wave_sums, e->n_waves, e->host_reward_replaces ? 1 : 0);
  HIP_TRY(hipGetLastError());
  if (reward_out_host != nullptr) HIP_TRY(hipMemcpyAsync(reward_out_host, e->reward, size_t(e->n) * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->host_reward_pending = false;
  return MBT_OK;
}

int mbt_env_host_step_outputs(mbt_env* e, double* state_host, uint8_t* events_host) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  if (!e->stage_outputs_valid) return fail(MBT_ERR_STATE, "no mirrored outputs: the last call was not a small-batch mbt_env_step_host of a host-callback environment");
  const size_t n = e->n, d = static_cast<size_t>(e->dim), r = static_cast<size_t>(e->res);
  const float* rows = e->h_stage + e->stage_state;
  const int32_t* lo = reinterpret_cast<const int32_t*>(e->h_stage + e->stage_resid);
  if (state_host != nullptr) {  // (the same assembly as mbt_env_get_state_f64_host, from the mirror instead of two device copies)
    for (size_t i = 0; i < n * d; ++i) state_host[i] = rows[i];
    for (size_t i = 0; i < n; ++i) {
      for (size_t j = 0; j < r; ++j) {
        const size_t column = static_cast<size_t>(e->res_col[j]);
        if (column < d) state_host[i * d + column] = exact_join_host(rows[i * d + column], lo[i * r + j]);
      }
      state_host[i * d + 2] = e->time;
    }
  }
  if (events_host != nullptr) std::memcpy(events_host, e->h_stage + e->stage_events, n);
  return MBT_OK;
}

int mbt_env_step_device(mbt_env* e, const float* action_device, int32_t* done) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  if (action_device != nullptr && e->n != e->n_pad) {
    // the kernel reads actions in pairs of rows: stage a caller buffer that has no pad row
    HIP_TRY(hipMemcpyAsync(e->action, action_device, size_t(e->n) * e->act_dim * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    e->action_in_stage = false;  // (as in mbt_env_step_many_device: these are the newest actions, not the stage's)
    action_device = nullptr;
  }
  return launch_step(e, action_device, done);
}

// ---- graph-capturable stepping: the clock on the device (step_kernel.hpp: captured_step_kernel) ----------------------------------
static int clock_download(mbt_env* e) {
  HIP_TRY(hipMemcpyAsync(e->clock_host, e->clock_dev, sizeof(mbt::DeviceClock), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->clock_host->faults != 0)  // (captured_epilogue's wall-clock bound: a launch lost a workgroup - the state is not to be trusted)
    return fail(MBT_ERR_HIP, "%u captured step launch(es) gave up waiting for their workgroups after 2 s: the device is not healthy, the environment's state is undefined",
                e->clock_host->faults);
  return MBT_OK;
}

int mbt_env_device_clock_begin(mbt_env* e, uint32_t flags) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  if (flags & ~(MBT_CLOCK_AUTO_RESET | MBT_CLOCK_TERMINAL_OBSERVATION)) return fail(MBT_ERR_INVALID, "unknown flag bits 0x%x", flags);
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);  // (one begin per end)
  if (!e->was_reset) return fail(MBT_ERR_STATE, "mbt_env_device_clock_begin before reset()");
  if (e->cfg.noise_mode != MBT_NOISE_PHILOX) return fail(MBT_ERR_STATE, "injected noise is handed over by the host before every step: such an environment cannot step from a graph");
  if (e->host_mask != 0) return fail(MBT_ERR_INVALID, "host-callback plugins (NumPy-only subclasses) are consulted between launches: such an environment cannot step from a graph");
  if (e->gate_chunk != 0) return fail(MBT_ERR_STATE, "the launch gate (mbt_env_set_launch_gate) belongs to mbt_env_step_many_device");
  if (e->jit_step != nullptr ? e->jit_step_captured == nullptr : e->kernel_captured == nullptr)
    return fail(MBT_ERR_INVALID, "this configuration has no graph-capturable step kernel");
  if ((flags & MBT_CLOCK_TERMINAL_OBSERVATION) && !(flags & MBT_CLOCK_AUTO_RESET))
    return fail(MBT_ERR_INVALID, "MBT_CLOCK_TERMINAL_OBSERVATION keeps what an automatic reset would overwrite: it needs MBT_CLOCK_AUTO_RESET");
  const int rc_file = file_staged_action(e);  // (the newest actions of an earlier host step belong in the action buffer the launches will read)
  if (rc_file != MBT_OK) return rc_file;
  if ((flags & MBT_CLOCK_TERMINAL_OBSERVATION) && e->terminal_obs == nullptr) {
    const int rc = dev_alloc(&e->terminal_obs, size_t(e->n_pad) * e->dim, e->stream);
    if (rc != MBT_OK) return rc;
  }
  std::memset(e->clock_host, 0, sizeof(mbt::DeviceClock));
  e->clock_host->slot[0].time = e->time;
  e->clock_host->slot[0].episode_step = e->episode_step;
  e->clock_host->slot[0].philox_step = e->philox_step;
  e->clock_host->shown = e->clock_host->slot[0];  // (current = 0)
  HIP_TRY(hipMemcpyAsync(e->clock_dev, e->clock_host, sizeof(mbt::DeviceClock), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemsetAsync(e->clock_counters, 0, 16u * (1u + (e->n_blocks + 31u) / 32u) * sizeof(uint32_t), e->stream));  // (armed: a launch leaves them at zero)
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->clock_auto_reset = (flags & MBT_CLOCK_AUTO_RESET) != 0;
  e->clock_terminal_obs = (flags & MBT_CLOCK_TERMINAL_OBSERVATION) != 0;
  e->device_clock = true;
  e->capture_open = false;
  return MBT_OK;
}

// Nothing but launches on the environment's stream, with arguments that do not depend on the step: what a stream capture records
// is valid for every replay.  The clock slot a launch reads (step_kernel.hpp: CapturedParams::parity) alternates from call to call
// WITHIN one stream capture, whose first call is preceded by the align kernel (the current slot becomes slot 0) - as is every call
// outside a capture: whatever ran before (another graph, a single call), the launch behind an align kernel reads slot 0.
int mbt_env_step_device_captured(mbt_env* e, const float* action_device) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  if (!e->device_clock) return fail(MBT_ERR_STATE, "mbt_env_step_device_captured outside mbt_env_device_clock_begin ... _end");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  unsigned long long capture_id = 0;
  HIP_TRY(hipStreamGetCaptureInfo(e->stream, &capture, &capture_id));
  const bool capturing = capture == hipStreamCaptureStatusActive;
  if (!capturing || !e->capture_open || capture_id != e->capture_id) {
    hipLaunchKernelGGL(mbt::captured_align_kernel, dim3(1), dim3(1), 0, e->stream, e->clock_dev);
    HIP_TRY(hipGetLastError());
    e->capture_parity = 0;
  }
  e->capture_open = capturing;
  e->capture_id = capture_id;
  if (action_device != nullptr && e->n != e->n_pad) {  // see mbt_env_step_device: stage a caller buffer that has no pad rows (a copy node in a capture)
    HIP_TRY(hipMemcpyAsync(e->action, action_device, size_t(e->n) * e->act_dim * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    action_device = nullptr;
  }
  mbt::StepBuffers B;
  std::memset(&B, 0, sizeof B);
  B.state_in = B.state_out = e->state[e->cur];  // in place: the observation a captured policy reads has ONE address
  B.action = action_device != nullptr ? action_device : e->action;
  B.reward = e->reward;
  B.obs = e->cfg.normalise_observation ? e->obs : nullptr;
  B.q_init = e->q_init_per_lane ? e->q_init : nullptr;
  B.resid = e->resid;
  B.events = e->record_events ? e->events : nullptr;
  B.lane_returns = e->track_returns ? e->lane_returns : nullptr;
  B.wave_sums = e->wave_sums;
  B.clip_count = e->clip_count;
  mbt::StepParams P = e->params;  // (philox_step, is_terminal, t_next, t_now: set by the kernel from the device's clock)
  P.philox_step = 0;
  P.is_terminal = 0;
  mbt::CapturedParams X;
  std::memset(&X, 0, sizeof X);
  X.clock = e->clock_dev;
  X.counters = e->clock_counters;
  X.parity = e->capture_parity;
  e->capture_parity ^= 1u;
  X.dt_f64 = e->dt;
  X.terminal_time = e->cfg.terminal_time;
  X.t_start = e->start_time;
  X.auto_reset = e->clock_auto_reset ? 1 : 0;
  X.dim = e->dim;
  X.tile_lanes = e->speed ? mbt::kSpeedTileLanes : mbt::kTileLanes;
  X.n_waves = e->n_waves;
  X.obs = B.obs;
  X.terminal_obs = e->clock_terminal_obs ? e->terminal_obs : nullptr;
  X.q0 = e->q0_per_lane_reset ? e->q_init : nullptr;
  X.row0 = make_reset_row(e, e->start_time, e->q0_per_lane_reset);
  if (e->jit_step != nullptr) {
    void* args[] = {&B, &P, &X};
    HIP_TRY(hipModuleLaunchKernel(e->jit_step_captured, e->n_blocks, 1, 1, mbt::kBlockThreads, 1, 1, e->step_dynamic_lds, e->stream, args, nullptr));
  } else {
    hipLaunchKernelGGL(mbt_table::as_captured_kernel(e->kernel_captured), dim3(e->n_blocks), dim3(mbt::kBlockThreads), e->step_dynamic_lds, e->stream, B, P, X);
    HIP_TRY(hipGetLastError());
  }
  return MBT_OK;
}

int mbt_env_device_clock_read(mbt_env* e, mbt_device_clock* out) {
  if (e == nullptr || out == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (!e->device_clock) return fail(MBT_ERR_STATE, "the environment's clock is on the host (mbt_env_get_clock)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int rc = clock_download(e);
  if (rc != MBT_OK) return rc;
  static_assert(sizeof(mbt_device_clock) == sizeof(mbt::ClockSlot) && offsetof(mbt::DeviceClock, shown) == 0 && offsetof(mbt_device_clock, log_count) == offsetof(mbt::ClockSlot, reserved),
                "struct mbt_device_clock is the clock block's first 32 bytes");
  std::memcpy(out, &e->clock_host->shown, sizeof *out);
  return MBT_OK;
}

void* mbt_env_device_clock_ptr(mbt_env* e) { return e != nullptr ? e->clock_dev : nullptr; }
float* mbt_env_terminal_obs_ptr(mbt_env* e) { return e != nullptr ? e->terminal_obs : nullptr; }

int mbt_env_device_clock_end(mbt_env* e) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  if (!e->device_clock) return MBT_OK;
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int rc = clock_download(e);
  if (rc != MBT_OK) {
    e->device_clock = false;  // (the mode ends either way: the caller resets)
    e->was_reset = false;
    return rc;
  }
  const mbt::DeviceClock& block = *e->clock_host;
  const mbt::ClockSlot& now = block.slot[block.current & 1u];
  e->time = now.time;
  e->episode_step = now.episode_step;
  e->philox_step = now.philox_step;
  e->device_clock = false;
  // the episodes that ended in the mode, oldest first, into the episode log (mbt_env_episode_log_pop) - through the communicator, if
  // one is set: every rank replays the same graph, so every rank files the same number of entries
  const uint32_t logged = block.shown.reserved, kept = logged < mbt::kClockLogSlots ? logged : mbt::kClockLogSlots;
  for (uint32_t k = logged - kept; k != logged; ++k) {
    if (e->log_count == mbt_env::kLogSlots) {
      double dropped[3];
      const int rc_drop = log_wait_oldest(e, dropped);
      if (rc_drop < 0) return rc_drop;
    }
    const uint32_t slot = (e->log_head + e->log_count) % mbt_env::kLogSlots;
    double* dev = e->log_dev + 3 * slot;
    HIP_TRY(hipMemcpyAsync(dev, &e->clock_dev->log[k % mbt::kClockLogSlots][0], 3 * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    if (e->comm != nullptr) {
      const RcclApi& api = rccl();
      if (!api.ok) return fail(MBT_ERR_HIP, "%s", api.why.c_str());
      ncclResult_t r = api.all_reduce(dev, dev, 3, ncclDouble, ncclSum, static_cast<ncclComm_t>(e->comm), e->stream);
      if (r != ncclSuccess) return rccl_fail(r, "ncclAllReduce");
    }
    HIP_TRY(hipMemcpyAsync(e->log_host + 3 * slot, dev, 3 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipEventRecord(e->log_event[slot], e->stream));
    e->log_count += 1;
  }
  return MBT_OK;
}

int mbt_env_policy_device(mbt_env* e, const mbt_policy* policy) {
  if (e == nullptr || policy == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (policy->kind != MBT_POLICY_LINEAR && policy->kind != MBT_POLICY_MLP) return fail(MBT_ERR_INVALID, "mbt_env_policy_device evaluates learned policies (MBT_POLICY_LINEAR / MBT_POLICY_MLP)");
  if (!e->was_reset) return fail(MBT_ERR_STATE, "policy evaluation before reset()");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  mbt::LearnedPolicyParams LP;
  int rc = prepare_learned_policy(e, policy, LP);
  if (rc != MBT_OK) return rc;
  hipLaunchKernelGGL(mbt::policy_kernel, dim3(policy_blocks(e)), dim3(mbt::kBlockThreads), 0, e->stream, current_obs(e), e->action, e->dim, e->act_dim, LP,
                     e->params.pair_offset, e->philox_step, e->params.key0, e->params.key1);
  HIP_TRY(hipGetLastError());
  e->action_in_stage = false;  // the policy's actions are the newest now
  return MBT_OK;
}

int mbt_env_record_floor_device(mbt_env* e, uint32_t steps, float* obs_traj, float* act_traj, float* rew_traj) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  if (e->speed || e->dim != 4) return fail(MBT_ERR_INVALID, "the recording floor is that of 16-byte observation rows (order-book dynamics without process columns)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  uint32_t dynamic_lds = 32u * 1024u;  // what a recording rollout is launched with (launch_rollout)
  if (const char* v = std::getenv("MBT_ROLLOUT_DYNAMIC_LDS")) dynamic_lds = static_cast<uint32_t>(std::strtoul(v, nullptr, 10));
  hipLaunchKernelGGL(mbt::record_floor_kernel, dim3(e->n_blocks), dim3(mbt::kBlockThreads), dynamic_lds, e->stream, obs_traj, act_traj, rew_traj, e->n_pad, steps, e->dim, e->act_dim);
  HIP_TRY(hipGetLastError());
  return MBT_OK;
}

uint64_t mbt_env_padded_lanes(mbt_env* e) { return e != nullptr ? e->n_pad : 0; }

int mbt_env_rollout_device(mbt_env* e, const mbt_policy* policy, uint32_t max_steps, float* obs_traj, float* act_traj,
                           float* rew_traj, uint32_t* steps_done, int32_t* done) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  return launch_rollout(e, policy, max_steps, obs_traj, act_traj, rew_traj, steps_done, done);
}

int mbt_env_rollout_host(mbt_env* e, const mbt_policy* policy, uint32_t max_steps, float* obs_traj, float* act_traj,
                         float* rew_traj, uint32_t* steps_done, int32_t* done) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  // staging in HBM, sized by the number of steps that will actually run (at most until the episode ends); the buffers
  // are kept by the environment (grow-only, up to 8 GiB each), so a consumer that records every episode allocates once
  const uint32_t remaining = static_cast<uint32_t>(std::ceil((e->cfg.terminal_time - e->time) / e->dt)) + 1;
  const size_t k_max = max_steps < remaining ? max_steps : remaining;
  const size_t np = e->n_pad;
  const size_t want[3] = {obs_traj != nullptr ? (k_max + 1) * np * e->dim : 0, act_traj != nullptr ? k_max * np * e->act_dim : 0,
                          rew_traj != nullptr ? k_max * np : 0};
  for (int j = 0; j < 3; ++j) {
    if (want[j] <= e->traj_stage_floats[j]) continue;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->traj_stage[j] != nullptr) (void)hipFree(e->traj_stage[j]);
    e->traj_stage[j] = nullptr;
    e->traj_stage_floats[j] = 0;
    if (hipMalloc(reinterpret_cast<void**>(&e->traj_stage[j]), want[j] * sizeof(float)) != hipSuccess) {
      (void)hipGetLastError();
      return fail(MBT_ERR_HIP, "out of device memory for the trajectory staging buffers (%zu MB)", want[j] * sizeof(float) >> 20);
    }
    e->traj_stage_floats[j] = want[j];
  }
  float* d_obs = obs_traj != nullptr ? e->traj_stage[0] : nullptr;
  float* d_act = act_traj != nullptr ? e->traj_stage[1] : nullptr;
  float* d_rew = rew_traj != nullptr ? e->traj_stage[2] : nullptr;
  uint32_t steps = 0;
  int rc = launch_rollout(e, policy, max_steps, d_obs, d_act, d_rew, &steps, done);
  if (rc == MBT_OK) {
    // compact the padded time slices (n_pad lanes) into the caller's (n lanes) with strided copies
    const size_t n = e->n;
    hipError_t he = hipSuccess;
    // (a batch that fills its tiles has no pad rows: one linear copy per array - the 2-D form is the slower path of the runtime)
    auto copy_out = [&](float* dst, const float* src, size_t width, size_t rows) {
      if (np == n) return hipMemcpyAsync(dst, src, rows * n * width * sizeof(float), hipMemcpyDeviceToHost, e->stream);
      return hipMemcpy2DAsync(dst, n * width * sizeof(float), src, np * width * sizeof(float), n * width * sizeof(float), rows, hipMemcpyDeviceToHost, e->stream);
    };
    if (d_obs != nullptr) he = copy_out(obs_traj, d_obs, e->dim, steps + 1);
    if (he == hipSuccess && d_act != nullptr) he = copy_out(act_traj, d_act, e->act_dim, steps);
    if (he == hipSuccess && d_rew != nullptr) he = copy_out(rew_traj, d_rew, 1, steps);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (he != hipSuccess) rc = fail(MBT_ERR_HIP, "trajectory copy failed: %s", hipGetErrorString(he));
  }
  for (int j = 0; j < 3; ++j)  // what is kept between calls is bounded (8 GiB per array of the 288 GB: a 2^20-lane, 200-step recording
                               // stays - allocating and freeing its 5.9 GB every episode cost as much as copying it out)
    if (e->traj_stage_floats[j] * sizeof(float) > (size_t(8) << 30)) {
      (void)hipStreamSynchronize(e->stream);
      (void)hipFree(e->traj_stage[j]);
      e->traj_stage[j] = nullptr;
      e->traj_stage_floats[j] = 0;
    }
  if (steps_done != nullptr) *steps_done = steps;
  return rc;
}

int mbt_env_release_staging(mbt_env* e) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (int j = 0; j < 3; ++j) {
    if (e->traj_stage[j] != nullptr) (void)hipFree(e->traj_stage[j]);
    e->traj_stage[j] = nullptr;
    e->traj_stage_floats[j] = 0;
  }
  if (e->h_bounce != nullptr) (void)hipHostFree(e->h_bounce);
  e->h_bounce = nullptr;
  e->bounce_floats = 0;
  return MBT_OK;
}

int mbt_env_set_user_noise_host(mbt_env* e, const float* z_user) {
  if (e == nullptr || z_user == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (e->cfg.noise_mode != MBT_NOISE_INJECTED) return fail(MBT_ERR_STATE, "environment was not created in injected-noise mode");
  if (!e->user_draws || e->z_user == nullptr) return fail(MBT_ERR_STATE, "this environment's user processes draw no extra normals (mbt_user_code.extra_normals)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  HIP_TRY(hipMemcpyAsync(e->z_user, z_user, size_t(e->n) * 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->user_noise_ready = true;
  return MBT_OK;
}

int mbt_env_set_noise_host(mbt_env* e, const float* u_arr, const float* u_fill, const float* z) {
  if (e == nullptr || z == nullptr || (!e->speed && (u_arr == nullptr || u_fill == nullptr))) return fail(MBT_ERR_INVALID, "null argument");
  if (e->cfg.noise_mode != MBT_NOISE_INJECTED) return fail(MBT_ERR_STATE, "environment was not created in injected-noise mode");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  if (!e->speed) {
    HIP_TRY(hipMemcpyAsync(e->u_arr, u_arr, size_t(e->n) * 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->u_fill, u_fill, size_t(e->n) * 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
  }
  HIP_TRY(hipMemcpyAsync(e->z, z, size_t(e->n) * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->noise_ready = true;
  return MBT_OK;
}

float* mbt_env_action_ptr(mbt_env* e) {
  if (e == nullptr) return nullptr;
  if (e->action_in_stage && hipSetDevice(e->cfg.device) == hipSuccess) (void)file_staged_action(e);  // a reader finds the newest actions
  return e->action;
}
float* mbt_env_obs_ptr(mbt_env* e) { return e != nullptr ? current_obs(e) : nullptr; }
int mbt_env_state_in_place(mbt_env* e) { return e != nullptr && (!e->ping_pong || e->cfg.normalise_observation != 0 || e->device_clock) ? 1 : 0; }
float* mbt_env_reward_ptr(mbt_env* e) {
  if (e == nullptr) return nullptr;
  // host-computed rewards of a small batch are filed by a kernel nobody waited for (mbt_env_set_host_rewards): a reader on another
  // stream - this getter is its way in - finds them filed, as it did when that call still waited
  if (e->callback_scratch_busy && hipSetDevice(e->cfg.device) == hipSuccess) (void)settle_callback_block(e, e->callback_scratch_busy);
  return e->reward;
}
int mbt_env_obs_dim(mbt_env* e) { return e != nullptr ? e->dim : 0; }
int mbt_env_action_dim(mbt_env* e) { return e != nullptr ? e->act_dim : 0; }

int mbt_env_get_state_host(mbt_env* e, float* state_host) {
  if (e == nullptr || state_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  HIP_TRY(hipMemcpyAsync(state_host, e->state[e->cur], size_t(e->n) * e->dim * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MBT_OK;
}

void* mbt_host_alloc(size_t bytes) {
  if (bytes == 0) return nullptr;
  void* p = nullptr;
  // (mapped + coherent: besides being DMA-able the block can be written by a kernel and read by the host while the kernel's
  // stream is live - what lets the small-batch step kernel put observations and rewards straight into the caller's arrays)
  if (hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    (void)hipGetLastError();
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      fail(MBT_ERR_HIP, "hipHostMalloc of %zu bytes failed", bytes);
      return nullptr;
    }
  }
  PinnedBlock block;
  block.bytes = bytes;
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, p, 0) == hipSuccess) block.device_base = reinterpret_cast<uintptr_t>(dev);
  else (void)hipGetLastError();
  std::lock_guard<std::mutex> guard(g_pinned_mutex);
  g_pinned_blocks[reinterpret_cast<uintptr_t>(p)] = block;
  g_pinned_generation.fetch_add(1, std::memory_order_acq_rel);
  return p;
}

void mbt_host_free(void* p) {
  if (p == nullptr) return;
  {
    std::lock_guard<std::mutex> guard(g_pinned_mutex);
    g_pinned_blocks.erase(reinterpret_cast<uintptr_t>(p));
  }
  g_pinned_generation.fetch_add(1, std::memory_order_acq_rel);
  (void)hipHostFree(p);
}

void mbt_exact_split(double x, float* hi, int32_t* lo) {
  float h = 0.0f;
  int32_t l = 0;
  exact_split_host(x, h, l);
  if (hi != nullptr) *hi = h;
  if (lo != nullptr) *lo = l;
}

double mbt_exact_join(float hi, int32_t lo) { return exact_join_host(hi, lo); }

int mbt_power_f32_device(int device, const float* x_host, double p, float* out_host, uint32_t n) {
  if (x_host == nullptr || out_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (n == 0) return MBT_OK;
  HIP_TRY(hipSetDevice(device));
  float *x = nullptr, *out = nullptr;
  HIP_TRY(hipMalloc(&x, size_t(n) * sizeof(float)));
  hipError_t err = hipMalloc(&out, size_t(n) * sizeof(float));
  if (err == hipSuccess) err = hipMemcpy(x, x_host, size_t(n) * sizeof(float), hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    hipLaunchKernelGGL(mbt::power_f32_kernel, dim3((n + mbt::kBlockThreads - 1) / mbt::kBlockThreads), dim3(mbt::kBlockThreads), 0, nullptr, x, p, out, n);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipMemcpy(out_host, out, size_t(n) * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(x);
  (void)hipFree(out);
  if (err != hipSuccess) return fail(MBT_ERR_HIP, "mbt_power_f32_device: %s", hipGetErrorString(err));
  return MBT_OK;
}

int mbt_env_get_state_f64_host(mbt_env* e, double* state_host) {
  if (e == nullptr || state_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  const size_t n = e->n, d = static_cast<size_t>(e->dim), r = static_cast<size_t>(e->res);
  std::vector<float> rows(n * d);
  std::vector<int32_t> lo(n * r);
  HIP_TRY(hipMemcpyAsync(rows.data(), e->state[e->cur], rows.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  if (r != 0) HIP_TRY(hipMemcpyAsync(lo.data(), e->resid, lo.size() * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (size_t i = 0; i < n * d; ++i) state_host[i] = rows[i];
  if (r != 0) {
    for (size_t i = 0; i < n; ++i)
      for (size_t j = 0; j < r; ++j) {
        const size_t column = static_cast<size_t>(e->res_col[j]);
        if (column < d) state_host[i * d + column] = exact_join_host(rows[i * d + column], lo[i * r + j]);
      }
  }
  for (size_t i = 0; i < n; ++i) state_host[i * d + 2] = e->time;  // the clock is kept in double on the host (TE:216) in either tier; every lane shares it
  return MBT_OK;
}

int mbt_env_get_obs_host(mbt_env* e, float* obs_host) {
  if (e == nullptr || obs_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  HIP_TRY(hipMemcpyAsync(obs_host, current_obs(e), size_t(e->n) * e->dim * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MBT_OK;
}

int mbt_env_set_action_host(mbt_env* e, const float* action_host) {
  if (e == nullptr || action_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  HIP_TRY(hipMemcpyAsync(e->action, action_host, size_t(e->n) * e->act_dim * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->action_in_stage = false;  // these are the newest actions now
  return MBT_OK;
}

int mbt_env_set_state_host(mbt_env* e, const float* state_host, double time, uint32_t philox_step) {
  if (e == nullptr || state_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  HIP_TRY(hipMemcpyAsync(e->state[e->cur], state_host, size_t(e->n) * e->dim * sizeof(float), hipMemcpyHostToDevice, e->stream));
  if (e->resid != nullptr) HIP_TRY(hipMemsetAsync(e->resid, 0, size_t(e->n_pad) * e->res * sizeof(int32_t), e->stream));  // a float32 state has no remainder
  if (e->cfg.normalise_observation) {
    const uint32_t threads = 256, blocks = (e->n_pad + threads - 1) / threads;
    hipLaunchKernelGGL(mbt::normalise_rows_kernel, dim3(blocks), dim3(threads), 0, e->stream, e->state[e->cur], e->obs,
                       e->n_pad, e->dim, e->params, e->cfg.precise_state ? 1 : 0);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->time = time;
  e->philox_step = philox_step;
  e->episode_step = static_cast<uint32_t>(std::llround((time - e->start_time) / e->dt));
  e->was_reset = true;
  return MBT_OK;
}

int mbt_env_get_clock(mbt_env* e, double* time, uint32_t* episode_step, uint32_t* philox_step) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  if (e->device_clock) {  // the clock lives on the device: wait for the stream and read it there
    mbt_device_clock now;
    const int rc = mbt_env_device_clock_read(e, &now);
    if (rc != MBT_OK) return rc;
    if (time != nullptr) *time = now.time;
    if (episode_step != nullptr) *episode_step = now.episode_step;
    if (philox_step != nullptr) *philox_step = now.philox_step;
    return MBT_OK;
  }
  if (time != nullptr) *time = e->time;
  if (episode_step != nullptr) *episode_step = e->episode_step;
  if (philox_step != nullptr) *philox_step = e->philox_step;
  return MBT_OK;
}

int mbt_env_record_events(mbt_env* e, int enabled) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  if (enabled && e->events == nullptr) {
    int rc = dev_alloc(&e->events, size_t(e->n_pad), e->stream);
    if (rc != MBT_OK) return rc;
  }
  e->record_events = enabled != 0;
  return MBT_OK;
}

int mbt_env_get_events_host(mbt_env* e, uint8_t* events_host) {
  if (e == nullptr || events_host == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (!e->record_events) return fail(MBT_ERR_STATE, "event recording is off");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  HIP_TRY(hipMemcpyAsync(events_host, e->events, size_t(e->n), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MBT_OK;
}

int mbt_env_clip_count(mbt_env* e, uint64_t* count) {
  if (e == nullptr || count == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "counter width");
  std::vector<unsigned long long> slots(mbt::kClipSlots);
  HIP_TRY(hipMemcpyAsync(slots.data(), e->clip_count, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  uint64_t total = 0;
  for (unsigned long long v : slots) total += v;
  *count = total;
  return MBT_OK;
}

int mbt_env_track_lane_returns(mbt_env* e, int enabled) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HOST_CLOCK(e);
  if (enabled && e->lane_returns == nullptr) {
    int rc = dev_alloc(&e->lane_returns, size_t(e->n_pad), e->stream);
    if (rc != MBT_OK) return rc;
  }
  e->track_returns = enabled != 0;
  return MBT_OK;
}

int mbt_env_return_sums(mbt_env* e, double sums[3]) {
  if (e == nullptr || sums == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  hipLaunchKernelGGL(mbt::reduce_returns_kernel, dim3(1), dim3(256), 0, e->stream, e->wave_sums, e->n_waves,
                     e->track_returns ? e->lane_returns : nullptr, e->n, e->reduce_out);
  HIP_TRY(hipGetLastError());
  double host[2] = {0.0, 0.0};
  HIP_TRY(hipMemcpyAsync(host, e->reduce_out, sizeof host, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  sums[0] = host[0];
  sums[1] = e->track_returns ? host[1] : NAN;
  sums[2] = static_cast<double>(e->n);
  return MBT_OK;
}

int mbt_env_return_sums_begin(mbt_env* e) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  if (e->sums_pending) return fail(MBT_ERR_STATE, "a return-sums request is already in flight: call mbt_env_return_sums_end first");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  hipLaunchKernelGGL(mbt::reduce_returns_kernel, dim3(1), dim3(256), 0, e->stream, e->wave_sums, e->n_waves,
                     e->track_returns ? e->lane_returns : nullptr, e->n, e->reduce_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(e->h_sums, e->reduce_out, 2 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipEventRecord(e->ev_sums, e->stream));
  e->sums_pending = true;
  return MBT_OK;
}

int mbt_env_return_sums_end(mbt_env* e, double sums[3]) {
  if (e == nullptr || sums == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (!e->sums_pending) return fail(MBT_ERR_STATE, "no return-sums request in flight");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  HIP_TRY(hipEventSynchronize(e->ev_sums));  // waits for the reduction only: later launches keep running
  e->sums_pending = false;
  sums[0] = e->h_sums[0];
  sums[1] = e->track_returns ? e->h_sums[1] : NAN;
  sums[2] = static_cast<double>(e->n);
  return MBT_OK;
}

int mbt_reward_calculate_host(int device, int reward_kind, double phi, double alpha, double inventory_exponent, const double* cur,
                              const double* nxt, int dim, uint64_t n, int is_terminal, const double* q_init,
                              const double* episode_length, const double* action, double risk_aversion, double* out) {
  if (cur == nullptr || nxt == nullptr || out == nullptr || n == 0 || dim < 4) return fail(MBT_ERR_INVALID, "bad argument");
  if (reward_kind < MBT_REW_PNL || reward_kind > MBT_REW_CJ_OE) return fail(MBT_ERR_INVALID, "reward kind %d has no device implementation", reward_kind);
  if ((reward_kind == MBT_REW_CJ_MM || reward_kind == MBT_REW_CJ_OE) && (q_init == nullptr || episode_length == nullptr))
    return fail(MBT_ERR_STATE, "calculate() before reset(): initial inventory / episode length unknown");
  if (reward_kind == MBT_REW_CJ_OE && action == nullptr) return fail(MBT_ERR_INVALID, "CjOeCriterion needs the action (trading speed)");
  int rc = check_device(device);
  if (rc != MBT_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  const size_t mat = n * dim * sizeof(double), vec = n * sizeof(double);
  double* d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), 2 * mat + 4 * vec));
  double *d_cur = d, *d_nxt = d + n * dim, *d_qi = d_nxt + n * dim, *d_len = d_qi + n, *d_out = d_len + n, *d_act = d_out + n;
  hipError_t he = hipMemcpy(d_cur, cur, mat, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(d_nxt, nxt, mat, hipMemcpyHostToDevice);
  if (he == hipSuccess && q_init != nullptr) he = hipMemcpy(d_qi, q_init, vec, hipMemcpyHostToDevice);
  if (he == hipSuccess && episode_length != nullptr) he = hipMemcpy(d_len, episode_length, vec, hipMemcpyHostToDevice);
  if (he == hipSuccess && action != nullptr) he = hipMemcpy(d_act, action, vec, hipMemcpyHostToDevice);
  if (he == hipSuccess) {
    hipLaunchKernelGGL(mbt::reward_calculate_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, reward_kind, d_cur, d_nxt, dim,
                       static_cast<uint32_t>(n), is_terminal, phi, alpha, inventory_exponent, d_qi, d_len, d_act, risk_aversion, d_out);
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = hipMemcpy(out, d_out, vec, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (he != hipSuccess) return fail(MBT_ERR_HIP, "reward_calculate failed: %s", hipGetErrorString(he));
  return MBT_OK;
}

int mbt_process_evaluate_host(int device, int op, const mbt_config* cfg, uint64_t n, const double* a, const double* b, const double* c, const double* d,
                              double* out) {
  if (cfg == nullptr || a == nullptr || out == nullptr || n == 0 || n > 0x7FFFFFFFull) return fail(MBT_ERR_INVALID, "bad argument");
  if (op < MBT_PROCESS_MIDPRICE_UPDATE || op > MBT_PROCESS_FILLS) return fail(MBT_ERR_INVALID, "unknown process operation %d", op);
  if ((op == MBT_PROCESS_MIDPRICE_UPDATE || op == MBT_PROCESS_HAWKES_UPDATE || op == MBT_PROCESS_FILLS) && b == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  if (op == MBT_PROCESS_ARRIVALS && cfg->arrival_kind == MBT_ARR_HAWKES && b == nullptr) return fail(MBT_ERR_INVALID, "Hawkes arrivals need the intensities");
  if (op == MBT_PROCESS_MIDPRICE_UPDATE && (cfg->midprice_kind < MBT_MID_BROWNIAN || cfg->midprice_kind > MBT_MID_LINEAR_SDE))
    return fail(MBT_ERR_INVALID, "midprice kind %d has no host-callable update (user expressions run inside an environment)", cfg->midprice_kind);
  if (op == MBT_PROCESS_FILLS && cfg->fill_kind != MBT_FILL_EXPONENTIAL && cfg->fill_kind != MBT_FILL_EXOGENOUS_MM) return fail(MBT_ERR_INVALID, "fill kind %d has no host-callable get_fills", cfg->fill_kind);
  int rc = check_device(device);
  if (rc != MBT_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  const double dt = cfg->terminal_time > 0.0 && cfg->n_steps > 0 ? cfg->terminal_time / cfg->n_steps : 0.0;
  const double mid_dt = cfg->midprice_step_size > 0.0 ? cfg->midprice_step_size : dt, arr_dt = cfg->arrival_step_size > 0.0 ? cfg->arrival_step_size : dt;
  mbt::PreciseParams X;
  fill_precise_params(*cfg, mid_dt, arr_dt, dt, X);
  const bool nonlinear = cfg->arrival_kind == MBT_ARR_POISSON_NONLINEAR;
  const double thr_bid = nonlinear ? 1.0 - std::exp(-cfg->intensity[0] * arr_dt) : cfg->intensity[0] * arr_dt;  // ARR:83 / ARR:56
  const double thr_ask = nonlinear ? 1.0 - std::exp(-cfg->intensity[1] * arr_dt) : cfg->intensity[1] * arr_dt;
  const size_t width = op == MBT_PROCESS_MIDPRICE_UPDATE ? 1 : 2, bytes = n * width * sizeof(double);
  const double* in[4] = {a, b, c, d};
  double* dev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipError_t he = hipSuccess;
  for (int k = 0; k < 5 && he == hipSuccess; ++k) {
    if (k < 4 && in[k] == nullptr) continue;
    he = hipMalloc(reinterpret_cast<void**>(&dev[k]), bytes);
    if (he == hipSuccess && k < 4) he = hipMemcpy(dev[k], in[k], bytes, hipMemcpyHostToDevice);
  }
  if (he == hipSuccess) {
    hipLaunchKernelGGL(mbt::process_evaluate_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, op, cfg->arrival_kind, cfg->fill_kind, X, thr_bid, thr_ask,
                       cfg->fill_exponent, cfg->exogenous_depth[0], cfg->exogenous_depth[1], cfg->base_fill_probability, dev[0], dev[1], dev[2], dev[3],
                       static_cast<uint32_t>(n), dev[4]);
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = hipMemcpy(out, dev[4], bytes, hipMemcpyDeviceToHost);
  for (double* p : dev)
    if (p != nullptr) (void)hipFree(p);
  if (he != hipSuccess) return fail(MBT_ERR_HIP, "process evaluation failed: %s", hipGetErrorString(he));
  return MBT_OK;
}

int mbt_rng_fill_host(int device, uint64_t seed, uint64_t trajectory_offset, uint32_t step, uint64_t n, float* u_arr,
                      float* u_fill, float* z) {
  if (trajectory_offset % mbt::kTileLanes != 0) return fail(MBT_ERR_INVALID, "trajectory_offset must be a multiple of 512");
  if (n == 0 || n > 0x7FFFF000ull) return fail(MBT_ERR_INVALID, "n out of range");
  int rc = check_device(device);
  if (rc != MBT_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  const uint32_t tiles = (static_cast<uint32_t>(n) + mbt::kTileLanes - 1u) / mbt::kTileLanes, n_pad = tiles * mbt::kTileLanes;
  float *d_ua = nullptr, *d_uf = nullptr, *d_z = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_ua), size_t(n_pad) * 2 * sizeof(float)));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_uf), size_t(n_pad) * 2 * sizeof(float)));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_z), size_t(n_pad) * sizeof(float)));
  hipLaunchKernelGGL(mbt::rng_fill_kernel, dim3(tiles), dim3(mbt::kBlockThreads), 0, nullptr, trajectory_offset >> 1, step,
                     static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), d_ua, d_uf, d_z);
  HIP_TRY(hipGetLastError());
  if (u_arr != nullptr) HIP_TRY(hipMemcpy(u_arr, d_ua, size_t(n) * 2 * sizeof(float), hipMemcpyDeviceToHost));
  if (u_fill != nullptr) HIP_TRY(hipMemcpy(u_fill, d_uf, size_t(n) * 2 * sizeof(float), hipMemcpyDeviceToHost));
  if (z != nullptr) HIP_TRY(hipMemcpy(z, d_z, size_t(n) * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipDeviceSynchronize());
  (void)hipFree(d_ua);
  (void)hipFree(d_uf);
  (void)hipFree(d_z);
  return MBT_OK;
}

int mbt_rng_fill_user_host(int device, uint64_t seed, uint64_t trajectory_offset, uint32_t step, uint64_t n, float* z_user) {
  if (trajectory_offset % mbt::kTileLanes != 0) return fail(MBT_ERR_INVALID, "trajectory_offset must be a multiple of 512");
  if (n == 0 || n > 0x7FFFF000ull || z_user == nullptr) return fail(MBT_ERR_INVALID, "bad argument");
  int rc = check_device(device);
  if (rc != MBT_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  const uint32_t tiles = (static_cast<uint32_t>(n) + mbt::kTileLanes - 1u) / mbt::kTileLanes, n_pad = tiles * mbt::kTileLanes;
  float* d_z = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_z), size_t(n_pad) * 2 * sizeof(float)));
  hipLaunchKernelGGL(mbt::rng_fill_user_kernel, dim3(tiles), dim3(mbt::kBlockThreads), 0, nullptr, trajectory_offset >> 1, step,
                     static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), d_z);
  hipError_t he = hipGetLastError();
  if (he == hipSuccess) he = hipMemcpy(z_user, d_z, size_t(n) * 2 * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(d_z);
  if (he != hipSuccess) return fail(MBT_ERR_HIP, "rng_fill_user failed: %s", hipGetErrorString(he));
  return MBT_OK;
}

int mbt_rng_fill_quad_host(int device, uint64_t seed, uint64_t trajectory_offset, uint32_t step, uint64_t n, float* z) {
  if (trajectory_offset % mbt::kSpeedTileLanes != 0) return fail(MBT_ERR_INVALID, "trajectory_offset must be a multiple of 1024");
  if (n == 0 || n > 0x7FFFF000ull || z == nullptr) return fail(MBT_ERR_INVALID, "bad argument");
  int rc = check_device(device);
  if (rc != MBT_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  const uint32_t n_pad = ((static_cast<uint32_t>(n) + mbt::kSpeedTileLanes - 1u) / mbt::kSpeedTileLanes) * mbt::kSpeedTileLanes, n_quads = n_pad / 4;
  float* d_z = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_z), size_t(n_pad) * sizeof(float)));
  hipLaunchKernelGGL(mbt::rng_fill_quad_kernel, dim3((n_quads + 255) / 256), dim3(256), 0, nullptr, trajectory_offset >> 2, step,
                     static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), n_quads, d_z);
  hipError_t he = hipGetLastError();
  if (he == hipSuccess) he = hipMemcpy(z, d_z, size_t(n) * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(d_z);
  if (he != hipSuccess) return fail(MBT_ERR_HIP, "rng_fill_quad failed: %s", hipGetErrorString(he));
  return MBT_OK;
}

int mbt_philox4x32_10_host(int device, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  if (ctr == nullptr || key == nullptr || out == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  int rc = check_device(device);
  if (rc != MBT_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  uint32_t* d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), 10 * sizeof(uint32_t)));
  HIP_TRY(hipMemcpy(d, ctr, 4 * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d + 4, key, 2 * sizeof(uint32_t), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mbt::philox_kat_kernel, dim3(1), dim3(1), 0, nullptr, d, d + 4, d + 6);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, d + 6, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return MBT_OK;
}

int mbt_env_timer_begin(mbt_env* e) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  HIP_TRY(hipEventRecord(e->ev_begin, e->stream));
  return MBT_OK;
}

int mbt_env_timer_stop(mbt_env* e) {
  if (e == nullptr) return fail(MBT_ERR_INVALID, "null env");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  HIP_TRY(hipEventRecord(e->ev_end, e->stream));
  return MBT_OK;
}

int mbt_env_timer_elapsed(mbt_env* e, float* elapsed_ms) {
  if (e == nullptr || elapsed_ms == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  RESIDENT_STOP(e);
  const auto t0 = std::chrono::steady_clock::now();  // poll before blocking, like mbt_env_synchronize
  while (hipEventQuery(e->ev_end) == hipErrorNotReady && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(200)) {
  }
  HIP_TRY(hipEventSynchronize(e->ev_end));
  HIP_TRY(hipEventElapsedTime(elapsed_ms, e->ev_begin, e->ev_end));
  return MBT_OK;
}

int mbt_env_timer_end(mbt_env* e, float* elapsed_ms) {
  if (e == nullptr || elapsed_ms == nullptr) return fail(MBT_ERR_INVALID, "null argument");
  const int rc = mbt_env_timer_stop(e);
  return rc != MBT_OK ? rc : mbt_env_timer_elapsed(e, elapsed_ms);
}

}  // extern "C"

// ---- the kernel source, embedded for mbt_env_create_jit (generated by mbt_gym_amd/build.py from csrc/philox.hpp and
//      csrc/step_kernel.hpp: the very files this library was compiled from) -------------------------------------------
#include "embedded_sources.inc"
