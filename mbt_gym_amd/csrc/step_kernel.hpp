// The fused TradingEnvironment.step() kernel for gfx950 (CDNA4, wave64).
//
// One launch = one env.step(action) over all lanes (reference: gym/TradingEnvironment.py:103-110 and the
// ~35 NumPy kernels it fans out to).  One GPU thread owns a PAIR of adjacent trajectories:
//   reads   2 state rows = D/2 x float4: [cash, inventory, time, midprice (, bid intensity, ask intensity)]
//           1 x float4 (A=2) or 2 x float4 (A=4) action rows
//   draws   3 Philox4x32-10 blocks = all the noise of the pair (philox.hpp), or loads injected noise
//   writes  2 next-state rows (D/2 x float4), 1 x float2 rewards
// The state is row-major (N, D) float32 - exactly the un-normalised observation the API returns - and is
// ping-ponged between two buffers, so the observation of step k stays valid while step k+1 is computed.
// Nothing else touches HBM: `dones` is a host scalar (TE:218-220), time is a kernel argument, the generator is
// stateless.  Algorithmic traffic per env-step: 4*(D + A + D + 1) bytes = 44 B for D=4, A=2.
//
// Schedule inside a wave: all loads are issued first; the Philox rounds (which depend on nothing in memory) run
// while they are in flight; an empty asm ties the loaded registers to the finished noise so the compiler cannot
// pull a consumer of the loads (and its s_waitcnt) above the generator.
//
// Numerics contract (checked by tests/ against oracle/ and the golden fixtures):
//   * arrivals, fills, market-order flags, inventory: BIT-EXACT against the float64 reference on the same
//     float32-representable draws.  Decisions are taken on exact thresholds: Poisson thresholds arrive rounded
//     UP to float32 (u < t_f64 <=> u < roundup32(t_f64) for float32 u); Hawkes thresholds and normalised
//     market-order flags are evaluated in double; the fill test uses v_exp_f32 and re-evaluates in double only
//     when the draw lies within the error band of the float32 exponential (about 4e-6 of draws).
//   * rewards: float32 but computed from the step's INCREMENTS, never as a difference of two large
//     mark-to-market values: PnL = n_b*d_b + n_a*d_a - h*(mb+ms) + q'*dS (+ clip corrections), which is the
//     reference's (c'+q'S') - (c+qS) (RW:27-33) with the S terms cancelled analytically.
//   * cash / midprice: float32 state, a few ulps from the float64 reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"

namespace mbt {

#ifndef MBT_BLOCK_THREADS
#define MBT_BLOCK_THREADS 256
#endif
constexpr int kBlockThreads = MBT_BLOCK_THREADS;  // wave64 x 4 per workgroup by default

enum : int { kMidBrownian = 0, kMidOu = 1 };
enum : int { kArrPoisson = 0, kArrHawkes = 1 };
enum : int { kDynLimit = 0, kDynLimitAndMarket = 1 };
enum : int { kRewPnl = 0, kRewRunning = 1, kRewCjMm = 2 };

// Compile-time shape of one kernel instantiation.
template <int MID_, int ARR_, int DYN_, int REW_, bool NORM_, bool INJECT_>
struct Variant {
  static constexpr int MID = MID_, ARR = ARR_, DYN = DYN_, REW = REW_;
  static constexpr bool NORM = NORM_;      // normalised actions and/or observations (TE:112-126)
  static constexpr bool INJECT = INJECT_;  // noise loaded from HBM instead of Philox
  static constexpr int DIM = (ARR_ == kArrHawkes) ? 6 : 4;
  static constexpr int VEC_PER_PAIR = DIM / 2;  // float4 per pair of state rows
};

// Wave-uniform parameters of one step: passed by value (kernarg -> SGPRs).
struct StepParams {
  uint32_t n;            // lanes of this shard
  uint32_t n_pairs;      // ceil(n / 2)
  uint64_t pair_offset;  // global pair index of local pair 0
  uint32_t key0, key1;   // Philox key (seed)
  uint32_t philox_step;  // Philox counter word 2
  int32_t is_terminal;   // this step ends the episode (TE:218-220), decided on the host
  float t_next;          // time written into the next state (TE:216)
  float dt;
  // midprice
  float drift_dt;        // mu * dt                      (MID:63)
  float vol_sqrt_dt;     // sigma * sqrt(dt)             (MID:64, MID:143)
  float ou_speed, ou_level;
  // arrivals
  float arr_thr_bid, arr_thr_ask;  // Poisson: smallest float32 >= lambda*dt computed in double (ARR:56)
  double dt_f64;                   // Hawkes: threshold lambda_lane * dt in double (ARR:123)
  float hawkes_base_bid, hawkes_base_ask, hawkes_speed, hawkes_jump;
  // fills
  float kappa;
  double kappa_f64;
  // dynamics
  float half_spread;
  float q_max, c_max;
  // reward
  int32_t exponent_is_two;
  float phi, alpha, exponent;
  float dt_over_episode;  // CjMm: dt / (T - t_start)    (RW:106)
  float q_init_scalar;    // CjMm: initial inventory when it is the same for every lane
  float reward_scale;     // TE:128-129
  // normalisation (TE:112-126); gradients are float32 like the reference's Box bounds
  int32_t norm_act, norm_obs;
  float act_lo[4], act_grad[4];
  float obs_lo[6], obs_grad[6];
};

struct StepBuffers {
  const float* state_in;   // (n_pad, D) row-major
  float* state_out;
  const float* action;     // (n_pad, A)
  float* reward;           // (n_pad)
  float* obs;              // normalised observation (n_pad, D) or nullptr
  const float* u_arr;      // injected noise (n_pad, 2), (n_pad, 2), (n_pad)
  const float* u_fill;
  const float* z;
  const float* q_init;     // CjMm per-lane initial inventory or nullptr
  uint8_t* events;         // nullptr unless recording
  float* lane_returns;     // nullptr unless tracking
  double* wave_sums;       // one slot per wave: running sum of rewards since reset
  unsigned long long* clip_count;
};

// ---- fill decisions --------------------------------------------------------------------------------------
// float32 exponential fill test (FILL:34, FILL:57-58) that is exact against float64: v_exp_f32 decides unless the
// draw lies inside its error band (plus the rounding of a normalised depth); `near` flags that case and ONE cold
// block per pair re-decides the flagged entries in double.
struct FillTest {
  bool fill;  // u < exp(-kappa * depth), fast evaluation
  bool near;  // the fast evaluation cannot be trusted
};

__device__ __forceinline__ FillTest fill_test(float u, float depth, const StepParams& P) {
  const float x = P.kappa * depth;
  const float p = __builtin_amdgcn_exp2f(-1.4426950408889634f * x);
  const float band = p * (4e-6f + 4e-7f * __builtin_fabsf(x)) + 1e-30f;
  const float d = u - p;
  return FillTest{d < 0.0f, __builtin_fabsf(d) <= band};
}

__device__ __forceinline__ float depth_of(float a, int side, bool norm, const StepParams& P) {
  return norm ? static_cast<float>((static_cast<double>(a) + 1.0) * P.act_grad[side] + P.act_lo[side]) : a;  // TE:124
}

// cold: exact re-decision of the flagged entries of a pair (about 4e-6 of draws get here)
__device__ __attribute__((cold)) void refine_fills_f64(const float (&u)[4], const float (&a)[4], bool norm, const StepParams& P,
                                                       bool (&fill)[4], const bool (&near)[4]) {
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    if (!near[k]) continue;
    double depth = a[k];
    if (norm) depth = (static_cast<double>(a[k]) + 1.0) * P.act_grad[k & 1] + P.act_lo[k & 1];
    fill[k] = static_cast<double>(u[k]) < exp(-P.kappa_f64 * depth);
  }
}

// numpy `q ** p` (RW:101-104, RW:133-137); p == 2 in every reference configuration
__device__ __forceinline__ float pow_inventory(float q, const StepParams& P) {
  return __builtin_expect(P.exponent_is_two, 1) ? q * q : powf(q, P.exponent);
}

struct LaneResult {
  float4 core;
  float2 lam;
  float reward;
  uint32_t events;
};

template <class V>
__device__ __forceinline__ LaneResult step_lane(const float4 core, const float2 lam, const float4 act, const float d_bid,
                                                const float d_ask, const bool raw_fill_bid, const bool raw_fill_ask,
                                                const LaneNoise nz, const float q_init, const float t_next,
                                                const bool is_terminal, const StepParams& P) {
  const float cash = core.x, q = core.y, mid = core.w;

  // -- arrivals (ARR:54-56 / ARR:121-123), strict '<'
  bool arr_bid, arr_ask;
  if (V::ARR == kArrPoisson) {
    arr_bid = nz.ua_bid < P.arr_thr_bid;
    arr_ask = nz.ua_ask < P.arr_thr_ask;
  } else {
    arr_bid = static_cast<double>(nz.ua_bid) < static_cast<double>(lam.x) * P.dt_f64;
    arr_ask = static_cast<double>(nz.ua_ask) < static_cast<double>(lam.y) * P.dt_f64;
  }

  // -- fills (FILL:28-34, FILL:57-58) masked by the PRE-update inventory (TE:323-327)
  const bool fill_bid = raw_fill_bid && !(q >= P.q_max);
  const bool fill_ask = raw_fill_ask && !(q <= -P.q_max);
  const float n_bid = (arr_bid && fill_bid) ? 1.0f : 0.0f;
  const float n_ask = (arr_ask && fill_ask) ? 1.0f : 0.0f;

  // -- cash / inventory with the OLD midprice (MD:82-84); market orders first (MD:208-214), then limit
  //    fills (MD:108-116 / MD:215-222)
  float q_new = q, cash_new = cash, gain = 0.0f;
  uint32_t ev = (arr_bid ? 1u : 0u) | (arr_ask ? 2u : 0u) | (fill_bid ? 4u : 0u) | (fill_ask ? 8u : 0u);
  if (V::DYN == kDynLimitAndMarket) {
    bool mo_buy, mo_sell;
    if (V::NORM && P.norm_act) {
      mo_buy = (static_cast<double>(act.z) + 1.0) * P.act_grad[2] + P.act_lo[2] > 0.5;
      mo_sell = (static_cast<double>(act.w) + 1.0) * P.act_grad[3] + P.act_lo[3] > 0.5;
    } else {
      mo_buy = act.z > 0.5f;
      mo_sell = act.w > 0.5f;
    }
    const float mb = mo_buy ? 1.0f : 0.0f, ms = mo_sell ? 1.0f : 0.0f;
    cash_new += ms * (mid - P.half_spread) - mb * (mid + P.half_spread);
    q_new += mb - ms;
    gain -= P.half_spread * (mb + ms);
    ev |= (mo_buy ? 16u : 0u) | (mo_sell ? 32u : 0u);
  }
  q_new += n_bid - n_ask;
  cash_new += n_ask * (mid + d_ask) - n_bid * (mid - d_bid);
  gain += n_bid * d_bid + n_ask * d_ask;

  // -- clip (TE:283-289)
  const float q_clip = __builtin_fminf(__builtin_fmaxf(q_new, -P.q_max), P.q_max);
  const float c_clip = __builtin_fminf(__builtin_fmaxf(cash_new, -P.c_max), P.c_max);
  const float dq_clip = q_clip - q_new;  // 0 unless the inventory clip fired
  const float dc_clip = c_clip - cash_new;
  ev |= (dq_clip != 0.0f ? 64u : 0u) | (dc_clip != 0.0f ? 128u : 0u);

  // -- midprice (MID:60-65 / MID:140-143: the OU pull is not scaled by dt in the reference)
  float d_mid;
  if (V::MID == kMidBrownian) {
    d_mid = P.drift_dt + P.vol_sqrt_dt * nz.z;
  } else {
    d_mid = -P.ou_speed * (mid - P.ou_level) + P.vol_sqrt_dt * nz.z;
  }
  const float mid_new = mid + d_mid;

  // -- Hawkes intensities jump on arrivals, not on fills (ARR:110-119)
  float2 lam_new = lam;
  if (V::ARR == kArrHawkes) {
    lam_new.x = (lam.x + P.hawkes_speed * (P.hawkes_base_bid - lam.x) * P.dt) + (arr_bid ? P.hawkes_jump : 0.0f);
    lam_new.y = (lam.y + P.hawkes_speed * (P.hawkes_base_ask - lam.y) * P.dt) + (arr_ask ? P.hawkes_jump : 0.0f);
  }

  // -- reward (RW:23-33, RW:96-109, RW:128-138): incremental mark-to-market, see the header comment
  float reward = gain + q_clip * d_mid + dq_clip * mid + dc_clip;
  if (V::REW != kRewPnl) {
    const float qp = pow_inventory(q_clip, P);
    reward -= P.dt * P.phi * qp;
    if (V::REW == kRewRunning) {
      reward -= is_terminal ? P.alpha * qp : 0.0f;
    } else {
      reward -= P.alpha * ((qp - pow_inventory(q, P)) + P.dt_over_episode * pow_inventory(q_init, P));
    }
  }
  reward *= P.reward_scale;

  LaneResult r;
  r.core = make_float4(c_clip, q_clip, t_next, mid_new);
  r.lam = lam_new;
  r.reward = reward;
  r.events = ev;
  return r;
}

// Sum over the 64 lanes of a wave with DPP row operations + 4 readlanes (no LDS traffic).
__device__ __forceinline__ float wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  const int iv = __builtin_bit_cast(int, v);  // every lane now holds the sum of its row of 16
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
}

__device__ __forceinline__ void write_obs_row(float* obs, uint32_t lane, int dim, const float4 core, const float2 lam,
                                              const StepParams& P) {
  float* row = obs + static_cast<size_t>(lane) * dim;
  const float v[6] = {core.x, core.y, core.z, core.w, lam.x, lam.y};
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    if (j < dim) row[j] = P.norm_obs ? (v[j] - P.obs_lo[j]) / P.obs_grad[j] - 1.0f : v[j];
  }
}

// (x - lo) / grad - 1 per column (TE:112-118), float32 division like the reference's float arithmetic on Box bounds
__device__ __forceinline__ void normalise_row(float4& core, float2& lam, int dim, const StepParams& P) {
  if (!P.norm_obs) return;
  core.x = (core.x - P.obs_lo[0]) / P.obs_grad[0] - 1.0f;
  core.y = (core.y - P.obs_lo[1]) / P.obs_grad[1] - 1.0f;
  core.z = (core.z - P.obs_lo[2]) / P.obs_grad[2] - 1.0f;
  core.w = (core.w - P.obs_lo[3]) / P.obs_grad[3] - 1.0f;
  if (dim == 6) {
    lam.x = (lam.x - P.obs_lo[4]) / P.obs_grad[4] - 1.0f;
    lam.y = (lam.y - P.obs_lo[5]) / P.obs_grad[5] - 1.0f;
  }
}

// Everything one pair of trajectories reads from HBM.
template <class V>
struct PairLoads {
  float4 s0, s1, s2;  // D/2 float4 of state (s2: Hawkes only)
  float4 a0, a1;      // actions (a1: limit+market only)
  float4 ua, uf;      // injected noise
  float2 zz;
  float2 qi;          // CjMm per-lane initial inventories
};

template <class V>
__device__ __forceinline__ PairLoads<V> load_pair(const StepBuffers& B, const StepParams& P, uint32_t pair) {
  PairLoads<V> L;
  const float4* src = reinterpret_cast<const float4*>(B.state_in) + static_cast<size_t>(pair) * V::VEC_PER_PAIR;
  L.s0 = src[0];
  L.s1 = src[1];
  if (V::ARR == kArrHawkes) L.s2 = src[2];
  if (V::DYN == kDynLimit) {
    L.a0 = reinterpret_cast<const float4*>(B.action)[pair];
  } else {
    L.a0 = reinterpret_cast<const float4*>(B.action)[2 * pair];
    L.a1 = reinterpret_cast<const float4*>(B.action)[2 * pair + 1];
  }
  if (V::INJECT) {
    L.ua = reinterpret_cast<const float4*>(B.u_arr)[pair];
    L.uf = reinterpret_cast<const float4*>(B.u_fill)[pair];
    L.zz = reinterpret_cast<const float2*>(B.z)[pair];
  }
  L.qi = make_float2(P.q_init_scalar, P.q_init_scalar);
  if (V::REW == kRewCjMm && B.q_init != nullptr) L.qi = reinterpret_cast<const float2*>(B.q_init)[pair];
  return L;
}

// Orders the schedule: every operand is an in/out of one empty asm, so the noise is complete before, and every
// consumer of the loaded state/action after, this point.  Costs no instruction.
template <class V>
__device__ __forceinline__ void tie_loads_to_noise(PairLoads<V>& L, LaneNoise& a, LaneNoise& b) {
  asm volatile("; loads are first consumed below this line"
               : "+v"(L.s0.x), "+v"(L.s0.y), "+v"(L.s0.z), "+v"(L.s0.w), "+v"(L.s1.x), "+v"(L.s1.y), "+v"(L.s1.z), "+v"(L.s1.w),
                 "+v"(L.a0.x), "+v"(L.a0.y), "+v"(L.a0.z), "+v"(L.a0.w), "+v"(a.ua_bid), "+v"(a.ua_ask), "+v"(a.uf_bid), "+v"(a.uf_ask), "+v"(a.z),
                 "+v"(b.ua_bid), "+v"(b.ua_ask), "+v"(b.uf_bid), "+v"(b.uf_ask), "+v"(b.z));
}

// One env-step of a pair of lanes held in registers: fill tests of the four quotes (TE:104 de-normalisation
// first), then the per-lane dynamics.
template <class V>
__device__ __forceinline__ void advance_pair(const float4 core0, const float2 lam0, const float4 core1, const float2 lam1,
                                             const float4 act0, const float4 act1, const LaneNoise& nz0, const LaneNoise& nz1,
                                             const float2 q_init, const float t_next, const bool is_terminal,
                                             const StepParams& P, LaneResult& r0, LaneResult& r1) {
  const bool norm_act = V::NORM && P.norm_act;
  const float a[4] = {act0.x, act0.y, act1.x, act1.y};
  const float u[4] = {nz0.uf_bid, nz0.uf_ask, nz1.uf_bid, nz1.uf_ask};
  float depth[4];
  bool fill[4], near[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    depth[k] = depth_of(a[k], k & 1, norm_act, P);
    const FillTest t = fill_test(u[k], depth[k], P);
    fill[k] = t.fill;
    near[k] = t.near;
  }
  if (__builtin_expect(near[0] | near[1] | near[2] | near[3], 0)) refine_fills_f64(u, a, norm_act, P, fill, near);
  r0 = step_lane<V>(core0, lam0, act0, depth[0], depth[1], fill[0], fill[1], nz0, q_init.x, t_next, is_terminal, P);
  r1 = step_lane<V>(core1, lam1, act1, depth[2], depth[3], fill[2], fill[3], nz1, q_init.y, t_next, is_terminal, P);
}

// rows of 6 are packed [c q t S | lb la c q | t S lb la] in three float4
template <class V>
__device__ __forceinline__ void unpack_rows(const float4 s0, const float4 s1, const float4 s2, float4& core0, float2& lam0,
                                            float4& core1, float2& lam1) {
  core0 = s0;
  core1 = s1;
  lam0 = lam1 = make_float2(0.f, 0.f);
  if (V::ARR == kArrHawkes) {
    lam0 = make_float2(s1.x, s1.y);
    core1 = make_float4(s1.z, s1.w, s2.x, s2.y);
    lam1 = make_float2(s2.z, s2.w);
  }
}

template <class V>
__device__ __forceinline__ void store_rows(float* base, uint32_t pair, const float4 core0, const float2 lam0, const float4 core1,
                                           const float2 lam1) {
  float4* dst = reinterpret_cast<float4*>(base) + static_cast<size_t>(pair) * V::VEC_PER_PAIR;
  if (V::ARR == kArrHawkes) {
    dst[0] = core0;
    dst[1] = make_float4(lam0.x, lam0.y, core1.x, core1.y);
    dst[2] = make_float4(core1.z, core1.w, lam1.x, lam1.y);
  } else {
    dst[0] = core0;
    dst[1] = core1;
  }
}

// normalised observation rows of a pair, as full-width vector stores
template <class V>
__device__ __forceinline__ void store_obs_rows(float* base, uint32_t pair, float4 core0, float2 lam0, float4 core1, float2 lam1,
                                               const StepParams& P) {
  normalise_row(core0, lam0, V::DIM, P);
  normalise_row(core1, lam1, V::DIM, P);
  store_rows<V>(base, pair, core0, lam0, core1, lam1);
}

// Arithmetic and stores of one pair; returns the pair's reward sum (pad lane excluded).
template <class V>
__device__ __forceinline__ float finish_pair(const StepBuffers& B, const StepParams& P, uint32_t pair, const PairLoads<V>& L,
                                             const LaneNoise& nz0, const LaneNoise& nz1) {
  const uint32_t lane0 = 2u * pair;
  float4 core0, core1;
  float2 lam0, lam1;
  unpack_rows<V>(L.s0, L.s1, L.s2, core0, lam0, core1, lam1);
  float4 act0, act1;
  if (V::DYN == kDynLimit) {
    act0 = make_float4(L.a0.x, L.a0.y, 0.f, 0.f);
    act1 = make_float4(L.a0.z, L.a0.w, 0.f, 0.f);
  } else {
    act0 = L.a0;
    act1 = L.a1;
  }

  LaneResult r0, r1;
  advance_pair<V>(core0, lam0, core1, lam1, act0, act1, nz0, nz1, L.qi, P.t_next, P.is_terminal != 0, P, r0, r1);

  store_rows<V>(B.state_out, pair, r0.core, r0.lam, r1.core, r1.lam);
  reinterpret_cast<float2*>(B.reward)[pair] = make_float2(r0.reward, r1.reward);

  const bool second = lane0 + 1 < P.n;  // the pad lane of an odd shard is computed but never reported

  // -- optional outputs (wave-uniform branches)
  if (V::NORM && B.obs != nullptr) store_obs_rows<V>(B.obs, pair, r0.core, r0.lam, r1.core, r1.lam, P);
  if (B.events != nullptr) {
    reinterpret_cast<uint16_t*>(B.events)[pair] = static_cast<uint16_t>(r0.events | (r1.events << 8));
  }
  if (B.lane_returns != nullptr) {
    float2 acc = reinterpret_cast<float2*>(B.lane_returns)[pair];
    acc.x += r0.reward;
    acc.y += r1.reward;
    reinterpret_cast<float2*>(B.lane_returns)[pair] = acc;
  }
  const uint32_t clipped = ((r0.events >> 6) != 0u ? 1u : 0u) + ((second && (r1.events >> 6) != 0u) ? 1u : 0u);
  if (__builtin_expect(clipped != 0u, 0)) atomicAdd(B.clip_count, static_cast<unsigned long long>(clipped));
  return r0.reward + (second ? r1.reward : 0.0f);
}

template <class V>
__global__ __launch_bounds__(kBlockThreads) void step_kernel(const StepBuffers B, const StepParams P) {
  const uint32_t pair = blockIdx.x * kBlockThreads + threadIdx.x;
  float r_sum = 0.0f;
  if (pair < P.n_pairs) {
    PairLoads<V> L = load_pair<V>(B, P, pair);  // issue every load ...
    LaneNoise nz0, nz1;
    if (V::INJECT) {
      nz0 = LaneNoise{L.ua.x, L.ua.y, L.uf.x, L.uf.y, L.zz.x};
      nz1 = LaneNoise{L.ua.z, L.ua.w, L.uf.z, L.uf.w, L.zz.y};
    } else {
      philox_pair_noise(P.pair_offset + pair, P.philox_step, P.key0, P.key1, nz0, nz1);  // ... draw while they fly
      tie_loads_to_noise<V>(L, nz0, nz1);
    }
    r_sum = finish_pair<V>(B, P, pair, L, nz0, nz1);
  }
  // -- per-wave running sum of rewards (numerator of the mean episode return): one slot per wave, one
  //    fire-and-forget hardware fp64 atomic per wave, no contention
  const float total = wave_sum(r_sum);
  if ((threadIdx.x & 63u) == 0u) {
    const uint32_t wave = blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6);
    unsafeAtomicAdd(&B.wave_sums[wave], static_cast<double>(total));
  }
}

// ---- fused rollout -----------------------------------------------------------------------
// Many consecutive env-steps in ONE launch with an on-device closed-form policy: the pair's state stays in
// registers, noise comes from the same Philox stream the step kernel would draw (philox step = first + k), so a
// rollout is bit-identical to the equivalent sequence of step() calls.  The caller's per-time-step Python loop
// (generate_trajectory.py:21-34) disappears; HBM is touched only to record the trajectory (optional, time-major
// so that every store is a coalesced float4) and once at the end for the final state.
enum : int { kPolicyFixed = 0, kPolicyAvellanedaStoikov = 1, kPolicyTable = 2 };

struct RolloutParams {
  uint32_t n_steps;        // env-steps to run in this launch
  int32_t last_is_terminal;  // the final one ends the episode (TE:218-220), decided on the host
  double t_start, dt_f64, terminal_time;  // the clock, advanced exactly like the host does (t += dt, TE:216)
  int32_t policy;
  float action[4];         // kPolicyFixed: the constant action (FixedActionAgent / FixedSpreadAgent, AG:25-42)
  float as_c1, as_c2;      // kPolicyAvellanedaStoikov: gamma sigma^2 and (2/gamma) ln(1 + gamma/kappa) (AG:70-83)
  const float2* table;     // kPolicyTable: (rows, cols) of (bid, ask) depths, device memory
  uint32_t table_row0, table_rows, table_cols;  // row of the first step of this launch
  int32_t table_q_offset;
  float* obs_traj;         // (n_steps + 1, n_pad, D) or nullptr; row 0 is the observation before the first step
  float* act_traj;         // (n_steps, n_pad, A) or nullptr
  float* rew_traj;         // (n_steps, n_pad) or nullptr
};

template <class V>
__global__ __launch_bounds__(kBlockThreads) void rollout_kernel(const StepBuffers B, const StepParams P, const RolloutParams R) {
  static_assert(!V::INJECT, "rollouts draw their own noise");
  constexpr int A = (V::DYN == kDynLimitAndMarket) ? 4 : 2;
  const uint32_t pair = blockIdx.x * kBlockThreads + threadIdx.x;
  float ret_sum = 0.0f;
  if (pair < P.n_pairs) {
    const uint32_t lane0 = 2u * pair;
    const size_t n_pad = static_cast<size_t>(P.n_pairs) * 2;
    const float4* src = reinterpret_cast<const float4*>(B.state_in) + static_cast<size_t>(pair) * V::VEC_PER_PAIR;
    float4 core0, core1;
    float2 lam0, lam1;
    unpack_rows<V>(src[0], src[1], V::ARR == kArrHawkes ? src[2] : make_float4(0.f, 0.f, 0.f, 0.f), core0, lam0, core1, lam1);
    float2 qi = make_float2(P.q_init_scalar, P.q_init_scalar);
    if (V::REW == kRewCjMm && B.q_init != nullptr) qi = reinterpret_cast<const float2*>(B.q_init)[pair];
    float ret0 = 0.0f, ret1 = 0.0f;
    uint32_t clipped = 0;
    double t = R.t_start;
    if (R.obs_traj != nullptr) {
      if (V::NORM) {
        store_obs_rows<V>(R.obs_traj, pair, core0, lam0, core1, lam1, P);
      } else {
        store_rows<V>(R.obs_traj, pair, core0, lam0, core1, lam1);
      }
    }
    for (uint32_t k = 0; k < R.n_steps; ++k) {
      LaneNoise nz0, nz1;
      philox_pair_noise(P.pair_offset + pair, P.philox_step + k, P.key0, P.key1, nz0, nz1);
      float4 act0, act1;
      if (R.policy == kPolicyFixed) {
        act0 = act1 = make_float4(R.action[0], R.action[1], R.action[2], R.action[3]);
      } else if (R.policy == kPolicyTable) {  // quotes tabulated over (time step, inventory), e.g. Cartea-Jaimungal
        const uint32_t row = min(R.table_row0 + k, R.table_rows - 1u);
        const int c0 = min(max(static_cast<int>(core0.y) + R.table_q_offset, 0), static_cast<int>(R.table_cols) - 1);
        const int c1 = min(max(static_cast<int>(core1.y) + R.table_q_offset, 0), static_cast<int>(R.table_cols) - 1);
        const float2 d0 = R.table[static_cast<size_t>(row) * R.table_cols + c0];
        const float2 d1 = R.table[static_cast<size_t>(row) * R.table_cols + c1];
        act0 = make_float4(d0.x, d0.y, 0.f, 0.f);
        act1 = make_float4(d1.x, d1.y, 0.f, 0.f);
      } else {  // Avellaneda-Stoikov quotes from (inventory, time) of the current observation
        const float tau = static_cast<float>(R.terminal_time - t);
        const float half = 0.5f * (R.as_c1 * tau + R.as_c2);
        const float s0 = core0.y * R.as_c1 * tau, s1 = core1.y * R.as_c1 * tau;
        act0 = make_float4(s0 + half, -s0 + half, 0.f, 0.f);
        act1 = make_float4(s1 + half, -s1 + half, 0.f, 0.f);
      }
      t += R.dt_f64;
      const bool terminal = (k + 1 == R.n_steps) && R.last_is_terminal != 0;
      LaneResult r0, r1;
      advance_pair<V>(core0, lam0, core1, lam1, act0, act1, nz0, nz1, qi, static_cast<float>(t), terminal, P, r0, r1);
      core0 = r0.core; lam0 = r0.lam; core1 = r1.core; lam1 = r1.lam;
      ret0 += r0.reward;
      ret1 += r1.reward;
      clipped += ((r0.events >> 6) != 0u ? 1u : 0u) + ((r1.events >> 6) != 0u && lane0 + 1 < P.n ? 1u : 0u);
      if (R.obs_traj != nullptr) {
        float* dst = R.obs_traj + static_cast<size_t>(k + 1) * n_pad * V::DIM;
        if (V::NORM) {
          store_obs_rows<V>(dst, pair, core0, lam0, core1, lam1, P);
        } else {
          store_rows<V>(dst, pair, core0, lam0, core1, lam1);
        }
      }
      if (R.act_traj != nullptr) {
        float* dst = R.act_traj + static_cast<size_t>(k) * n_pad * A;
        if (A == 2) {
          reinterpret_cast<float4*>(dst)[pair] = make_float4(act0.x, act0.y, act1.x, act1.y);
        } else {
          reinterpret_cast<float4*>(dst)[2 * pair] = act0;
          reinterpret_cast<float4*>(dst)[2 * pair + 1] = act1;
        }
      }
      if (R.rew_traj != nullptr) reinterpret_cast<float2*>(R.rew_traj + static_cast<size_t>(k) * n_pad)[pair] = make_float2(r0.reward, r1.reward);
      if (k + 1 == R.n_steps) {  // what step() leaves behind: last rewards (and events) of the final step
        reinterpret_cast<float2*>(B.reward)[pair] = make_float2(r0.reward, r1.reward);
        if (B.events != nullptr) reinterpret_cast<uint16_t*>(B.events)[pair] = static_cast<uint16_t>(r0.events | (r1.events << 8));
      }
    }
    store_rows<V>(B.state_out, pair, core0, lam0, core1, lam1);
    if (V::NORM && B.obs != nullptr) store_obs_rows<V>(B.obs, pair, core0, lam0, core1, lam1, P);
    if (B.lane_returns != nullptr) {
      float2 acc = reinterpret_cast<float2*>(B.lane_returns)[pair];
      acc.x += ret0;
      acc.y += ret1;
      reinterpret_cast<float2*>(B.lane_returns)[pair] = acc;
    }
    if (__builtin_expect(clipped != 0u, 0)) atomicAdd(B.clip_count, static_cast<unsigned long long>(clipped));
    ret_sum = ret0 + (lane0 + 1 < P.n ? ret1 : 0.0f);
  }
  const float total = wave_sum(ret_sum);
  if ((threadIdx.x & 63u) == 0u) {
    const uint32_t wave = blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6);
    unsafeAtomicAdd(&B.wave_sums[wave], static_cast<double>(total));
  }
}

// ---- small helper kernels ----------------------------------------------------------------------------------

// reset (TE:131-140): rows [initial_cash, q0, start_time, initial_price (, Hawkes baselines)], zeroed accumulators.
__global__ void reset_kernel(float* state, float* obs, float* lane_returns, double* wave_sums, const float* q0,
                             float q0_scalar, float cash0, float t0, float s0, float lam_bid, float lam_ask,
                             uint32_t n_pad, uint32_t n_waves, int dim, const StepParams P) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_waves) wave_sums[i] = 0.0;
  if (i >= n_pad) return;
  const float4 c = make_float4(cash0, q0 != nullptr ? q0[i] : q0_scalar, t0, s0);
  const float2 l = make_float2(lam_bid, lam_ask);
  float* row = state + static_cast<size_t>(i) * dim;
  row[0] = c.x; row[1] = c.y; row[2] = c.z; row[3] = c.w;
  if (dim == 6) { row[4] = l.x; row[5] = l.y; }
  if (obs != nullptr) write_obs_row(obs, i, dim, c, l, P);
  if (lane_returns != nullptr) lane_returns[i] = 0.0f;
}

// un-normalised state rows -> normalised observation rows (after set_state)
__global__ void normalise_rows_kernel(const float* state, float* obs, uint32_t n_pad, int dim, const StepParams P) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  const float* r = state + static_cast<size_t>(i) * dim;
  const float4 c = make_float4(r[0], r[1], r[2], r[3]);
  const float2 l = dim == 6 ? make_float2(r[4], r[5]) : make_float2(0.f, 0.f);
  write_obs_row(obs, i, dim, c, l, P);
}

// [sum of wave_sums, sum of lane_returns^2] -> out[0], out[1]; one block.
__global__ void reduce_returns_kernel(const double* wave_sums, uint32_t n_waves, const float* lane_returns, uint32_t n,
                                      double* out) {
  __shared__ double s_sum[256], s_sq[256];
  double a = 0.0, b = 0.0;
  for (uint32_t i = threadIdx.x; i < n_waves; i += blockDim.x) a += wave_sums[i];
  if (lane_returns != nullptr)
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) b += static_cast<double>(lane_returns[i]) * lane_returns[i];
  s_sum[threadIdx.x] = a;
  s_sq[threadIdx.x] = b;
  __syncthreads();
  for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      s_sum[threadIdx.x] += s_sum[threadIdx.x + s];
      s_sq[threadIdx.x] += s_sq[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = s_sum[0];
    out[1] = s_sq[0];
  }
}

// RewardFunction.calculate on caller-supplied state matrices (RW:23-33, RW:96-109, RW:128-138), in DOUBLE and in
// the reference's order of operations, so host code that calls `calculate()` on stored trajectories (and the
// reference's own unit tests) gets the reference's float64 values without a CPU implementation.
__global__ void reward_calculate_kernel(int kind, const double* cur, const double* nxt, int dim, uint32_t n, int is_terminal,
                                        double phi, double alpha, double p, const double* q_init, const double* episode_length,
                                        double* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* c = cur + static_cast<size_t>(i) * dim;
  const double* x = nxt + static_cast<size_t>(i) * dim;
  const double pnl = (x[0] + x[1] * x[3]) - (c[0] + c[1] * c[3]);
  double r = pnl;
  if (kind != kRewPnl) {
    const double dt = x[2] - c[2];
    const double qp = (p == 2.0) ? x[1] * x[1] : pow(x[1], p);
    r = pnl - dt * phi * qp;
    if (kind == kRewRunning) {
      r = r - alpha * static_cast<double>(is_terminal) * qp;
    } else {
      const double q0p = (p == 2.0) ? c[1] * c[1] : pow(c[1], p);
      const double qip = (p == 2.0) ? q_init[i] * q_init[i] : pow(q_init[i], p);
      r = r - alpha * (qp - q0p + dt / episode_length[i] * qip);
    }
  }
  out[i] = r;
}

// The production noise, written out (tests pin the generator and tie Philox mode to injected mode with it).
__global__ void rng_fill_kernel(uint64_t pair_offset, uint32_t step, uint32_t k0, uint32_t k1, uint32_t n_pairs,
                                float* u_arr, float* u_fill, float* z) {
  const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= n_pairs) return;
  LaneNoise a, b;
  philox_pair_noise(pair_offset + pair, step, k0, k1, a, b);
  if (u_arr != nullptr) reinterpret_cast<float4*>(u_arr)[pair] = make_float4(a.ua_bid, a.ua_ask, b.ua_bid, b.ua_ask);
  if (u_fill != nullptr) reinterpret_cast<float4*>(u_fill)[pair] = make_float4(a.uf_bid, a.uf_ask, b.uf_bid, b.uf_ask);
  if (z != nullptr) reinterpret_cast<float2*>(z)[pair] = make_float2(a.z, b.z);
}

__global__ void philox_kat_kernel(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
  const PhiloxWords w = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
  out[0] = w.w0; out[1] = w.w1; out[2] = w.w2; out[3] = w.w3;
}

}  // namespace mbt
