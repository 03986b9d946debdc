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

constexpr int kBlockThreads = 256;  // 4 wave64 per workgroup

enum : int { kMidBrownian = 0, kMidOu = 1 };
enum : int { kArrPoisson = 0, kArrHawkes = 1 };
enum : int { kDynLimit = 0, kDynLimitAndMarket = 1 };
enum : int { kRewPnl = 0, kRewRunning = 1, kRewCjMm = 2 };

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
  int32_t reward_kind;
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

__device__ __forceinline__ float pow_inventory(float q, const StepParams& P) {
  // numpy `q ** p` (RW:101-104, RW:133-137): p == 2 is the case every reference config uses
  return P.exponent_is_two ? q * q : powf(q, P.exponent);
}

// float32 exponential fill test, exact against float64 (FILL:34, FILL:57-58).
__device__ __forceinline__ bool fill_decision(float u, float depth_f32, double depth_f64, const StepParams& P) {
  const float x = P.kappa * depth_f32;
  const float p = __builtin_amdgcn_exp2f(-1.4426950408889634f * x);
  const float band = p * (4e-6f + 4e-7f * __builtin_fabsf(x)) + 1e-30f;
  const float d = u - p;
  if (__builtin_expect(__builtin_fabsf(d) <= band, 0)) {
    return static_cast<double>(u) < exp(-P.kappa_f64 * depth_f64);
  }
  return d < 0.0f;
}

struct LaneResult {
  float4 core;
  float2 lam;
  float reward;
  uint32_t events;
};

template <int MID, int ARR, int DYN>
__device__ __forceinline__ LaneResult step_lane(const float4 core, const float2 lam, const float a0, const float a1,
                                                const float a2, const float a3, const LaneNoise nz,
                                                const float q_init, const StepParams& P) {
  const float cash = core.x, q = core.y, mid = core.w;

  // -- action (TE:104, TE:120-126): depths in float32 for the arithmetic, in double for exact decisions
  float d_bid = a0, d_ask = a1;
  double d_bid64 = a0, d_ask64 = a1;
  if (P.norm_act) {
    d_bid64 = (static_cast<double>(a0) + 1.0) * P.act_grad[0] + P.act_lo[0];
    d_ask64 = (static_cast<double>(a1) + 1.0) * P.act_grad[1] + P.act_lo[1];
    d_bid = static_cast<float>(d_bid64);
    d_ask = static_cast<float>(d_ask64);
  }

  // -- arrivals (ARR:54-56 / ARR:121-123), strict '<'
  bool arr_bid, arr_ask;
  if (ARR == kArrPoisson) {
    arr_bid = nz.ua_bid < P.arr_thr_bid;
    arr_ask = nz.ua_ask < P.arr_thr_ask;
  } else {
    arr_bid = static_cast<double>(nz.ua_bid) < static_cast<double>(lam.x) * P.dt_f64;
    arr_ask = static_cast<double>(nz.ua_ask) < static_cast<double>(lam.y) * P.dt_f64;
  }

  // -- fills (FILL:28-34, FILL:57-58) masked by the PRE-update inventory (TE:323-327)
  const bool fill_bid = fill_decision(nz.uf_bid, d_bid, d_bid64, P) && !(q >= P.q_max);
  const bool fill_ask = fill_decision(nz.uf_ask, d_ask, d_ask64, P) && !(q <= -P.q_max);
  const float n_bid = (arr_bid && fill_bid) ? 1.0f : 0.0f;
  const float n_ask = (arr_ask && fill_ask) ? 1.0f : 0.0f;

  // -- cash / inventory with the OLD midprice (MD:82-84); market orders first (MD:208-214), then limit
  //    fills (MD:108-116 / MD:215-222)
  float q_new = q, cash_new = cash, gain = 0.0f;
  uint32_t ev = (arr_bid ? 1u : 0u) | (arr_ask ? 2u : 0u) | (fill_bid ? 4u : 0u) | (fill_ask ? 8u : 0u);
  if (DYN == kDynLimitAndMarket) {
    bool mo_buy, mo_sell;
    if (P.norm_act) {
      mo_buy = (static_cast<double>(a2) + 1.0) * P.act_grad[2] + P.act_lo[2] > 0.5;
      mo_sell = (static_cast<double>(a3) + 1.0) * P.act_grad[3] + P.act_lo[3] > 0.5;
    } else {
      mo_buy = a2 > 0.5f;
      mo_sell = a3 > 0.5f;
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
  if (MID == kMidBrownian) {
    d_mid = P.drift_dt + P.vol_sqrt_dt * nz.z;
  } else {
    d_mid = -P.ou_speed * (mid - P.ou_level) + P.vol_sqrt_dt * nz.z;
  }
  const float mid_new = mid + d_mid;

  // -- Hawkes intensities jump on arrivals, not on fills (ARR:110-119)
  float2 lam_new = lam;
  if (ARR == kArrHawkes) {
    lam_new.x = (lam.x + P.hawkes_speed * (P.hawkes_base_bid - lam.x) * P.dt) + (arr_bid ? P.hawkes_jump : 0.0f);
    lam_new.y = (lam.y + P.hawkes_speed * (P.hawkes_base_ask - lam.y) * P.dt) + (arr_ask ? P.hawkes_jump : 0.0f);
  }

  // -- reward (RW:23-33, RW:96-109, RW:128-138): incremental mark-to-market, see the header comment
  float reward = gain + q_clip * d_mid + dq_clip * mid + dc_clip;
  if (P.reward_kind != kRewPnl) {
    const float qp = pow_inventory(q_clip, P);
    reward -= P.dt * P.phi * qp;
    if (P.reward_kind == kRewRunning) {
      reward -= P.is_terminal ? P.alpha * qp : 0.0f;
    } else {
      reward -= P.alpha * ((qp - pow_inventory(q, P)) + P.dt_over_episode * pow_inventory(q_init, P));
    }
  }
  reward *= P.reward_scale;

  LaneResult r;
  r.core = make_float4(c_clip, q_clip, P.t_next, mid_new);
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

template <int MID, int ARR, int DYN, bool INJECT>
__global__ __launch_bounds__(kBlockThreads) void step_kernel(const StepBuffers B, const StepParams P) {
  const uint32_t pair = blockIdx.x * kBlockThreads + threadIdx.x;
  float r_sum = 0.0f;
  if (pair < P.n_pairs) {
    const uint32_t lane0 = 2u * pair;
    constexpr int kVecPerPair = (ARR == kArrHawkes) ? 3 : 2;  // float4 per pair of rows
    const float4* src = reinterpret_cast<const float4*>(B.state_in) + static_cast<size_t>(pair) * kVecPerPair;
    const float4 v0 = src[0], v1 = src[1];
    float4 core0 = v0, core1 = v1;
    float2 lam0 = make_float2(0.f, 0.f), lam1 = lam0;
    if (ARR == kArrHawkes) {  // rows of 6: [c q t S | lb la c q | t S lb la]
      const float4 v2 = src[2];
      lam0 = make_float2(v1.x, v1.y);
      core1 = make_float4(v1.z, v1.w, v2.x, v2.y);
      lam1 = make_float2(v2.z, v2.w);
    }
    float4 act0, act1;
    if (DYN == kDynLimit) {
      const float4 a = reinterpret_cast<const float4*>(B.action)[pair];
      act0 = make_float4(a.x, a.y, 0.f, 0.f);
      act1 = make_float4(a.z, a.w, 0.f, 0.f);
    } else {
      act0 = reinterpret_cast<const float4*>(B.action)[lane0];
      act1 = reinterpret_cast<const float4*>(B.action)[lane0 + 1];
    }
    LaneNoise nz0, nz1;
    if (INJECT) {
      const float4 ua = reinterpret_cast<const float4*>(B.u_arr)[pair];
      const float4 uf = reinterpret_cast<const float4*>(B.u_fill)[pair];
      const float2 zz = reinterpret_cast<const float2*>(B.z)[pair];
      nz0 = LaneNoise{ua.x, ua.y, uf.x, uf.y, zz.x};
      nz1 = LaneNoise{ua.z, ua.w, uf.z, uf.w, zz.y};
    } else {
      philox_pair_noise(P.pair_offset + pair, P.philox_step, P.key0, P.key1, nz0, nz1);
    }
    float qi0 = P.q_init_scalar, qi1 = P.q_init_scalar;
    if (B.q_init != nullptr) {
      const float2 qi = reinterpret_cast<const float2*>(B.q_init)[pair];
      qi0 = qi.x;
      qi1 = qi.y;
    }

    const LaneResult r0 = step_lane<MID, ARR, DYN>(core0, lam0, act0.x, act0.y, act0.z, act0.w, nz0, qi0, P);
    const LaneResult r1 = step_lane<MID, ARR, DYN>(core1, lam1, act1.x, act1.y, act1.z, act1.w, nz1, qi1, P);

    float4* dst = reinterpret_cast<float4*>(B.state_out) + static_cast<size_t>(pair) * kVecPerPair;
    if (ARR == kArrHawkes) {
      dst[0] = r0.core;
      dst[1] = make_float4(r0.lam.x, r0.lam.y, r1.core.x, r1.core.y);
      dst[2] = make_float4(r1.core.z, r1.core.w, r1.lam.x, r1.lam.y);
    } else {
      dst[0] = r0.core;
      dst[1] = r1.core;
    }
    reinterpret_cast<float2*>(B.reward)[pair] = make_float2(r0.reward, r1.reward);

    const bool second = lane0 + 1 < P.n;  // the pad lane of an odd shard is computed but never reported
    r_sum = r0.reward + (second ? r1.reward : 0.0f);

    // -- optional outputs (wave-uniform branches)
    if (B.obs != nullptr) {
      constexpr int dim = (ARR == kArrHawkes) ? 6 : 4;
      write_obs_row(B.obs, lane0, dim, r0.core, r0.lam, P);
      write_obs_row(B.obs, lane0 + 1, dim, r1.core, r1.lam, P);
    }
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
  }
  // -- per-wave running sum of rewards: the numerator of the mean episode return
  const float total = wave_sum(r_sum);
  if ((threadIdx.x & 63u) == 0u) {
    const uint32_t wave = blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6);
    B.wave_sums[wave] += static_cast<double>(total);
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
