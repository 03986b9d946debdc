// Trading-with-speed (optimal execution) dynamics: the fused step and rollout kernels for gfx950.
//
// Reference: TradinghWithSpeedModelDynamics (gym/ModelDynamics.py:243-275) with the price-impact models of
// stochastic_processes/price_impact_models.py:34-179 and, typically, CjOeCriterion (rewards/RewardFunctions.py:39-74).
// The action is ONE number per lane, the trading speed v (positive buys):
//     impact     = temp * v^e | temp * v + y | temp * v + kappa * y | kappa * y           (IMP:55, :90, :134, :174)
//     volume     = v * dt_mid      (the MIDPRICE model's step size, MD:265)
//     cash      -= volume * (S + impact);   inventory += volume                            (MD:262-267)
//     y         <- y + perm * v * dt_imp    |   y - rho * y * dt_imp + gamma * v * dt_imp  (IMP:87, IMP:130, IMP:170)
// There are no arrivals or fills (MD:47-48): the only noise is the midprice's normal draw.
//
// Layout: rows of D = 4 (no impact state) or D = 5 ([cash, inventory, time, midprice, y]) float32; one GPU thread
// owns a QUAD of adjacent lanes = D float4 of state, one float4 of actions, one float4 of rewards - every access is a
// 16-byte coalesced vector, and ONE Philox4x32-10 block (counter word 3 = 3) feeds the two Box-Muller transforms the
// quad needs.  Algorithmic traffic per env-step: 4*(D + 1 + D + 1) = 40 B (D = 4) or 48 B (D = 5).
// Inventory is real-valued here, so "bit-exact inventory" does not apply; all state is float32 (a few ulps).
#pragma once
#include "step_kernel.hpp"

namespace mbt {

template <bool HAS_IMPACT_STATE_, bool NORM_, bool INJECT_>
struct SpeedVariant {
  static constexpr bool HAS_IMPACT_STATE = HAS_IMPACT_STATE_, NORM = NORM_, INJECT = INJECT_;
  static constexpr bool PENALISED = true;  // optimal-execution rewards are almost never plain PnL: one (general) variant
  static constexpr int REWARD = kRewardGeneral;
  static constexpr int DIM = HAS_IMPACT_STATE_ ? 5 : 4;
};

struct QuadNoise {
  float z[4];
};

// quad stream: ctr = (quad.lo, quad.hi, step, 3); words (0,1) -> z of lanes 4q, 4q+1; words (2,3) -> lanes 4q+2, 4q+3
__device__ __forceinline__ QuadNoise philox_quad_noise(uint64_t quad, uint32_t step, uint32_t k0, uint32_t k1) {
  const PhiloxWords w = philox4x32_10(static_cast<uint32_t>(quad), static_cast<uint32_t>(quad >> 32), step, 3u, k0, k1);
  QuadNoise nz;
  box_muller(w.w0, w.w1, nz.z[0], nz.z[1]);
  box_muller(w.w2, w.w3, nz.z[2], nz.z[3]);
  return nz;
}

struct SpeedLane {
  float cash, q, mid, y;
};

struct SpeedResult {
  SpeedLane next;
  float reward;
  uint32_t events;  // bit6 inventory clipped, bit7 cash clipped
};

template <class V>
__device__ __forceinline__ SpeedResult speed_lane(const SpeedLane s, float a_raw, float z, float q_init, bool is_terminal,
                                                  const StepParams& P) {
  float v = a_raw;
  if (V::NORM && P.norm_act) v = static_cast<float>((static_cast<double>(a_raw) + 1.0) * P.act_grad[0] + P.act_lo[0]);  // TE:124
  float impact, y_new = s.y;
  switch (P.impact_kind) {
    case kImpactTempPower: impact = P.temp_coef * (P.impact_exponent_is_one ? v : powf(v, P.impact_exponent)); break;
    case kImpactTempPerm:
      impact = P.temp_coef * v + s.y;
      y_new = s.y + P.perm_coef * v * P.impact_dt;
      break;
    case kImpactTempTransient:
      impact = P.temp_coef * v + P.trans_coef * s.y;
      y_new = (s.y - P.resilience * s.y * P.impact_dt) + P.kernel_coef * v * P.impact_dt;
      break;
    default:
      impact = P.trans_coef * s.y;
      y_new = (s.y - P.resilience * s.y * P.impact_dt) + P.kernel_coef * v * P.impact_dt;
  }
  const float volume = v * P.speed_dt;
  const float cash_new = s.cash - volume * (s.mid + impact);
  const float q_new = s.q + volume;
  const float q_clip = __builtin_fminf(__builtin_fmaxf(q_new, -P.q_max), P.q_max);  // TE:283-289
  const float c_clip = __builtin_fminf(__builtin_fmaxf(cash_new, -P.c_max), P.c_max);
  const float dq_clip = q_clip - q_new, dc_clip = c_clip - cash_new;
  const float d_mid = midprice_increment(s.mid, __builtin_fmaf(P.vol_sqrt_dt, z, P.drift_dt), 0.0f, 0.0f, P);
  const float mid_new = s.mid + d_mid;
  // mark-to-market change with the S terms cancelled: dc + dq * S = -volume * impact
  const float pnl = -volume * impact + q_clip * d_mid + dq_clip * s.mid + dc_clip;
  SpeedResult r;
  r.next = SpeedLane{c_clip, q_clip, mid_new, y_new};
  r.reward = finish_reward(pnl, s.q, q_clip, c_clip, mid_new, q_init, v, is_terminal, P);
  r.events = (dq_clip != 0.0f ? 64u : 0u) | (dc_clip != 0.0f ? 128u : 0u);
  return r;
}

// quad <-> D float4
template <class V>
__device__ __forceinline__ void unpack_quad(const float4* src, SpeedLane (&s)[4]) {
  float f[4 * V::DIM];
#pragma unroll
  for (int j = 0; j < V::DIM; ++j) {
    const float4 v = src[j];
    f[4 * j] = v.x; f[4 * j + 1] = v.y; f[4 * j + 2] = v.z; f[4 * j + 3] = v.w;
  }
#pragma unroll
  for (int l = 0; l < 4; ++l) s[l] = SpeedLane{f[l * V::DIM], f[l * V::DIM + 1], f[l * V::DIM + 3], V::HAS_IMPACT_STATE ? f[l * V::DIM + 4] : 0.0f};
}

template <class V>
__device__ __forceinline__ void pack_quad(float4* dst, const SpeedLane (&s)[4], float t, bool normalise, const StepParams& P) {
  float f[4 * V::DIM];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    float row[5] = {s[l].cash, s[l].q, t, s[l].mid, s[l].y};
    if (normalise) {
#pragma unroll
      for (int c = 0; c < V::DIM; ++c) row[c] = (row[c] - P.obs_lo[c]) / P.obs_grad[c] - 1.0f;  // TE:112-118
    }
#pragma unroll
    for (int c = 0; c < V::DIM; ++c) f[l * V::DIM + c] = row[c];
  }
#pragma unroll
  for (int j = 0; j < V::DIM; ++j) dst[j] = make_float4(f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3]);
}

__device__ __forceinline__ float lane_of(const float4 v, int l) { return l == 0 ? v.x : l == 1 ? v.y : l == 2 ? v.z : v.w; }

template <class V>
__global__ __launch_bounds__(kBlockThreads) void speed_step_kernel(const StepBuffers B, const StepParams P) {
  const uint32_t quad = blockIdx.x * kBlockThreads + threadIdx.x;
  const uint32_t n_quads = (P.n_pairs + 1) / 2;
  float r_sum = 0.0f;
  if (quad < n_quads) {
    const float4* src = reinterpret_cast<const float4*>(B.state_in) + static_cast<size_t>(quad) * V::DIM;
    float4 loaded[V::DIM];
#pragma unroll
    for (int j = 0; j < V::DIM; ++j) loaded[j] = src[j];
    const float4 act = reinterpret_cast<const float4*>(B.action)[quad];
    float4 qi = make_float4(P.q_init_scalar, P.q_init_scalar, P.q_init_scalar, P.q_init_scalar);
    if (B.q_init != nullptr) qi = reinterpret_cast<const float4*>(B.q_init)[quad];
    QuadNoise nz;
    if (V::INJECT) {
      const float4 zz = reinterpret_cast<const float4*>(B.z)[quad];
      nz = QuadNoise{{zz.x, zz.y, zz.z, zz.w}};
    } else {
      nz = philox_quad_noise((P.pair_offset >> 1) + quad, P.philox_step, P.key0, P.key1);
    }
    SpeedLane s[4];
    unpack_quad<V>(loaded, s);
    float rew[4];
    uint32_t clipped = 0, ev = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const SpeedResult r = speed_lane<V>(s[l], lane_of(act, l), nz.z[l], lane_of(qi, l), P.is_terminal != 0, P);
      s[l] = r.next;
      rew[l] = r.reward;
      const bool real = 4u * quad + l < P.n;
      r_sum += real ? r.reward : 0.0f;
      clipped += (real && r.events != 0u) ? 1u : 0u;
      ev |= r.events << (8 * l);
    }
    pack_quad<V>(reinterpret_cast<float4*>(B.state_out) + static_cast<size_t>(quad) * V::DIM, s, P.t_next, false, P);
    reinterpret_cast<float4*>(B.reward)[quad] = make_float4(rew[0], rew[1], rew[2], rew[3]);
    if (V::NORM && B.obs != nullptr) pack_quad<V>(reinterpret_cast<float4*>(B.obs) + static_cast<size_t>(quad) * V::DIM, s, P.t_next, P.norm_obs != 0, P);
    if (B.events != nullptr) reinterpret_cast<uint32_t*>(B.events)[quad] = ev;
    if (B.lane_returns != nullptr) {
      float4 acc = reinterpret_cast<float4*>(B.lane_returns)[quad];
      acc.x += rew[0]; acc.y += rew[1]; acc.z += rew[2]; acc.w += rew[3];
      reinterpret_cast<float4*>(B.lane_returns)[quad] = acc;
    }
    if (__builtin_expect(clipped != 0u, 0)) atomicAdd(&B.clip_count[blockIdx.x & (kClipSlots - 1u)], static_cast<unsigned long long>(clipped));
  }
  const float total = wave_sum(r_sum);
  if ((threadIdx.x & 63u) == 0u) {
    const uint32_t wave = blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6);
    unsafeAtomicAdd(&B.wave_sums[wave], static_cast<double>(total));
  }
}

// Fused rollout for the speed family: fixed speed, or an open-loop schedule tabulated over time steps (e.g. the
// Cartea-Jaimungal optimal-execution speed, agents/BaselineAgents.py:173-210, which depends on time only).
template <class V>
__global__ __launch_bounds__(kBlockThreads) void speed_rollout_kernel(const StepBuffers B, const StepParams P, const RolloutParams R) {
  static_assert(!V::INJECT, "rollouts draw their own noise");
  const uint32_t quad = blockIdx.x * kBlockThreads + threadIdx.x;
  const uint32_t n_quads = (P.n_pairs + 1) / 2;
  float ret_sum = 0.0f;
  if (quad < n_quads) {
    const size_t n_pad4 = static_cast<size_t>(n_quads) * 4;
    const float4* src = reinterpret_cast<const float4*>(B.state_in) + static_cast<size_t>(quad) * V::DIM;
    float4 loaded[V::DIM];
#pragma unroll
    for (int j = 0; j < V::DIM; ++j) loaded[j] = src[j];
    SpeedLane s[4];
    unpack_quad<V>(loaded, s);
    float4 qi = make_float4(P.q_init_scalar, P.q_init_scalar, P.q_init_scalar, P.q_init_scalar);
    if (B.q_init != nullptr) qi = reinterpret_cast<const float4*>(B.q_init)[quad];
    float ret[4] = {0.f, 0.f, 0.f, 0.f}, rew[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t clipped = 0;
    double t = R.t_start;
    if (R.obs_traj != nullptr) pack_quad<V>(reinterpret_cast<float4*>(R.obs_traj) + static_cast<size_t>(quad) * V::DIM, s, static_cast<float>(t), V::NORM && P.norm_obs != 0, P);
    for (uint32_t k = 0; k < R.n_steps; ++k) {
      const QuadNoise nz = philox_quad_noise((P.pair_offset >> 1) + quad, P.philox_step + k, P.key0, P.key1);
      float speed = R.action[0];
      if (R.policy == kPolicyTimeTable) speed = reinterpret_cast<const float*>(R.table)[min(R.table_row0 + k, R.table_rows - 1u)];
      t += R.dt_f64;
      const bool terminal = (k + 1 == R.n_steps) && R.last_is_terminal != 0;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const SpeedResult r = speed_lane<V>(s[l], speed, nz.z[l], lane_of(qi, l), terminal, P);
        s[l] = r.next;
        rew[l] = r.reward;
        ret[l] += r.reward;
        clipped += (4u * quad + l < P.n && r.events != 0u) ? 1u : 0u;
      }
      if (R.obs_traj != nullptr)
        pack_quad<V>(reinterpret_cast<float4*>(R.obs_traj + static_cast<size_t>(k + 1) * n_pad4 * V::DIM) + static_cast<size_t>(quad) * V::DIM, s,
                     static_cast<float>(t), V::NORM && P.norm_obs != 0, P);
      if (R.act_traj != nullptr) reinterpret_cast<float4*>(R.act_traj + static_cast<size_t>(k) * n_pad4)[quad] = make_float4(speed, speed, speed, speed);
      if (R.rew_traj != nullptr) reinterpret_cast<float4*>(R.rew_traj + static_cast<size_t>(k) * n_pad4)[quad] = make_float4(rew[0], rew[1], rew[2], rew[3]);
    }
    pack_quad<V>(reinterpret_cast<float4*>(B.state_out) + static_cast<size_t>(quad) * V::DIM, s, static_cast<float>(t), false, P);
    reinterpret_cast<float4*>(B.reward)[quad] = make_float4(rew[0], rew[1], rew[2], rew[3]);
    if (V::NORM && B.obs != nullptr) pack_quad<V>(reinterpret_cast<float4*>(B.obs) + static_cast<size_t>(quad) * V::DIM, s, static_cast<float>(t), P.norm_obs != 0, P);
    if (B.lane_returns != nullptr) {
      float4 acc = reinterpret_cast<float4*>(B.lane_returns)[quad];
      acc.x += ret[0]; acc.y += ret[1]; acc.z += ret[2]; acc.w += ret[3];
      reinterpret_cast<float4*>(B.lane_returns)[quad] = acc;
    }
    if (__builtin_expect(clipped != 0u, 0)) atomicAdd(&B.clip_count[blockIdx.x & (kClipSlots - 1u)], static_cast<unsigned long long>(clipped));
#pragma unroll
    for (int l = 0; l < 4; ++l) ret_sum += (4u * quad + l < P.n) ? ret[l] : 0.0f;
  }
  const float total = wave_sum(ret_sum);
  if ((threadIdx.x & 63u) == 0u) {
    const uint32_t wave = blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6);
    unsafeAtomicAdd(&B.wave_sums[wave], static_cast<double>(total));
  }
}

// the quad stream's normals, written out for tests
__global__ void rng_fill_quad_kernel(uint64_t quad_offset, uint32_t step, uint32_t k0, uint32_t k1, uint32_t n_quads, float* z) {
  const uint32_t quad = blockIdx.x * blockDim.x + threadIdx.x;
  if (quad >= n_quads) return;
  const QuadNoise nz = philox_quad_noise(quad_offset + quad, step, k0, k1);
  reinterpret_cast<float4*>(z)[quad] = make_float4(nz.z[0], nz.z[1], nz.z[2], nz.z[3]);
}

}  // namespace mbt
