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
// Layout: rows of D = 4 (no impact state) or D = 5 ([cash, inventory, time, midprice, y]) float32.  A workgroup of 256
// threads owns a TILE of 1024 consecutive lanes and thread j the QUAD of lanes tile*1024 + j + {0, 256, 512, 768}: as in
// the order-book kernel every wave-level access covers one contiguous span of rows (a D = 4 row is one dwordx4 per lane; a
// D = 5 row, 20 bytes and only 4-byte aligned, is five dwords per lane that walk the same cache lines), and ONE
// Philox4x32-10 block (counter word 3 = 3) feeds the two Box-Muller transforms the quad needs.  The first version gave a
// thread four ADJACENT rows (D float4 at a stride of 64-80 bytes between lanes): 11.1 / 13.5 us per step at 2^20 lanes
// (0.47 of peak) against the figures in profiles/ for this mapping.
// Algorithmic traffic per env-step: 4*(D + 1 + D + 1) = 40 B (D = 4) or 48 B (D = 5).
// Inventory is real-valued here, so "bit-exact inventory" does not apply; all state is float32 (a few ulps).
#pragma once
#include "step_kernel.hpp"

namespace mbt {

// The shape of a speed kernel, as a list of named tags like the order-book kernels' (step_kernel.hpp: Variant):
// `SpeedVariant<shape::impact_state, shape::normalised, shape::precise>`.  kernels_speed.hip holds the mapping from a configuration.
namespace shape {
struct impact_state {};  // the price impact model owns a state column y: rows of D = 5
struct powers {};        // the instantiation can raise to arbitrary powers (see POWERS below)
struct host_impact {};   // a PriceImpactModel subclass that only has HOST code
template <class T> constexpr bool known_speed = has<T, impact_state, normalised, injected, precise, powers, host_impact>;
}  // namespace shape

template <class... Tags>
struct SpeedVariant {
  static_assert((shape::known_speed<Tags> && ... && true), "unknown shape tag of a speed kernel");
  static constexpr bool HAS_IMPACT_STATE = shape::has<shape::impact_state, Tags...>, NORM = shape::has<shape::normalised, Tags...>, INJECT = shape::has<shape::injected, Tags...>;
  // precise_state: [cash, inventory, midprice, impact state] held exactly as the reference's float64 values (the row's
  // float32 + an int32 remainder each, step_kernel.hpp: exact_join) and stepped in double in the reference's operation order
  static constexpr bool PRECISE = shape::has<shape::precise, Tags...>;
  // HOST_IMPACT: a PriceImpactModel subclass that only has HOST code (MBT_IMPACT_HOST / MBT_IMPACT_HOST_STATE): the (N) float64
  // impacts its get_impact(action) returned for this step are read from StepBuffers::host_fill_p, its state column (if it owns
  // one) passes through unchanged - ITS update() advances it on the host (mbt_env_set_host_state_columns).  precise_state only.
  static constexpr bool HOST_IMPACT = shape::has<shape::host_impact, Tags...>;
  static_assert(!HOST_IMPACT || PRECISE, "host-computed price impacts are float64 values: the precise_state tier");
  // POWERS: the instantiation can raise to arbitrary powers (a temporary impact with exponent != 1, IMP:55; an inventory
  // penalty with exponent != 2, RW:59-68; exponential utility).  Every reference configuration has exponent 1 / 2: the host
  // picks the instantiation WITHOUT them then (kernels_speed.hip) - four inlined powf bodies made the kernel 3300
  // instructions (20 KB of code for a 40-byte-per-lane copy), 740 without.
  static constexpr bool POWERS = shape::has<shape::powers, Tags...>;
  static constexpr int RES = PRECISE ? 4 : 0;
  static constexpr bool PENALISED = true;  // optimal-execution rewards are almost never plain PnL: one (general) variant
  static constexpr int REWARD = kRewardGeneral;
  static constexpr int DIM = HAS_IMPACT_STATE ? 5 : 4;
};

struct QuadNoise {
  float z[4];
};

// quad stream: ctr = (quad.lo, quad.hi, step, 3), quad = (g / 1024) * 256 + g % 256 for the global lanes g, g + 256, g + 512,
// g + 768 of a 1024-lane tile; words (0,1) -> z of the first two of them, words (2,3) -> the other two
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
    case kImpactTempPower: impact = P.temp_coef * ((!V::POWERS || P.impact_exponent_is_one) ? v : power_f32(v, P.X.impact_exponent)); break;
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
  r.reward = V::POWERS ? finish_reward(pnl, s.q, q_clip, c_clip, mid_new, q_init, v, is_terminal, P)
                       : finish_reward_squares(pnl, s.q, q_clip, q_init, v, is_terminal, P);
  r.events = (dq_clip != 0.0f ? 64u : 0u) | (dc_clip != 0.0f ? 128u : 0u);
  return r;
}

// precise_state: the same lane-step on the reference's float64 state (MD:262-267, IMP:55, :87-91, :130-135, :170-175), every
// expression in the operation order of the NumPy statement it restates; rewards per reward_exact (RW:57-70 for CjOe).
struct SpeedExact {
  double cash, q, mid, y;
};
struct SpeedResultExact {
  SpeedExact next;
  float reward;
  uint32_t events;
};

template <class V>
__device__ __forceinline__ SpeedResultExact speed_lane_exact(const SpeedExact s, float a_raw, float z, float q_init, bool is_terminal,
                                                             double t_now, double t_next, const StepParams& P, double host_impact = 0.0) {
  const PreciseParams& X = P.X;
  double v = a_raw;
  if (V::NORM && P.norm_act) v = (static_cast<double>(a_raw) + 1.0) * P.act_grad[0] + P.act_lo[0];  // TE:124
  double impact, y_new = s.y;
  if (V::HOST_IMPACT) {
    impact = host_impact;  // price_impact_model.get_impact(action) (MD:263), evaluated by the caller's own class on the host
  } else switch (P.impact_kind) {
    case kImpactTempPower: impact = X.temp_coef * (V::POWERS ? numpy_power_out_of_line(v, X.impact_exponent) : numpy_power_1_or_2(v, X.impact_exponent)); break;  // IMP:55-56
    case kImpactTempPerm:
      impact = X.temp_coef * v + s.y;                         // IMP:90-91
      y_new = s.y + X.perm_coef * v * X.impact_dt;            // IMP:87-88
      break;
    case kImpactTempTransient:
      impact = X.temp_coef * v + X.trans_coef * s.y;          // IMP:134-135
      y_new = (s.y - X.resilience * s.y * X.impact_dt) + X.kernel_coef * v * X.impact_dt;  // IMP:130-132
      break;
    default:
      impact = X.trans_coef * s.y;                            // IMP:174-175
      y_new = (s.y - X.resilience * s.y * X.impact_dt) + X.kernel_coef * v * X.impact_dt;  // IMP:170-172
  }
  const double volume = v * X.speed_dt;                       // MD:265: the MIDPRICE model's step size
  const double cash_new = s.cash - volume * (s.mid + impact); // MD:263-266
  const double q_new = s.q + volume;
  const double q_clip = fmin(fmax(q_new, -X.q_max), X.q_max), c_clip = fmin(fmax(cash_new, -X.c_max), X.c_max);  // TE:283-289
  const double mid_new = midprice_step_exact(s.mid, z, 0.0, 0.0, X);
  SpeedResultExact r;
  r.next = SpeedExact{c_clip, q_clip, mid_new, y_new};
  // (POWERS = false: the host knows every exponent is 1 or 2 and the reward is not the exponential utility - the same operations
  // without pow() / exp() in the instruction stream, which is what kept this kernel at 143 registers and 3 waves per SIMD)
  // (POWERS = true: pow() / exp() out of line - one body each per kernel instead of up to sixteen: 47-59 KB and 131-133 registers inlined)
  r.reward = static_cast<float>(reward_exact<V::POWERS ? kRewardGeneral : kRewardQuadratic, true>(s.cash, s.q, s.mid, c_clip, q_clip, mid_new, q_init, v, is_terminal, t_now, t_next, P));
  r.events = (q_clip != q_new ? 64u : 0u) | (c_clip != cash_new ? 128u : 0u);
  return r;
}

constexpr uint32_t kSpeedTileLanes = 4 * kBlockThreads;

// one state row of a lane: [cash, inventory, time, midprice (, y)]
template <class V, bool NT = false>
__device__ __forceinline__ SpeedLane load_speed_row(const float* state, uint32_t lane) {
  if (V::DIM == 4) {
    const float4 r = load4<NT>(state + static_cast<size_t>(lane) * 4);
    return SpeedLane{r.x, r.y, r.w, 0.0f};
  }
  const float* r = state + static_cast<size_t>(lane) * 5;
  if (NT) return SpeedLane{__builtin_nontemporal_load(r), __builtin_nontemporal_load(r + 1), __builtin_nontemporal_load(r + 3), __builtin_nontemporal_load(r + 4)};
  return SpeedLane{r[0], r[1], r[3], r[4]};
}

template <class V, bool THROUGH = true>
__device__ __forceinline__ void store_speed_row(float* base, uint32_t lane, const SpeedLane& s, float t, bool normalise, const StepParams& P) {
  float row[5] = {s.cash, s.q, t, s.mid, s.y};
  if (normalise) {
#pragma unroll
    for (int c = 0; c < V::DIM; ++c) row[c] = normalise_column(row[c], c, P);  // TE:112-118
  }
  if (V::DIM == 4) {  // one whole row per lane: written through the L2 (step_kernel.hpp: store_through)
    if (THROUGH) store_through(reinterpret_cast<float4*>(base) + lane, make_float4(row[0], row[1], row[2], row[3]));
    else reinterpret_cast<float4*>(base)[lane] = make_float4(row[0], row[1], row[2], row[3]);
  } else {
    float* r = base + static_cast<size_t>(lane) * 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) r[c] = row[c];
  }
}

// ---- precise_state (SpeedVariant<..., PRECISE = true>) -----------------------------------------------------------------
// Same lane <-> thread mapping and the same Philox quad stream as the float32 kernels; the state row holds the float32
// rounding of the reference's float64 state and B.resid (n_pad, 4) int32 the remainders of [cash, inventory, midprice, y].
template <class V>
__device__ __forceinline__ SpeedExact load_speed_exact(const StepBuffers& B, uint32_t lane) {
  const SpeedLane f = load_speed_row<V>(B.state_in, lane);
  const ldi4_t lo = *reinterpret_cast<const ldi4_t*>(B.resid + static_cast<size_t>(lane) * 4);
  return SpeedExact{exact_join(f.cash, lo.x), exact_join(f.q, lo.y), exact_join(f.mid, lo.z), V::DIM == 5 ? exact_join(f.y, lo.w) : 0.0};
}

// row (float32 roundings) [+ remainders] of one lane; `obs`: the normalised observation row from the float64 values (TE:112-118)
template <class V, bool THROUGH>
__device__ __forceinline__ void store_speed_exact(float* state, int32_t* resid, float* obs, uint32_t lane, const SpeedExact& s, double t, const StepParams& P) {
  const double x[5] = {s.cash, s.q, t, s.mid, s.y};
  float hi[5];
  int32_t lo[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) exact_split(x[c], hi[c], lo[c]);
  hi[2] = static_cast<float>(t);
  if (state != nullptr) store_speed_row<V, THROUGH>(state, lane, SpeedLane{hi[0], hi[1], hi[3], hi[4]}, hi[2], false, P);
  if (resid != nullptr) store_through(reinterpret_cast<float4*>(resid) + lane, make_float4(__builtin_bit_cast(float, lo[0]), __builtin_bit_cast(float, lo[1]), __builtin_bit_cast(float, lo[3]), __builtin_bit_cast(float, V::DIM == 5 ? lo[4] : 0)));
  if (obs != nullptr) {
    float row[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) row[c] = P.norm_obs ? normalise_column_exact(x[c], c, P) : hi[c];
    if (V::DIM == 4) {
      reinterpret_cast<float4*>(obs)[lane] = make_float4(row[0], row[1], row[2], row[3]);
    } else {
      float* r = obs + static_cast<size_t>(lane) * 5;
#pragma unroll
      for (int c = 0; c < 5; ++c) r[c] = row[c];
    }
  }
}

// Rows of 20 bytes (D = 5) through LDS, PER WAVE.  Thread t of wave w owns the lanes 64 w + t + {0, 256, 512, 768} of the
// tile: four spans of 64 consecutive rows = 4 x 1280 B, each a whole number of 64-byte lines.  The wave moves each span as
// 80 float4 (every thread one, the first 16 a second) between HBM and ITS OWN 5 KB of LDS and picks its rows out of that -
// whole-line accesses on the memory side (five dword accesses per row walk the same lines: 7.9 vs 7.5 us for the traffic
// alone at 2^20 lanes, tools/microbench/mb_floor.hip; written through the L2 piecewise they double the step time,
// step_kernel.hpp: store_through) with NO workgroup barrier: a wave's LDS operations are executed in order, so only the
// compiler has to be told (the round-2 kernel staged the whole tile through three __syncthreads: with four workgroups on a
// CU at 2^20 lanes, waves waiting at barriers are what kept it 8 % above its floor).
constexpr int kSpanFloat4 = 64 * 5 / 4;  // float4 per span of 64 rows
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// STAGED (D = 5 only): rows LOADED through LDS as above instead of five dword loads per lane - better while the launch's
// working set is cache-resident (2^20 lanes: 7.94 us for the dword loads, 7.50 us staged, traffic alone), worse beyond
// (2^24 lanes: 134.9 vs 144.3 us the other way round).  The host picks the instantiation by size (mbt_env.hip:
// tune_for_size).  STREAM: beyond the Infinity Cache the direct loads carry the non-temporal bit, as in step_kernel.
// precise_state: the quad is advanced as kSpeedPreciseGroups groups of lanes (loads, arithmetic, stores of one group before
// the loads of the next): the double-precision step on top of sixteen loaded vectors needed 140-147 registers - three waves
// per SIMD, and the counters said that is what bounds it (waves stalled on memory 60 % of the time, profiles/r03_experiments.txt).
// (round 5, tools/dbg/r05_speed_variants.sh, speed + impact state at 2^20 lanes, two alternating runs: 4 groups 14.30 us | 2 groups
// 13.98 / 13.99 | 1 group 15.38 (139 registers, three waves) | 2 groups with five waves forced through __launch_bounds__ 16.7 and 4
// groups with six 16.7 - the forced ones spill 12 registers to scratch.  profiles/r05_experiments.txt)
#ifndef MBT_SPEED_PRECISE_GROUPS
#define MBT_SPEED_PRECISE_GROUPS 2
#endif
#ifndef MBT_SPEED_PRECISE_WAVES
#define MBT_SPEED_PRECISE_WAVES 1
#endif
// CAPTURED: the graph-capturable instantiation (step_kernel.hpp: captured_prologue / captured_epilogue) - the step's clock is read
// from device memory between the loads and the generator.
template <class V, bool STAGED = false, bool STREAM = false, bool MIRROR = false, bool CAPTURED = false>
__device__ __forceinline__ void speed_step_body(const StepBuffers& B, const StepParams& P_in, const CapturedParams* C = nullptr, CapturedStep* captured = nullptr) {
  constexpr bool kStaged = STAGED && V::DIM == 5 && !V::INJECT;
  static_assert(!(STAGED && STREAM), "the staged instantiation serves cache-resident sizes");
  static_assert(!(CAPTURED && (MIRROR || V::INJECT || V::HOST_IMPACT)), "a captured step has no host in its loop");
  if (V::POWERS && !V::PRECISE) {
    // A launch of this kernel is ONE round of waves (2^20 lanes: 4096 waves, four per SIMD, all resident at once), and with x ** p in it the vector
    // pipe has ~530 instructions per wave to issue: at equal priority the four waves of a SIMD compute side by side, finish together and store
    // together - the memory system idles while they compute and they idle while it stores.  Distinct instruction priorities make them finish
    // one after the other, so that one wave's stores overlap the next one's arithmetic: 8.15 -> 7.74 us at 2^20 lanes.  Measured and NOT
    // applied elsewhere (tools/dbg/r06_pow_variants.sh, profiles/r06_speed_variants.txt): the kernels without powers gain 0-1 %, precise_state
    // loses 2.6 %, the order-book step kernel (eight waves per SIMD, bandwidth-bound) loses 8 %.
    const uint32_t slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 3u;  // HW_REG_HW_ID bits 3:0: the wave's slot on its SIMD
    if (slot == 0u) __builtin_amdgcn_s_setprio(3);
    else if (slot == 1u) __builtin_amdgcn_s_setprio(2);
    else if (slot == 2u) __builtin_amdgcn_s_setprio(1);
  }
  clock_words_t clock_words = {0u, 0u, 0u, 0u};
  if (CAPTURED) clock_words = captured_clock_issue(&C->clock->slot[C->parity]);
  StepParams P_step;  // (CAPTURED only: the kernel arguments with this step's clock filled in by captured_prologue, below)
  if (CAPTURED) P_step = P_in;
  const StepParams& P = CAPTURED ? P_step : P_in;
  const uint32_t lane0 = blockIdx.x * kSpeedTileLanes + threadIdx.x;  // the quad: lane0 + 256 * l
  const uint64_t quad = (P.pair_offset >> 1) + blockIdx.x * kBlockThreads + threadIdx.x;
  __shared__ __attribute__((aligned(16))) float staged_rows[V::DIM == 5 ? kSpeedTileLanes * 5 : 4];  // 20 KB: 5 KB per wave
  const uint32_t wave = threadIdx.x >> 6, t = threadIdx.x & 63u;
  // float4 index of the first row of this wave's span l, in the tile (global: + tile base) and in LDS alike: span0 + 320 l
  const uint32_t span0 = (64u * wave) * 5u / 4u;
  // Schedule: every load of the quad is issued first, lane by lane; the generator (which depends on nothing in memory) runs
  // while they are in flight; then the four lanes are CONSUMED IN LOAD ORDER, each behind an empty asm that ties its own
  // loaded registers to its draw - the compiler can neither pull a consumer (and its s_waitcnt) above the generator nor
  // wait for all four lanes at once: lane l is computed and stored while the data of lanes l+1.. are still arriving.  (One
  // tie over all lanes made every wave wait for its last load before its first store: 7.3 instead of 6.8 us at 2^20 lanes.)
  SpeedLane s[4];
  float act[4], qi[4], z[4];
  ld4_t row4[4];               // D = 4: the rows as loaded (whole vectors are tied: a dead component - the time column - would otherwise
                               // be re-used as a temporary by the generator, behind a wait for the load that wrote it)
  ld4_t span_a[4], span_b[4];  // kStaged: the wave's four spans of 64 rows, 80 float4 each
  ldi4_t lo4[4];               // precise_state: the int32 remainders of [cash, inventory, midprice, y] (one 16-byte row per lane, updated in place)
  constexpr int kGroups = V::PRECISE ? MBT_SPEED_PRECISE_GROUPS : 1, kPerGroup = 4 / kGroups;
  float r_sum = 0.0f;
  uint32_t n_clipped = 0;
#pragma unroll
  for (int g = 0; g < kGroups; ++g) {
#pragma unroll
  for (int l = g * kPerGroup; l < (g + 1) * kPerGroup; ++l) {  // buffers are padded to whole tiles: no load is out of bounds
    const uint32_t lane = lane0 + l * kBlockThreads;
    if (V::PRECISE) {
      const ldi4_t* src = reinterpret_cast<const ldi4_t*>(B.resid) + lane;
      lo4[l] = STREAM ? __builtin_nontemporal_load(src) : *src;
    }
    if (kStaged) {
      const ld4_t* in4 = reinterpret_cast<const ld4_t*>(B.state_in) + static_cast<size_t>(blockIdx.x) * (kSpeedTileLanes * 5 / 4);
      span_a[l] = in4[span0 + l * 320 + t];
      span_b[l] = in4[span0 + l * 320 + 64u + (t & 15u)];  // (every thread loads: the upper 48 repeat addresses the first 16 fetch anyway)
    } else if (V::DIM == 4) {
      const ld4_t* src = reinterpret_cast<const ld4_t*>(B.state_in) + lane;
      row4[l] = STREAM ? __builtin_nontemporal_load(src) : *src;
    } else {
      s[l] = load_speed_row<V, STREAM>(B.state_in, lane);
    }
    act[l] = STREAM ? __builtin_nontemporal_load(B.action + lane) : B.action[lane];
    if (V::INJECT) z[l] = B.z[lane];
    qi[l] = P.q_init_scalar;
  }
  if (!V::INJECT && g == 0) {
    if (CAPTURED) *captured = captured_prologue(clock_words, *C, P_step);
    const QuadNoise nz = philox_quad_noise(quad, P.philox_step, P.key0, P.key1);
#pragma unroll
    for (int l = 0; l < 4; ++l) z[l] = nz.z[l];
  }
  if (B.q_init != nullptr) {  // (per-lane initial inventories: after the generator - a pointer test between the loads would put a wait there)
#pragma unroll
    for (int l = g * kPerGroup; l < (g + 1) * kPerGroup; ++l) qi[l] = B.q_init[lane0 + l * kBlockThreads];
  }
#pragma unroll
  for (int l = g * kPerGroup; l < (g + 1) * kPerGroup; ++l) {
    const uint32_t lane = lane0 + l * kBlockThreads;
    float* lds_row = staged_rows + (threadIdx.x + l * kBlockThreads) * 5;
    if (!V::INJECT) {
      if (kStaged) asm volatile("; lane is first consumed below this line" : "+v"(span_a[l]), "+v"(span_b[l]), "+v"(act[l]), "+v"(z[l]), "+v"(z[(l + 1) & 3]), "+v"(z[(l + 2) & 3]), "+v"(z[(l + 3) & 3]));
      else if (V::DIM == 4) asm volatile("; lane is first consumed below this line" : "+v"(row4[l]), "+v"(act[l]), "+v"(z[l]), "+v"(z[(l + 1) & 3]), "+v"(z[(l + 2) & 3]), "+v"(z[(l + 3) & 3]));
      else asm volatile("; lane is first consumed below this line" : "+v"(s[l].cash), "+v"(s[l].q), "+v"(s[l].mid), "+v"(s[l].y), "+v"(act[l]), "+v"(z[l]), "+v"(z[(l + 1) & 3]), "+v"(z[(l + 2) & 3]), "+v"(z[(l + 3) & 3]));
      if (V::PRECISE) asm volatile("" : "+v"(lo4[l]));
    }
    if (kStaged) {  // span l: into the wave's LDS, this thread's row out of it
      ld4_t* lds4 = reinterpret_cast<ld4_t*>(staged_rows);
      lds4[span0 + l * 320 + t] = span_a[l];
      if (t < 16u) lds4[span0 + l * 320 + 64u + t] = span_b[l];
      wave_lds_fence();
      s[l] = SpeedLane{lds_row[0], lds_row[1], lds_row[3], lds_row[4]};
      wave_lds_fence();  // every row of the span has been read before results overwrite it
    } else if (V::DIM == 4) {
      s[l] = SpeedLane{row4[l].x, row4[l].y, row4[l].w, 0.0f};
    }
    SpeedResult r;
    SpeedExact exact_next = {0.0, 0.0, 0.0, 0.0};
    if (V::PRECISE) {  // the reference's float64 arithmetic on the exactly held state (speed_lane_exact); the row keeps the float32 roundings
      const SpeedExact e = {exact_join(s[l].cash, lo4[l].x), exact_join(s[l].q, lo4[l].y), exact_join(s[l].mid, lo4[l].z), V::DIM == 5 ? exact_join(s[l].y, lo4[l].w) : 0.0};
      const SpeedResultExact rx = speed_lane_exact<V>(e, act[l], z[l], qi[l], P.is_terminal != 0, P.t_now, P.t_next_f64, P,
                                                      V::HOST_IMPACT ? B.host_fill_p[lane] : 0.0);
      exact_next = rx.next;
      int32_t lo_c, lo_q, lo_m, lo_y = 0;
      exact_split(rx.next.cash, r.next.cash, lo_c);
      exact_split(rx.next.q, r.next.q, lo_q);
      exact_split(rx.next.mid, r.next.mid, lo_m);
      r.next.y = 0.0f;
      if (V::DIM == 5) exact_split(rx.next.y, r.next.y, lo_y);
      r.reward = rx.reward;
      r.events = rx.events;
      store_through(reinterpret_cast<float4*>(B.resid) + lane, make_float4(__builtin_bit_cast(float, lo_c), __builtin_bit_cast(float, lo_q), __builtin_bit_cast(float, lo_m), __builtin_bit_cast(float, lo_y)));
    } else {
      r = speed_lane<V>(s[l], act[l], z[l], qi[l], P.is_terminal != 0, P);
    }
    if (V::DIM == 5) {  // the wave's span l leaves as whole lines, through the L2
      lds_row[0] = r.next.cash; lds_row[1] = r.next.q; lds_row[2] = P.t_next; lds_row[3] = r.next.mid; lds_row[4] = r.next.y;
      wave_lds_fence();
      const float4* lds4 = reinterpret_cast<const float4*>(staged_rows);
      float4* out4 = reinterpret_cast<float4*>(B.state_out) + static_cast<size_t>(blockIdx.x) * (kSpeedTileLanes * 5 / 4);
      store_through(out4 + span0 + l * 320 + t, lds4[span0 + l * 320 + t]);
      if (t < 16u) store_through(out4 + span0 + l * 320 + 64u + t, lds4[span0 + l * 320 + 64u + t]);
    } else {
      store_speed_row<V>(B.state_out, lane, r.next, P.t_next, false, P);
    }
    store_through(B.reward + lane, r.reward);
    if (V::NORM && B.obs != nullptr) {
      if (V::PRECISE) store_speed_exact<V, false>(nullptr, nullptr, B.obs, lane, exact_next, P.t_next_f64, P);  // normalised from the float64 values (TE:112-118)
      else store_speed_row<V>(B.obs, lane, r.next, P.t_next, P.norm_obs != 0, P);
    }
    if (B.events != nullptr) B.events[lane] = static_cast<uint8_t>(r.events);
    if (B.lane_returns != nullptr) B.lane_returns[lane] += r.reward;
    const bool real = lane < P.n;
    if (MIRROR && real) {  // small-batch host API: what env.step() returns, straight into host memory (step_kernel.hpp: signal_host)
      B.host_reward[lane] = r.reward;
      if (V::PRECISE) store_speed_exact<V, false>(nullptr, nullptr, B.host_obs, lane, exact_next, P.t_next_f64, P);
      else store_speed_row<V, false>(B.host_obs, lane, r.next, P.t_next, V::NORM && P.norm_obs != 0, P);
    }
    r_sum += real ? r.reward : 0.0f;
    n_clipped += __builtin_popcountll(__builtin_amdgcn_ballot_w64(real && r.events != 0u));
  }
  }  // groups
  const float total = wave_sum(r_sum);
  if ((threadIdx.x & 63u) == 0u) {
    const uint32_t wave_id = blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6);
    unsafeAtomicAdd(&B.wave_sums[wave_id], static_cast<double>(total));
    if (__builtin_expect(n_clipped != 0u, 0)) atomicAdd(&B.clip_count[wave_id & (kClipSlots - 1u)], static_cast<unsigned long long>(n_clipped));
  }
  if (MIRROR) signal_host(B);
}

template <class V, bool STAGED = false, bool STREAM = false, bool MIRROR = false>
__global__ __launch_bounds__(kBlockThreads, V::PRECISE ? MBT_SPEED_PRECISE_WAVES : 1) void speed_step_kernel(const StepBuffers B, const StepParams P) {
  speed_step_body<V, STAGED, STREAM, MIRROR>(B, P);
}

template <class V, bool STAGED = false, bool STREAM = false>
__global__ __launch_bounds__(kBlockThreads, V::PRECISE ? MBT_SPEED_PRECISE_WAVES : 1) void captured_speed_step_kernel(const StepBuffers B, const StepParams P,
                                                                                                                         const CapturedParams C) {
  CapturedStep s = {0.0, 0.0, 0u, false};
  speed_step_body<V, STAGED, STREAM, false, true>(B, P, &C, &s);
  captured_epilogue(B, P, C, s);
}

// Fused rollout for the speed family: fixed speed, or an open-loop schedule tabulated over time steps (e.g. the
// Cartea-Jaimungal optimal-execution speed, agents/BaselineAgents.py:173-210, which depends on time only).
template <class V>
__global__ __launch_bounds__(kBlockThreads) void speed_rollout_kernel(const StepBuffers B, const StepParams P, const RolloutParams R) {
  static_assert(!V::INJECT, "rollouts draw their own noise");
  const uint32_t lane0 = blockIdx.x * kSpeedTileLanes + threadIdx.x;
  const uint64_t quad = (P.pair_offset >> 1) + blockIdx.x * kBlockThreads + threadIdx.x;
  const size_t n_pad = static_cast<size_t>(P.n_pairs) * 2;
  SpeedLane s[4];
  float qi[4], ret[4] = {0.f, 0.f, 0.f, 0.f}, rew[4] = {0.f, 0.f, 0.f, 0.f};
  uint32_t ev[4] = {0u, 0u, 0u, 0u};
  double t = R.t_start;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const uint32_t lane = lane0 + l * kBlockThreads;
    s[l] = load_speed_row<V>(B.state_in, lane);
    qi[l] = B.q_init != nullptr ? B.q_init[lane] : P.q_init_scalar;
    if (R.obs_traj != nullptr) store_speed_row<V, false>(R.obs_traj, lane, s[l], static_cast<float>(t), V::NORM && P.norm_obs != 0, P);
  }
  float held[4] = {0.f, 0.f, 0.f, 0.f};
  if (R.policy == kPolicyBuffer) {  // action repeat: each lane keeps its entry of the action buffer
#pragma unroll
    for (int l = 0; l < 4; ++l) held[l] = B.action[lane0 + l * kBlockThreads];
    asm volatile("; loaded actions settled before the loop (step_kernel.hpp: settle_load)" : "+v"(held[0]), "+v"(held[1]), "+v"(held[2]), "+v"(held[3]));
  }
  uint32_t clipped = 0;
  for (uint32_t k = 0; k < R.n_steps; ++k) {
    const QuadNoise nz = philox_quad_noise(quad, P.philox_step + k, P.key0, P.key1);
    float speed = R.action[0];
    if (R.policy == kPolicyTimeTable) speed = reinterpret_cast<const float*>(R.table)[min(R.table_row0 + k, R.table_rows - 1u)];
    t += R.dt_f64;
    const bool terminal = (k + 1 == R.n_steps) && R.last_is_terminal != 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const uint32_t lane = lane0 + l * kBlockThreads;
      const float v = R.policy == kPolicyBuffer ? held[l] : speed;
      const SpeedResult r = speed_lane<V>(s[l], v, nz.z[l], qi[l], terminal, P);
      s[l] = r.next;
      rew[l] = r.reward;
      ev[l] = r.events;
      ret[l] += r.reward;
      clipped += (lane < P.n && r.events != 0u) ? 1u : 0u;
      if (R.obs_traj != nullptr)
        store_speed_row<V, false>(R.obs_traj + static_cast<size_t>(k + 1) * n_pad * V::DIM, lane, s[l], static_cast<float>(t), V::NORM && P.norm_obs != 0, P);
      if (R.act_traj != nullptr) R.act_traj[static_cast<size_t>(k) * n_pad + lane] = v;
      if (R.rew_traj != nullptr) R.rew_traj[static_cast<size_t>(k) * n_pad + lane] = r.reward;
    }
  }
  float ret_sum = 0.0f;
#pragma unroll
  for (int l = 0; l < 4; ++l) {  // what step() leaves behind: final state, rewards (and events) of the final step
    const uint32_t lane = lane0 + l * kBlockThreads;
    store_speed_row<V>(B.state_out, lane, s[l], static_cast<float>(t), false, P);
    if (V::NORM && B.obs != nullptr) store_speed_row<V>(B.obs, lane, s[l], static_cast<float>(t), P.norm_obs != 0, P);
    if (R.n_steps > 0) {
      B.reward[lane] = rew[l];
      if (B.events != nullptr) B.events[lane] = static_cast<uint8_t>(ev[l]);
    }
    if (B.lane_returns != nullptr) B.lane_returns[lane] += ret[l];
    ret_sum += lane < P.n ? ret[l] : 0.0f;
  }
  if (__builtin_expect(clipped != 0u, 0)) atomicAdd(&B.clip_count[blockIdx.x & (kClipSlots - 1u)], static_cast<unsigned long long>(clipped));
  const float total = wave_sum(ret_sum);
  if ((threadIdx.x & 63u) == 0u) {
    const uint32_t wave = blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6);
    unsafeAtomicAdd(&B.wave_sums[wave], static_cast<double>(total));
  }
}

template <class V>
__global__ __launch_bounds__(kBlockThreads) void speed_rollout_exact_kernel(const StepBuffers B, const StepParams P, const RolloutParams R) {
  static_assert(V::PRECISE && !V::INJECT, "rollouts draw their own noise");
  const uint32_t lane0 = blockIdx.x * kSpeedTileLanes + threadIdx.x;
  const uint64_t quad = (P.pair_offset >> 1) + blockIdx.x * kBlockThreads + threadIdx.x;
  const size_t n_pad = static_cast<size_t>(P.n_pairs) * 2;
  SpeedExact s[4];
  float qi[4], ret[4] = {0.f, 0.f, 0.f, 0.f}, rew[4] = {0.f, 0.f, 0.f, 0.f};
  uint32_t ev[4] = {0u, 0u, 0u, 0u};
  double t = R.t_start;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const uint32_t lane = lane0 + l * kBlockThreads;
    s[l] = load_speed_exact<V>(B, lane);
    qi[l] = B.q_init != nullptr ? B.q_init[lane] : P.q_init_scalar;
    if (R.obs_traj != nullptr) store_speed_exact<V, false>(nullptr, nullptr, R.obs_traj, lane, s[l], t, P);
  }
  float held[4] = {0.f, 0.f, 0.f, 0.f};
  if (R.policy == kPolicyBuffer) {
#pragma unroll
    for (int l = 0; l < 4; ++l) held[l] = B.action[lane0 + l * kBlockThreads];
    asm volatile("; loaded actions settled before the loop (step_kernel.hpp: settle_load)" : "+v"(held[0]), "+v"(held[1]), "+v"(held[2]), "+v"(held[3]));
  }
  uint32_t clipped = 0;
  for (uint32_t k = 0; k < R.n_steps; ++k) {
    const QuadNoise nz = philox_quad_noise(quad, P.philox_step + k, P.key0, P.key1);
    float speed = R.action[0];
    if (R.policy == kPolicyTimeTable) speed = reinterpret_cast<const float*>(R.table)[min(R.table_row0 + k, R.table_rows - 1u)];
    const double t_now = t;
    t += R.dt_f64;
    const bool terminal = (k + 1 == R.n_steps) && R.last_is_terminal != 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const uint32_t lane = lane0 + l * kBlockThreads;
      const float v = R.policy == kPolicyBuffer ? held[l] : speed;
      const SpeedResultExact r = speed_lane_exact<V>(s[l], v, nz.z[l], qi[l], terminal, t_now, t, P);
      s[l] = r.next;
      rew[l] = r.reward;
      ev[l] = r.events;
      ret[l] += r.reward;
      clipped += (lane < P.n && r.events != 0u) ? 1u : 0u;
      if (R.obs_traj != nullptr) store_speed_exact<V, false>(nullptr, nullptr, R.obs_traj + static_cast<size_t>(k + 1) * n_pad * V::DIM, lane, s[l], t, P);
      if (R.act_traj != nullptr) R.act_traj[static_cast<size_t>(k) * n_pad + lane] = v;
      if (R.rew_traj != nullptr) R.rew_traj[static_cast<size_t>(k) * n_pad + lane] = r.reward;
    }
  }
  float ret_sum = 0.0f;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const uint32_t lane = lane0 + l * kBlockThreads;
    store_speed_exact<V, false>(B.state_out, B.resid, B.obs, lane, s[l], t, P);
    if (R.n_steps > 0) {
      B.reward[lane] = rew[l];
      if (B.events != nullptr) B.events[lane] = static_cast<uint8_t>(ev[l]);
    }
    if (B.lane_returns != nullptr) B.lane_returns[lane] += ret[l];
    ret_sum += lane < P.n ? ret[l] : 0.0f;
  }
  if (__builtin_expect(clipped != 0u, 0)) atomicAdd(&B.clip_count[blockIdx.x & (kClipSlots - 1u)], static_cast<unsigned long long>(clipped));
  const float total = wave_sum(ret_sum);
  if ((threadIdx.x & 63u) == 0u) {
    const uint32_t wave = blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6);
    unsafeAtomicAdd(&B.wave_sums[wave], static_cast<double>(total));
  }
}

#ifndef MBT_KERNEL_TU  // (a non-template kernel: defined once, in the host translation unit)
// the quad stream's normals, written out for tests: tile-split like the kernels (lane = tile * 1024 + slot + 256 * l)
__global__ void rng_fill_quad_kernel(uint64_t quad_offset, uint32_t step, uint32_t k0, uint32_t k1, uint32_t n_quads, float* z) {
  const uint32_t quad = blockIdx.x * blockDim.x + threadIdx.x;
  if (quad >= n_quads) return;
  const QuadNoise nz = philox_quad_noise(quad_offset + quad, step, k0, k1);
  const uint32_t lane0 = (quad / kBlockThreads) * kSpeedTileLanes + quad % kBlockThreads;
#pragma unroll
  for (int l = 0; l < 4; ++l) z[lane0 + l * kBlockThreads] = nz.z[l];
}
#endif  // MBT_KERNEL_TU

}  // namespace mbt
