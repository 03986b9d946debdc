// Learned policies evaluated INSIDE the kernels: a linear map, or a two-hidden-layer MLP of
// width <= 64 - the shape of Stable-Baselines3's MlpPolicy actor ([64, 64], tanh), the consumer the reference trains
// (agents/SbAgent.py, experiments/helpers.py:63-96) - acting on the observation the environment would hand out.
//
// This is the one dense contraction on the path, hence the one use of the matrix cores: per environment step a wave
// evaluates 128 rows (64 threads x the 2 lanes each owns) x [D -> 64 -> 64 -> A] = 8960 flop per row on
// v_mfma_f32_16x16x16_f16 (first layer) and v_mfma_f32_16x16x32_f16 (the other two): fp16 operands, fp32 accumulate.  Formulated TRANSPOSED, H^T = W X^T: the weights are the
// A operand (M = output features; the fragments live in LDS and are streamed into the tile loop), the batch rows are N, and the
// accumulator layout of one layer (lane L: rows m = 4 (L / 16) + r, r = 0..3, column n = L % 16) IS the B-operand layout
// of the next (lane L: k = 4 (L / 16) + j, n = L % 16), so hidden activations never leave registers: activation,
// convert to fp16, feed the next MFMA.  Only the observations (in) and the actions (out) cross lanes, through 6 KB of
// LDS per wave.  Biases are free: the first layer's rides on a constant-one input feature, the others initialise the
// accumulators.
//
// Numerics: operands are rounded to fp16 (observations normalised to [-1, 1] or raw, weights, hidden activations);
// products and sums are fp32.  tests/test_gpu_policy.py compares with a NumPy restatement that rounds where the kernel
// rounds (<= 2e-3 on actions of magnitude ~1) and with the plain fp32 network (<= 2e-2).  The policy kernel
// (policy_kernel, used by a step loop) and the fused rollout run THIS code on the same values, so a rollout remains
// bit-identical to "evaluate policy, step" repeated.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "philox.hpp"

namespace mbt {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float acc4_t __attribute__((ext_vector_type(4)));

constexpr int kMlpHidden = 64;              // hidden width (narrower networks are zero-padded by the host)
constexpr int kMlpTiles = kMlpHidden / 16;  // 16-feature tiles per hidden layer
constexpr int kMlpInPad = 16;               // observation features + the constant one, padded to one K step
// the observation row is an fp16 operand: a column bounded by this is resolved to 2e-3 or better (normalised ones, in [-1, 1],
// to 5e-4); anything wider is refused by the host (mbt_env.hip: prepare_learned_policy)
constexpr float kMlpMaxObservationBound = 4.0f;
constexpr int kMlpRowsPerWave = 128;        // 64 threads x 2 lanes
enum : int { kActTanh = 0, kActRelu = 1 };

// What the host uploads (mbt_env.hip: pack_mlp): fragments in MFMA A-operand order, one 8-byte entry per lane.
//   w1[mt][lane]      4 halfs: W1p[16 mt + lane % 16][4 (lane / 16) + j]          W1p = [W1 | b1 | 0]  (64 x 16), K = 16 form
//   w2[mt][c][lane]   8 halfs: W2 [16 mt + lane % 16][16 (2c + j / 4) + 4 (lane / 16) + j % 4]      (64 x 64), K = 32 form
//   w3[c][lane]       8 halfs: W3p[lane % 16][16 (2c + j / 4) + 4 (lane / 16) + j % 4]    W3p = rows >= A zero (16 x 64)
//   b2[64], b3[16] float32; linear policies: lin_w[A][8], lin_b[4] float32
// (8-half entries are the operands of the K = 32 instruction as they are: two 16-feature chunks side by side, see `pair`)
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
struct MlpDeviceWeights {
  const half4_t* w1;   // [4][64]
  const half8_t* w2;   // [4][2][64]
  const half8_t* w3;   // [2][64]
  const float* b2;     // [64]
  const float* b3;     // [16]
  const float* lin_w;  // [4][8]  (linear policy)
  const float* lin_b;  // [4]
};

struct LearnedPolicyParams {
  MlpDeviceWeights w;
  int32_t is_linear;    // 1: action = clip(lin_w obs + lin_b); 0: the MLP
  int32_t activation;   // kActTanh / kActRelu
  int32_t obs_dim, act_dim;
  float act_lo[4], act_hi[4];  // the action space the agent acts in: outputs are clipped to it, as SB3 does before env.step
  // Exploration, for consumers that COLLECT training data (SB3's PPO: a state-independent std per action dimension;
  // the reference's PolicyGradientAgent, agents/PolicyGradientAgent.py:34-47: one scalar): action = mean + std * eps,
  // eps ~ N(0, 1) from Philox blocks of their own (counter word 3 = 8, 9: independent of the environment's draws).
  float act_std[4];     // all zero = deterministic
  int32_t stochastic;   // any act_std != 0
  int32_t clip;         // 1: clip to [act_lo, act_hi] after the noise (SB3 before env.step); 0: pass the sample on (PolicyGradientAgent)
};

// eps of the two lanes of a pair: block (pair, step, 8) -> two Box-Muller transforms -> lane a: (e0, e1), lane b: (e2, e3);
// a second block (word 3 = 9) for the third and fourth action components of limit + market dynamics.
__device__ __forceinline__ void explore_and_clip(const LearnedPolicyParams& L, uint64_t pair, uint32_t step, uint32_t k0, uint32_t k1, float (&act)[2][4]) {
  if (L.stochastic) {
    const uint32_t plo = static_cast<uint32_t>(pair), phi = static_cast<uint32_t>(pair >> 32);
    const PhiloxWords w = philox4x32_10(plo, phi, step, 8u, k0, k1);
    float e[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    box_muller(w.w0, w.w1, e[0][0], e[0][1]);
    box_muller(w.w2, w.w3, e[1][0], e[1][1]);
    if (L.act_dim > 2) {
      const PhiloxWords v = philox4x32_10(plo, phi, step, 9u, k0, k1);
      box_muller(v.w0, v.w1, e[0][2], e[0][3]);
      box_muller(v.w2, v.w3, e[1][2], e[1][3]);
    }
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int a = 0; a < 4; ++a) act[l][a] = __builtin_fmaf(L.act_std[a], e[l][a], act[l][a]);
  }
  if (L.clip) {
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int a = 0; a < 4; ++a) act[l][a] = __builtin_amdgcn_fmed3f(act[l][a], L.act_lo[a], L.act_hi[a]);
  }
}

// The network as the workgroup keeps it in LDS (12.3 KB, staged once per launch): the fragments in the host's order, so a
// lane's operand of one MFMA is ONE ds_read_b64 / ds_read_b128 at [fragment][lane].  The tile loop streams fragments from
// there just before the instruction that consumes them - a tile then holds ~40 registers of network state instead of
// 68 for the whole step - and the bias reads land directly in the accumulators.
constexpr int kMlpLdsW1 = 0;                                              // [4][64] half4
constexpr int kMlpLdsW2 = kMlpLdsW1 + kMlpTiles * 64 * 8;                 // [4][2][64] half8
constexpr int kMlpLdsW3 = kMlpLdsW2 + kMlpTiles * (kMlpTiles / 2) * 64 * 16;  // [2][64] half8
constexpr int kMlpLdsB2 = kMlpLdsW3 + (kMlpTiles / 2) * 64 * 16;          // [64] float
constexpr int kMlpLdsB3 = kMlpLdsB2 + kMlpHidden * 4;                     // [16] float
constexpr int kMlpLdsWeightBytes = kMlpLdsB3 + 16 * 4;

// every thread of the workgroup (kBlockThreads = 256) calls this once, before the first evaluation
__device__ __forceinline__ void stage_mlp_weights(const MlpDeviceWeights& w, char* lds) {
  static_assert(kMlpTiles * 64 == 256 && kMlpTiles * (kMlpTiles / 2) * 64 == 512, "the copy below is laid out for 256 threads and a 64-wide network");
  const int t = threadIdx.x;
  reinterpret_cast<half4_t*>(lds + kMlpLdsW1)[t] = w.w1[t];
  reinterpret_cast<half8_t*>(lds + kMlpLdsW2)[t] = w.w2[t];
  reinterpret_cast<half8_t*>(lds + kMlpLdsW2)[t + 256] = w.w2[t + 256];
  if (t < (kMlpTiles / 2) * 64) reinterpret_cast<half8_t*>(lds + kMlpLdsW3)[t] = w.w3[t];
  if (t < kMlpHidden) reinterpret_cast<float*>(lds + kMlpLdsB2)[t] = w.b2[t];
  if (t < 16) reinterpret_cast<float*>(lds + kMlpLdsB3)[t] = w.b3[t];
  __syncthreads();
}

template <int ACT>
__device__ __forceinline__ float activate(float x) {
  if (ACT == kActRelu) return __builtin_fmaxf(x, 0.0f);
  // tanh(x) = 1 - 2 / (1 + e^{2x}) on the hardware exp2 / rcp (two transcendental instructions); saturates cleanly at +-1
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// The same tanh on two values: what is not a transcendental runs on the packed fp32 forms (v_pk_mul_f32, v_pk_add_f32,
// v_pk_fma_f32: two values per issue slot).  Every operation is the IEEE operation of `activate` - 2r is exact, so
// fma(r, -2, 1) rounds once exactly like 1 - 2r - hence the SAME bits; per value 2 plain + 2 transcendental issue slots
// instead of 3.5 + 2 (the compiler packs only the tail of the scalar form: a literal cannot be a packed operand).
typedef float float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2_t tanh_pair(float2_t x) {
  const float2_t two_log2e = {2.8853900817779268f, 2.8853900817779268f};
  const float2_t t = x * two_log2e;
  const float2_t e = float2_t{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + float2_t{1.0f, 1.0f};
  const float2_t r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
  return __builtin_elementwise_fma(r, float2_t{-2.0f, -2.0f}, float2_t{1.0f, 1.0f});
}

// Two accumulator tiles (features 16 (2c) .. and 16 (2c + 1) ..) -> activation -> ONE 8-half operand of the K = 32 instruction.
template <int ACT>
__device__ __forceinline__ half8_t activate_pair(acc4_t lo, acc4_t hi) {
  if (ACT == kActRelu) {  // round first, then max(x, 0) on packed halves (v_pk_max_f16: two values per instruction); same result
    const half8_t h = {static_cast<_Float16>(lo.x), static_cast<_Float16>(lo.y), static_cast<_Float16>(lo.z), static_cast<_Float16>(lo.w),
                       static_cast<_Float16>(hi.x), static_cast<_Float16>(hi.y), static_cast<_Float16>(hi.z), static_cast<_Float16>(hi.w)};
    const half8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_elementwise_max(h, zero);
  }
  const float2_t a = tanh_pair(float2_t{lo.x, lo.y}), b = tanh_pair(float2_t{lo.z, lo.w});
  const float2_t c = tanh_pair(float2_t{hi.x, hi.y}), d = tanh_pair(float2_t{hi.z, hi.w});
  return half8_t{static_cast<_Float16>(a.x), static_cast<_Float16>(a.y), static_cast<_Float16>(b.x), static_cast<_Float16>(b.y),
                 static_cast<_Float16>(c.x), static_cast<_Float16>(c.y), static_cast<_Float16>(d.x), static_cast<_Float16>(d.y)};
}

// Two 16-feature fragments side by side = the 8-element operand of the K = 32 instruction.  The hardware pairs element
// (lane group g = lane / 16, j) of A with element (g, j) of B, whatever k those stand for - so as long as the weight
// fragment is assembled the same way (elements 0..3 from K-chunk 2c, 4..7 from chunk 2c + 1, both at 4 g + j; the host
// packs them so), the accumulator layout of the previous layer still IS the operand layout of this one, and the hidden
// layers run on v_mfma_f32_16x16x32_f16 at twice the rate of the K = 16 form.

// LDS scratch of one wave: 128 observation rows of 16 halfs (features, the constant one, zeros) + 128 action rows of 4 floats
constexpr int kMlpLdsBytesPerWave = kMlpRowsPerWave * (kMlpInPad * 2 + 16);

// obs[l][c]: the observation rows of this thread's two lanes (columns >= obs_dim ignored).  act[l][a]: the actor's mean (unclipped).
// Every lane of the wave must call this together (MFMA and the wave-level LDS exchange).
template <int ACT>
__device__ __forceinline__ void mlp_forward_wave_act(const char* W, const LearnedPolicyParams& L, const float (&obs)[2][8], float (&act)[2][4],
                                                     char* lds_wave) {
  const int lane = threadIdx.x & 63;
  const half4_t* w1 = reinterpret_cast<const half4_t*>(W + kMlpLdsW1) + lane;
  const half8_t* w2 = reinterpret_cast<const half8_t*>(W + kMlpLdsW2) + lane;
  const half8_t* w3 = reinterpret_cast<const half8_t*>(W + kMlpLdsW3) + lane;
  const acc4_t* b2 = reinterpret_cast<const acc4_t*>(W + kMlpLdsB2) + (lane >> 4);  // accumulator rows of this lane: 4 (lane / 16) + r
  const acc4_t* b3 = reinterpret_cast<const acc4_t*>(W + kMlpLdsB3) + (lane >> 4);
  _Float16* x_rows = reinterpret_cast<_Float16*>(lds_wave);                                // [128][16]
  float* a_rows = reinterpret_cast<float*>(lds_wave + kMlpRowsPerWave * kMlpInPad * 2);      // [128][4]
  // 1. observations -> fp16 rows [features | 1 | 0...] in LDS (row = l * 64 + lane)
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    _Float16 row[kMlpInPad];
#pragma unroll
    for (int c = 0; c < kMlpInPad; ++c) {
      const float feature = c < 8 ? obs[l][c] : 0.0f;
      row[c] = static_cast<_Float16>(c < L.obs_dim ? feature : (c == L.obs_dim ? 1.0f : 0.0f));  // [features | 1 | 0 ...]
    }
    half4_t* dst = reinterpret_cast<half4_t*>(x_rows + (l * 64 + lane) * kMlpInPad);
#pragma unroll
    for (int c4 = 0; c4 < kMlpInPad / 4; ++c4) dst[c4] = half4_t{row[4 * c4], row[4 * c4 + 1], row[4 * c4 + 2], row[4 * c4 + 3]};
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // 2. eight tiles of 16 rows through the three layers, activations in registers
#pragma unroll 1
  for (int t = 0; t < kMlpRowsPerWave / 16; ++t) {
    asm volatile("" ::: "memory");  // the fragments are STREAMED: read here, every tile, not hoisted into 68 registers held across the loop
    // B operand of layer 1: k = 4 (lane / 16) + j of row n = lane % 16
    const half4_t x = *reinterpret_cast<const half4_t*>(x_rows + (16 * t + (lane & 15)) * kMlpInPad + 4 * (lane >> 4));
    // Software pipeline over the LDS reads: the fragments of the NEXT group of matrix instructions are requested before the
    // current group is issued (LDS returns in order, so the wait before a group leaves the newer requests in flight), and the
    // scheduler may not sink them back down to their use (sched_barrier): without this every one of the 14 instructions of
    // a tile waited for its own round trip to LDS.
    half4_t f1[kMlpTiles];
#pragma unroll
    for (int mt = 0; mt < kMlpTiles; ++mt) f1[mt] = w1[mt * 64];
    half8_t fa = w2[0], fb = w2[64];
    acc4_t fc = b2[0];
    __builtin_amdgcn_sched_barrier(0);
    acc4_t a1[kMlpTiles], a2[kMlpTiles];
#pragma unroll
    for (int mt = 0; mt < kMlpTiles; ++mt) a1[mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(f1[mt], x, acc4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    const half8_t h1_lo = activate_pair<ACT>(a1[0], a1[1]), h1_hi = activate_pair<ACT>(a1[2], a1[3]);
#pragma unroll
    for (int mt = 0; mt < kMlpTiles; ++mt) {
      // next: the fragments of hidden tile mt + 1, or the output layer's
      const half8_t na = mt + 1 < kMlpTiles ? w2[((mt + 1) * (kMlpTiles / 2) + 0) * 64] : w3[0];
      const half8_t nb = mt + 1 < kMlpTiles ? w2[((mt + 1) * (kMlpTiles / 2) + 1) * 64] : w3[64];
      const acc4_t nc = mt + 1 < kMlpTiles ? b2[4 * (mt + 1)] : b3[0];
      __builtin_amdgcn_sched_barrier(0);
      a2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, h1_lo, fc, 0, 0, 0);
      a2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb, h1_hi, a2[mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      fa = na;
      fb = nb;
      fc = nc;
    }
    const half8_t h2_lo = activate_pair<ACT>(a2[0], a2[1]), h2_hi = activate_pair<ACT>(a2[2], a2[3]);
    acc4_t out = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, h2_lo, fc, 0, 0, 0);
    out = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb, h2_hi, out, 0, 0, 0);
    // lanes 0..15 hold output rows 0..3 (= the action components) of batch row 16 t + lane
    if (lane < 16) *reinterpret_cast<acc4_t*>(a_rows + (16 * t + lane) * 4) = out;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // 3. each thread takes its own two rows back and clips them to the action space
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const acc4_t a = *reinterpret_cast<const acc4_t*>(a_rows + (l * 64 + lane) * 4);
    act[l][0] = a.x;  // the actor's mean: exploration noise and the clip to the action space follow (explore_and_clip)
    act[l][1] = a.y;
    act[l][2] = a.z;
    act[l][3] = a.w;
  }
  __builtin_amdgcn_wave_barrier();  // the rows are reused by the next call
}

// (the activation is a template parameter: chosen once per call, outside the tile loop, instead of per value)
__device__ __forceinline__ void mlp_forward_wave(const char* W, const LearnedPolicyParams& L, const float (&obs)[2][8], float (&act)[2][4],
                                                 char* lds_wave) {
  if (L.activation == kActRelu) mlp_forward_wave_act<kActRelu>(W, L, obs, act, lds_wave);
  else mlp_forward_wave_act<kActTanh>(W, L, obs, act, lds_wave);
}

// mean action = W obs + b: D x A <= 32 FMAs per row in fp32 on the vector unit (no contraction worth a matrix core)
__device__ __forceinline__ void linear_forward(const LearnedPolicyParams& L, const float (&obs)[8], float (&act)[4]) {
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float s = L.w.lin_b[a];
#pragma unroll
    for (int c = 0; c < 8; ++c) s = __builtin_fmaf(L.w.lin_w[a * 8 + c], c < L.obs_dim ? obs[c] : 0.0f, s);
    act[a] = a < L.act_dim ? s : 0.0f;
  }
}

}  // namespace mbt
