// Counter-based Philox4x32-10 (Salmon et al., SC'11) for gfx950, and the noise layout of the step kernel.
//
// The reference draws from three numpy PCG64 generators (StochasticProcessModel.py:27); a device kernel
// cannot reproduce that stream, so production noise is defined here instead - as a pure function of
// (seed, global trajectory id, philox step).  It costs zero bytes of HBM traffic, is restart-safe and is
// independent of how the trajectory axis is sharded over GPUs.
//
// Stream layout (order-book dynamics).  Global lane ids are grouped in tiles of 512; lanes g and g + 256 of a tile
// form PAIR p = (g / 512) * 256 + g % 256 - the two lanes one GPU thread owns (step_kernel.hpp).  A pair consumes
// exactly two blocks, every bit of which is used:
//   block 0: ctr = (p.lo, p.hi, step, 0)  -> lane g       : u_arr_bid, u_arr_ask, u_fill_bid, u_fill_ask
//   block 1: ctr = (p.lo, p.hi, step, 1)  -> lane g + 256 : same four
//            each uniform takes the TOP 24 bits of its word (u = (w >> 8) * 2^-24)
//   the four LOW bytes of block 0 form the 32-bit word of the Box-Muller radius, those of block 1 the angle:
//            z(g) = r cos(theta), z(g + 256) = r sin(theta)
//   key = (seed.lo, seed.hi)
// (The first layout spent a third block on the two normals; the generator is the largest share of the arithmetic of
// every kernel here, and of the fused rollout in particular.)
// (Speed dynamics need one normal per lane: one block per quad of adjacent lanes, counter word 3 = 3, speed_kernel.hpp.)
// Uniforms are u = (w >> 8) * 2^-24 in [0,1): exactly representable in float32, which is what lets the
// Bernoulli decisions of the step be bit-exact against a float64 evaluation of the same draws.
#pragma once
#ifndef __HIPCC_RTC__  // hiprtc (the run-time compiler of user plugins, mbt_env_create_jit) pre-includes the HIP runtime ...
#include <hip/hip_runtime.h>
#include <stdint.h>
#else  // ... and keeps the fixed-width integer types in a namespace of its own
typedef __hip_internal::uint8_t uint8_t;
typedef __hip_internal::uint32_t uint32_t;
typedef __hip_internal::uint64_t uint64_t;
typedef __hip_internal::int32_t int32_t;
#endif

#ifndef MBT_PHILOX_ROUNDS
#define MBT_PHILOX_ROUNDS 10  // the standard Philox4x32-10; fewer rounds are for sensitivity experiments only
#endif

namespace mbt {

struct PhiloxWords {
  uint32_t w0, w1, w2, w3;
};

__device__ __forceinline__ PhiloxWords philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                     uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;  // round multipliers
  constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;  // Weyl key increments
#pragma unroll
  for (int round = 0; round < MBT_PHILOX_ROUNDS; ++round) {
    const uint64_t p0 = static_cast<uint64_t>(M0) * c0;  // one v_mad_u64_u32 yields hi and lo
    const uint64_t p1 = static_cast<uint64_t>(M1) * c2;
    // three-way XOR in ONE instruction: gfx950's v_bitop3_b32 with the parity truth table 0x96 (the compiler emits two
    // v_xor_b32 for `a ^ b ^ c`); 40 VALU fewer per pair of lanes and step - the fused rollout is bound by exactly that
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32(static_cast<uint32_t>(p1 >> 32), c1, k0, 0x96);
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32(static_cast<uint32_t>(p0 >> 32), c3, k1, 0x96);
    c1 = static_cast<uint32_t>(p1);
    c3 = static_cast<uint32_t>(p0);
    c0 = n0;
    c2 = n2;
    k0 += W0;  // keys are wave-uniform: this runs on the scalar unit
    k1 += W1;
  }
  return PhiloxWords{c0, c1, c2, c3};
}

// [0,1) on the 2^-24 grid - exact in float32.
__device__ __forceinline__ float uniform24(uint32_t w) { return static_cast<float>(w >> 8) * 0x1.0p-24f; }

// Box-Muller on the hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (which take
// their argument in revolutions, so the 32-bit word maps straight onto the angle).
__device__ __forceinline__ void box_muller(uint32_t wr, uint32_t wt, float& z_cos, float& z_sin) {
  const float u1 = (static_cast<float>(wr >> 8) + 0.5f) * 0x1.0p-24f;  // (0,1): log never sees 0
  // -2 ln2 log2(u1) lies in [6e-8, 34]: never denormal, so the bare v_sqrt_f32 (1 ulp) is used - the correctly rounded sqrtf is
  // the same instruction plus ~13 more (two Newton corrections and denormal scaling), 6 % of the fused rollout's vector work
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  const float rev = static_cast<float>(wt >> 8) * 0x1.0p-24f;  // angle in revolutions, [0,1)
  z_cos = r * __builtin_amdgcn_cosf(rev);
  z_sin = r * __builtin_amdgcn_sinf(rev);
}

// The byte of each word that uniform24() discards, packed w0 | w1 << 8 | w2 << 16 | w3 << 24 (three v_perm_b32).
__device__ __forceinline__ uint32_t low_bytes(const PhiloxWords& w) {
  const uint32_t lo = __builtin_amdgcn_perm(w.w1, w.w0, 0x0c0c0400u);  // byte 0 <- w0.b0, byte 1 <- w1.b0, rest zero
  const uint32_t hi = __builtin_amdgcn_perm(w.w3, w.w2, 0x04000c0cu);  // byte 2 <- w2.b0, byte 3 <- w3.b0
  return lo | hi;
}

struct LaneNoise {
  float ua_bid, ua_ask, uf_bid, uf_ask, z;
  uint32_t wa_bid, wa_ask;  // the generator's own 32-bit words behind the arrival uniforms (u = (w >> 8) * 2^-24): a Poisson arrival is decided on these
};

// Noise of the pair `pair` (global pair index) at `step`: a = its lower lane, b = the lane 256 above.
__device__ __forceinline__ void philox_pair_noise(uint64_t pair, uint32_t step, uint32_t k0, uint32_t k1,
                                                  LaneNoise& a, LaneNoise& b) {
  const uint32_t plo = static_cast<uint32_t>(pair), phi = static_cast<uint32_t>(pair >> 32);
  const PhiloxWords wa = philox4x32_10(plo, phi, step, 0u, k0, k1);
  const PhiloxWords wb = philox4x32_10(plo, phi, step, 1u, k0, k1);
  a.ua_bid = uniform24(wa.w0); a.ua_ask = uniform24(wa.w1); a.uf_bid = uniform24(wa.w2); a.uf_ask = uniform24(wa.w3);
  b.ua_bid = uniform24(wb.w0); b.ua_ask = uniform24(wb.w1); b.uf_bid = uniform24(wb.w2); b.uf_ask = uniform24(wb.w3);
  a.wa_bid = wa.w0; a.wa_ask = wa.w1; b.wa_bid = wb.w0; b.wa_ask = wb.w1;
  box_muller(low_bytes(wa), low_bytes(wb), a.z, b.z);
}

// Two more standard normals per lane for USER-DEFINED processes that draw noise of their own (a second midprice factor, a
// stochastic intensity): a third block of the pair, ctr = (p.lo, p.hi, step, 2), words (0, 1) -> z1 of the pair's two lanes,
// words (2, 3) -> z2.  Drawn only by the run-time compiled kernels that ask for it (Variant::USER_DRAWS).
__device__ __forceinline__ void philox_pair_user_noise(uint64_t pair, uint32_t step, uint32_t k0, uint32_t k1, float& a_z1, float& a_z2,
                                                       float& b_z1, float& b_z2) {
  const PhiloxWords w = philox4x32_10(static_cast<uint32_t>(pair), static_cast<uint32_t>(pair >> 32), step, 2u, k0, k1);
  box_muller(w.w0, w.w1, a_z1, b_z1);
  box_muller(w.w2, w.w3, a_z2, b_z2);
}

}  // namespace mbt
