// The fused TradingEnvironment.step() kernel for gfx950 (CDNA4, wave64).
//
// One launch = one env.step(action) over all lanes (reference: gym/TradingEnvironment.py:103-110 and the
// ~35 NumPy kernels it fans out to).  One GPU thread owns a PAIR of trajectories, 256 lanes apart inside a 512-lane tile
// (see "lane <-> thread mapping" below), so that every memory instruction of a wave covers one contiguous span:
//   reads   2 state rows [cash, inventory, time, midprice (, bid intensity, ask intensity)]: one float4 each (D = 4)
//           2 action rows: float2 (A=2) or float4 (A=4)
//   draws   2 Philox4x32-10 blocks = all the noise of the pair (philox.hpp), or loads injected noise
//   writes  2 next-state rows, 2 rewards
// The state is row-major (N, D) float32 - exactly the un-normalised observation the API returns.  state_out may BE state_in: a lane
// reads its own row and writes its own row, and rows that leave through LDS are written by threads of the same workgroup behind the
// barrier that follows the last use of the loads - the host updates the state in place from 80 MB moved per launch up (which keeps
// launches like Hawkes + OU at 2^22 lanes inside the Infinity Cache) and steps between two buffers below (mbt_env.hip: mbt_env::state).
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
//     market-order flags are evaluated in double; the fill test compares the depth with a bracket of the threshold
//     -ln(u)/kappa (v_log_f32, computed before the loads are consumed) and re-decides in double only a depth inside
//     the bracket (about 2e-6 of draws).
//   * rewards: float32 but computed from the step's INCREMENTS, never as a difference of two large
//     mark-to-market values: PnL = n_b*d_b + n_a*d_a - h*(mb+ms) + q'*dS (+ clip corrections), which is the
//     reference's (c'+q'S') - (c+qS) (RW:27-33) with the S terms cancelled analytically.
//   * cash / midprice: float32 state, a few ulps from the float64 reference.
#pragma once
#ifndef __HIPCC_RTC__  // (see philox.hpp)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "philox.hpp"
#ifndef MBT_JIT_USER_CODE
#include "policy_mlp.hpp"
#endif

namespace mbt {

#ifndef MBT_BLOCK_THREADS
#define MBT_BLOCK_THREADS 256
#endif
constexpr int kBlockThreads = MBT_BLOCK_THREADS;
constexpr uint32_t kClipSlots = 1024;  // power of two  // wave64 x 4 per workgroup by default

enum : int { kMidBrownian = 0, kMidOu = 1, kMidGbm = 2, kMidBrownianJump = 3, kMidOuJump = 4, kMidConstant = 5, kMidLinearSde = 6, kMidUser = 7 };
enum : int { kArrPoisson = 0, kArrHawkes = 1 };
enum : int { kDynLimit = 0, kDynLimitAndMarket = 1, kDynTouch = 2, kDynSpeed = 3 };
enum : int { kRewPnl = 0, kRewRunning = 1, kRewCjMm = 2, kRewExpUtility = 3, kRewCjOe = 4 };
enum : int { kImpactTempPower = 0, kImpactTempPerm = 1, kImpactTempTransient = 2, kImpactTransient = 3 };

// Compile-time shape of one kernel instantiation: what changes the memory layout or the amount of noise is a
// template parameter; the midprice and reward kinds are wave-uniform runtime switches (a few scalar branches).
enum : int { kRewardPnl = 0, kRewardQuadratic = 1, kRewardGeneral = 2 };
enum : int { kHostFill = 1, kHostArrival = 2, kHostReward = 4, kHostImpact = 8 /* speed kernels: SpeedVariant::HOST_IMPACT; host library only */ };

// A kernel's shape is a LIST OF NAMED TAGS - `Variant<shape::brownian, shape::pnl>` is the Avellaneda-Stoikov kernel of BASELINE
// configs[1], `Variant<shape::hawkes_exact, shape::normalised>` a Hawkes kernel with exact intensities and normalised spaces - not a
// row of sixteen positional booleans: what is not named takes its default, the order of the tags does not matter to what is compiled, an
// unknown tag does not compile, and the name a profiler prints for a kernel says what the kernel is.  (Rounds 1-5:
// `Variant<1, 0, false, 0, false, false, false, false, false, false, false, false, 0, false, 0, true>` - one transposed `false` selected
// another kernel silently.)  kernel_table.hpp holds the ONE mapping from a configuration's values to a shape (OrderBookShape).
namespace shape {
// arrivals (default: the Poisson layout - Poisson, PoissonNonLinear and user arrival expressions share it)
struct hawkes {};        // two intensity columns, float32 (mbt_config::hawkes_float32_intensities)
struct hawkes_exact {};  // ... held exactly: float32 rounding in the row + an int32 remainder beside it (the default Hawkes tier)
// dynamics (default: limit orders)
struct limit_and_market {};
struct touch {};
// midprice (default: any model, through run-time coefficients)
struct brownian {};      // plain Brownian motion: the increment needs nothing from memory
// how heavy the reward is (default: general - other exponents via powf, exponential utility)
struct pnl {};           // plain PnL
struct quadratic {};     // RunningInventoryPenalty / CjMmCriterion with exponent 2 (branch-free)
struct normalised {};    // normalised actions and / or observations (TE:112-126)
struct injected {};      // noise loaded from HBM instead of Philox (parity mode)
struct exogenous {};     // ExogenousMmFillProbabilityModel: two more columns holding the exogenous best depths
struct precise {};       // precise_state: the reference's float64 state, exactly
// run-time compiled plugins (mbt_env_create_jit)
struct user_fill {};
struct user_reward {};
struct user_arrival {};
struct user_mid {};
struct user_draws {};                  // the user's processes consume two more normals per lane and step
template <int N> struct user_state {}; // state columns owned by user processes: 1 or 2
template <int MASK> struct host {};    // host-callback plugins: kHostFill | kHostArrival | kHostReward

template <class A, class B> struct same_tag { static constexpr bool value = false; };
template <class A> struct same_tag<A, A> { static constexpr bool value = true; };
template <class T, class... Tags> constexpr bool has = (same_tag<T, Tags>::value || ... || false);
template <class T> struct columns_of { static constexpr int value = 0; };
template <int N> struct columns_of<user_state<N>> { static constexpr int value = N; };
template <class T> struct host_mask_of { static constexpr int value = 0; };
template <int MASK> struct host_mask_of<host<MASK>> { static constexpr int value = MASK; };
template <class T> constexpr bool known = has<T, hawkes, hawkes_exact, limit_and_market, touch, brownian, pnl, quadratic, normalised, injected, exogenous, precise, user_fill,
                                              user_reward, user_arrival, user_mid, user_draws> || columns_of<T>::value != 0 || host_mask_of<T>::value != 0;

// lists of tags, for code that assembles a shape from values (kernel_table.hpp): make<when<COND, tag>, ...> is the Variant of the tags whose
// condition holds, in the order written
template <class... Tags> struct tags {};
template <bool COND, class T> struct when_impl { using type = tags<>; };
template <class T> struct when_impl<true, T> { using type = tags<T>; };
template <bool COND, class T> using when = typename when_impl<COND, T>::type;
template <class A, class B> struct joined;
template <class... A, class... B> struct joined<tags<A...>, tags<B...>> { using type = tags<A..., B...>; };
template <class... Lists> struct join_all { using type = tags<>; };
template <class First, class... Rest> struct join_all<First, Rest...> { using type = typename joined<First, typename join_all<Rest...>::type>::type; };
}  // namespace shape

template <class... Tags>
struct Variant {
  static_assert((shape::known<Tags> && ... && true), "unknown shape tag");
  template <class T> static constexpr bool has = shape::has<T, Tags...>;
  static constexpr int ARR = (has<shape::hawkes> || has<shape::hawkes_exact>) ? kArrHawkes : kArrPoisson;
  static constexpr int DYN = has<shape::limit_and_market> ? kDynLimitAndMarket : has<shape::touch> ? kDynTouch : kDynLimit;
  static_assert(!(has<shape::hawkes> && has<shape::hawkes_exact>) && !(has<shape::limit_and_market> && has<shape::touch>) && !(has<shape::pnl> && has<shape::quadratic>),
                "contradicting shape tags");
  static constexpr bool BROWNIAN = has<shape::brownian>;  // plain Brownian midprice: the increment needs nothing from memory
  // how heavy the reward is: plain PnL | RunningInventoryPenalty / CjMmCriterion with exponent 2 (branch-free) |
  // everything else (other exponents via powf, exponential utility) - keeps the common kernels free of that code
  static constexpr int REWARD = has<shape::pnl> ? kRewardPnl : has<shape::quadratic> ? kRewardQuadratic : kRewardGeneral;
  static constexpr bool PENALISED = REWARD != kRewardPnl;
  static constexpr bool NORM = has<shape::normalised>;  // normalised actions and/or observations (TE:112-126)
  static constexpr bool INJECT = has<shape::injected>;  // noise loaded from HBM instead of Philox
  // ExogenousMmFillProbabilityModel (FILL:126-170): two more columns holding the exogenous best depths, which the
  // reference never advances (FILL:168-170) - the kernel does not load them, it writes the constants.  Instantiated
  // only on the general tier (no brownian / pnl / quadratic tag, normalised: each a superset of the specialised code).
  static constexpr bool EXO = has<shape::exogenous>;
  // precise_state (mbt_config): every real-valued state column is held EXACTLY as the reference's float64 value - the row
  // holds its float32 rounding (what the observation shows), a side buffer of int32 the rest (exact_join / exact_split) -
  // and the step is evaluated in double in the reference's own order of operations (lane_step_exact): state and rewards
  // are the reference's float64 results, bit for bit, rounded once to float32 on the way out.
  static constexpr bool PRECISE = has<shape::precise>;
  // Exact Hawkes intensities in the float32 tier (mbt_config::hawkes_float32_intensities == 0, the default): ARR:110-123 keeps
  // lambda in float64 and decides `u < lambda dt` on it, so a float32 lambda decides ~1e-6 of lane-steps differently.  Here ONLY
  // the two intensity columns are held exactly (float32 rounding in the row + int32 remainder beside it, exact_join /
  // exact_split), the recursion and the threshold are evaluated in double in the reference's order; cash and midprice stay
  // float32 with the increment-form reward.  +16 B per env-step (76 instead of 60); arrivals, fills and inventory are then the
  // reference's on the same draws in EVERY tier.  Meaningless (and off) under precise, which holds every column exactly.
  static constexpr bool EXACT_LAM = has<shape::hawkes_exact> && !PRECISE;
  // User-defined plugins (mbt_env_create_jit): this header is compiled at RUN TIME (hiprtc) together with the user's
  // device expressions for FillProbabilityModel._get_fill_probabilities (FILL:22-34) and / or RewardFunction.calculate
  // (RW:8-17); general tier only.  Never instantiated in the ahead-of-time library.
  static constexpr bool USER_FILL = has<shape::user_fill>, USER_REWARD = has<shape::user_reward>;
  static constexpr bool USER_MID = has<shape::user_mid>;  // MidpriceModel.update as an expression for S' - S (one column, one normal per step)
  static constexpr bool USER_ARRIVAL = has<shape::user_arrival>;  // a stateless ArrivalModel.get_arrivals (ARR:27-29) as an expression of time
  static_assert(!(USER_ARRIVAL && ARR == kArrHawkes), "a user arrival model replaces the arrival model: Poisson layout (its state columns are user_state)");
  // State columns OWNED by user-defined processes (SP:8-53: a subclass carries its own (N, d) state): 0, 1 or 2 columns
  // x0, x1 right after the midprice, in the reference's registry order (a second midprice factor first, then the arrival
  // model's columns, TE:303-318).  They live where the Hawkes intensities of the built-in model live (`lam`), are advanced
  // by the user's state_update expressions, and may be read by the user's midprice and arrival expressions.  user_draws:
  // the user's processes consume two more standard normals per lane and step (a third Philox block per pair of lanes).
  static constexpr int USER_STATE = (shape::columns_of<Tags>::value + ... + 0);
  static constexpr bool USER_DRAWS = has<shape::user_draws>;
  static_assert(USER_STATE >= 0 && USER_STATE <= 2, "at most two user state columns");
  static_assert(!(USER_STATE != 0 && (ARR == kArrHawkes || EXO)), "user state columns take the place of the Hawkes intensities / exogenous depths");
  // residual columns: precise [cash, midprice (, the two columns after it)]; exact intensities [bid intensity, ask intensity]
  static constexpr int RES = PRECISE ? ((ARR == kArrHawkes || USER_STATE != 0) ? 4 : 2) : (EXACT_LAM ? 2 : 0);
  // HOST-CALLBACK plugins (mbt_env_create_jit, no device expression): a subclass of the reference's plugin contract that only
  // has NumPy code (FILL:22-34 `_get_fill_probabilities`, ARR:27-29 `get_arrivals`, RW:10-13 `calculate`) keeps running on the
  // HOST, between launches; the kernel takes what the host computed for this step instead of evaluating a model -
  //   kHostFill     the (N, 2) fill probabilities of the step's depths (StepBuffers::host_fill_p): fill <=> u < p, in double
  //   kHostArrival  the (N, 2) arrival indicators get_arrivals() returned (StepBuffers::host_arrivals)
  //   kHostReward   the reward is calculate()'d on the host from the float64 states: the kernel reports 0 and the host's
  //                 values are filed afterwards (host_reward_kernel)
  // and everything else of the step (masking, cash / inventory, clip, midprice, Hawkes intensities, normalisation) stays here.
  static constexpr int HOST = (shape::host_mask_of<Tags>::value | ... | 0);
  static constexpr bool HOST_FILL = (HOST & 1) != 0, HOST_ARRIVAL = (HOST & 2) != 0, HOST_REWARD = (HOST & 4) != 0;
  static_assert(!(HOST_FILL && (USER_FILL || EXO)) && !(HOST_ARRIVAL && (USER_ARRIVAL || ARR == kArrHawkes)) && !(HOST_REWARD && USER_REWARD),
                "a plugin is either a built-in, a device expression or a host callback");
  static constexpr int EXTRA = (ARR == kArrHawkes) ? 2 : USER_STATE;  // columns between the midprice and the exogenous depths
  static constexpr int EXO_COL = 4 + EXTRA;
  static constexpr int DIM = EXO_COL + (EXO ? 2 : 0);
};
namespace shape {
template <class List> struct variant_of;
template <class... Tags> struct variant_of<tags<Tags...>> { using type = Variant<Tags...>; };
template <class... Lists> using make = typename variant_of<typename join_all<Lists...>::type>::type;
}  // namespace shape

// Wave-uniform parameters of one step: passed by value (kernarg -> SGPRs).
// What the precise_state tier computes with: the constructor arguments of the reference's classes as float64, kept
// SEPARATE (mu and dt, not mu*dt) wherever the reference multiplies them inside the step - the order of the roundings is
// part of the contract there.
struct PreciseParams {
  int32_t mid_kind, reserved;  // MBT_MID_* (the exact tier follows each model's own expression, not the coefficient form)
  double mu, sigma, mid_dt, sqrt_mid_dt;  // MID:63-64, MID:98-102: drift, volatility, the model's step size and its square root
  double mu_dt, sigma_sqrt_dt;            // mu * dt and sigma * sqrt(dt): the products the reference forms before touching the state
  double mid_add, mid_mul, ou_speed, ou_level, jump_size;
  double hawkes_speed, hawkes_base_bid, hawkes_base_ask, hawkes_jump, arr_dt;  // ARR:110-119
  double half_spread, q_max, c_max;
  double phi, alpha, exponent, risk_aversion, episode_length, reward_scale;
  // trading-with-speed dynamics (MD:262-267, IMP:34-179)
  double temp_coef, impact_exponent, perm_coef, trans_coef, resilience, kernel_coef, impact_dt, speed_dt;
};

// LAYOUT (round 6).  Kernel arguments are read with scalar loads from a segment that is fresh memory at every launch, and a
// launch pays for every distinct 64-byte line of it the kernel must have before its first vector load: the same 40-byte-per-lane
// kernel takes 5.9 us reading one line of a 1.3 KB block, 6.1 reading two to four, 6.45 reading eight
// (tools/microbench/mb_lanes_per_thread.hip).  The benchmark kernel read parameters from five lines of this struct (fields in
// the order five rounds added them); they are now grouped BY WHO READS THEM, the lightest kernels' first, one group per 64 bytes:
// two lines for the benchmark kernel (tools/dbg/kernarg_loads.py lists what a kernel reads, from its disassembly).  In the
// kernels the gain is a fraction of the micro-benchmark's - medians of eight alternating processes (tools/dbg/ab_variants.sh,
// profiles/r06_kernarg_layout.txt): Avellaneda-Stoikov 6.62 -> 6.56 us, CJP 6.79 -> 6.64, Hawkes float32 36.65 -> 36.24,
// precise_state AS 9.86 -> 9.51; limit + market 16.6 -> 16.7 and speed precise_state 13.68 -> 13.81 the other way.  Measured and not taken: the
// struct 64-byte aligned (the segment then is, too: the benchmark kernel +0.1 us), the pointers of StepBuffers regrouped the
// same way (one 64-byte load of eight pointers: the benchmark kernel -0.15 us in some processes and +0.15 in others, CJP +0.1).
// Keep it that way: a new field goes into the group of the kernels that read it, or to the end.
struct StepParams {
  // ---- line 0: every kernel (lanes, generator, clock) + Brownian midprice + Poisson arrival thresholds --------------------
  uint32_t n;            // lanes of this shard
  uint32_t n_pairs;      // padded lanes / 2 (a whole number of 512-lane tiles for order-book dynamics)
  uint64_t pair_offset;  // global pair index of local pair 0
  uint32_t key0, key1;   // Philox key (seed)
  uint32_t philox_step;  // Philox counter word 2
  int32_t is_terminal;   // this step ends the episode (TE:218-220), decided on the host
  float t_next;          // time written into the next state (TE:216)
  float dt;              // terminal_time / n_steps (TE:49): the clock and the reward penalties
  // midprice (each process scales with ITS OWN step size, SP:21; the environment never synchronises them)
  float drift_dt;        // mu * dt_mid                  (MID:63, MID:98)
  float vol_sqrt_dt;     // sigma * sqrt(dt_mid)         (MID:64, MID:100-102, MID:143)
  // Poisson thresholds on the generator's 32-bit word w (u = (w >> 8) * 2^-24):  u < thr  <=>  (w >> 8) < K, K = ceil(thr * 2^24)
  // <=>  w < (K << 8) - no shift, no conversion, no multiply.  K = 2^24 (a probability of one) does not fit: the word is then
  // 0xFFFFFFFF and arr_always_* says that every draw arrives.
  uint32_t arr_thr_w_bid, arr_thr_w_ask;
  int32_t arr_always_bid, arr_always_ask;
  // ---- line 1: fills, clip, reward scale; the quadratic inventory penalties --------------------------------------------------
  float fill_depth_per_log2;  // -ln(2) / kappa: the depth threshold is log2(u) times this
  float fill_band_abs;        // 2e-7 / kappa
  double kappa_f64;
  float q_max, c_max;
  float reward_scale;     // TE:128-129
  float half_spread;
  float alpha_running, alpha_cjmm;  // kRewardQuadratic: alpha routed to the terminal (RW:135-137) or the spread (RW:102-108) term
  float quad_new, quad_init;        // kRewardQuadratic: dt phi + alpha_cjmm, and alpha_cjmm dt / (T - t_start) (set by reset)
  float dt_over_episode;  // CjMm: dt / (T - t_start)    (RW:106)
  float q_init_scalar;    // CjMm: initial inventory when it is the same for every lane
  int32_t reward_kind;
  int32_t exponent_is_two;
  // ---- line 2: the other midprice models, Hawkes arrivals, normalisation flags and the action's affine map -----------------
  float mid_add, mid_mul;  // midprice model as coefficients, see midprice_increment()
  float ou_speed, ou_level;
  float jump_size;       // MID:226, MID:269
  float arr_dt;
  double arr_dt_f64;               // Hawkes: threshold lambda_lane * dt_arr in double (ARR:123)
  float hawkes_base_bid, hawkes_base_ask, hawkes_speed, hawkes_jump;
  // normalisation (TE:112-126); gradients are float32 like the reference's Box bounds
  int32_t norm_act, norm_obs;
  float arr_thr_bid, arr_thr_ask;  // Poisson: smallest float32 >= lambda*dt_arr (ARR:56) or 1-exp(-lambda*dt_arr) (ARR:83) - injected uniforms compare with these
  // ---- line 3: trading-with-speed dynamics and price impact (MD:262-267, IMP:34-179), the general reward ----------------------
  int32_t impact_kind;
  int32_t impact_exponent_is_one;
  float speed_dt;         // the MIDPRICE model's step size (MD:265)
  float impact_dt;        // the impact model's own step size (IMP:75)
  float temp_coef, impact_exponent, perm_coef, trans_coef, resilience, kernel_coef;
  float phi, alpha, exponent;
  float risk_aversion;    // ExponentialUtility (RW:150)
  float episode_length;   // CjOe: T - t_start (RW:73-74)
  int32_t reserved_pad;
  // ---- lines 4-5: the Box bounds of normalised spaces --------------------------------------------------------------------------
  float act_lo[4], act_grad[4];
  float obs_lo[8], obs_grad[8];
  // ---- the exogenous-depth fill model ----------------------------------------------------------------------------------------
  float exo_depth[2], exo_base;          // exogenous best depths (bid, ask) and base fill probability (FILL:159-163)
  float kappa_log2e_neg;  // -kappa * log2(e): p = 2^(kappa_log2e_neg * depth)  (exogenous-depth model)
  double exo_depth_f64[2], exo_base_f64;
  // ---- precise_state: its two clocks, then the reference's constructor arguments in double ---------------------------------
  double t_now;  // the clock BEFORE this step (TE:216 accumulates it in double on the host): what a user arrival model sees
  double t_next_f64;  // ... and after it: the precise_state tier's TIME column and the `dt` of its rewards (RW:99, RW:131)
  double mid_dt_f64;  // the midprice model's own step size (SP:21), for a user midprice expression
  PreciseParams X;
  double user_fill_p[8], user_reward_p[8], user_arrival_p[8], user_mid_p[8], user_state_p[8];  // parameters of the user's device expressions (mbt_user_code)
};

static_assert(__builtin_offsetof(StepParams, fill_depth_per_log2) == 64 && __builtin_offsetof(StepParams, mid_add) == 128 && __builtin_offsetof(StepParams, impact_kind) == 192 &&
                  __builtin_offsetof(StepParams, act_lo) == 256 && __builtin_offsetof(StepParams, exo_depth) == 352,
              "StepParams: one group of readers per 64-byte line (see LAYOUT above)");

// what the user's process expressions may read beyond their own arguments: the user state columns before the step and the
// two extra normals of the step (zero when the configuration has none)
struct UserProcessState {
  double x0, x1, z1, z2;
  // what the state matrix holds when the reference calls a process's update() (TE:206-211): cash, inventory and time after the
  // agent's update and the clip (TE:213-216), the midprice already advanced (first in the registry) - `S_next`, `t_next`,
  // `q_next`, `cash_next` of the state-update expressions; zero (unset) for the arrival / midprice expressions, which run earlier
  double S_next, t_next, q_next, cash_next;
};

#ifdef MBT_JIT_USER_CODE
// Defined by the translation unit mbt_env_create_jit generates in front of this header.
__device__ double mbt_user_fill_probability(double depth, int side, const double* p);
struct UserRewardArgs {
  double cash, q, t, mid;                      // current state (before the step)
  double cash_next, q_next, t_next, mid_next;  // next state
  double a0, a1, a2, a3;                       // the action as the agent gave it
  double pnl;                                  // mark-to-market change (c' + q' S') - (c + q S), from the step's increments
  double dt, is_terminal, q0, episode_length;  // step size, 1.0 on the terminal step, initial inventory, T - t_start
};
__device__ double mbt_user_reward(const UserRewardArgs& s, const double* p);
__device__ double mbt_user_arrival_probability(double t, int side, double dt, const UserProcessState& u, const double* p);
__device__ double mbt_user_midprice_increment(double S, double t, double z, double dt, double fills_bid, double fills_ask, const UserProcessState& u, const double* p);
// the user state columns after the step (which = 0, 1): StochasticProcessModel.update (SP:8-53) of the processes that own them
// (dt_mid / dt_arr: the step sizes of the midprice and of the arrival model - each column is advanced with its owner's, SP:21)
__device__ double mbt_user_state_next(int which, double S, double t, double dt_mid, double dt_arr, double z, double arr_bid, double arr_ask, double fills_bid,
                                      double fills_ask, const UserProcessState& u, const double* p);
#endif

struct StepBuffers {  // (field order: as it grew; regrouping the pointers by reader was measured and not taken - profiles/r06_kernarg_layout.txt)
  const float* state_in;   // (n_pad, D) row-major
  float* state_out;
  const float* action;     // (n_pad, A)
  float* reward;           // (n_pad)
  float* obs;              // normalised observation (n_pad, D) or nullptr
  const float* u_arr;      // injected noise (n_pad, 2), (n_pad, 2), (n_pad)
  const float* u_fill;
  const float* z;
  const float* z_user;     // injected noise of user processes (n_pad, 2), or nullptr
  const float* q_init;     // CjMm per-lane initial inventory or nullptr
  int32_t* resid;          // precise_state: (n_pad, RES) int32 remainders of the float64 state (exact_join), updated in place; else nullptr
  uint8_t* events;         // nullptr unless recording
  float* lane_returns;     // nullptr unless tracking
  double* wave_sums;       // one slot per wave: running sum of rewards since reset
  unsigned long long* clip_count;  // kClipSlots counters, indexed by workgroup: a step in which every lane clips must not
                                   // serialise 8192 atomics on one address (17 -> 144 us at 2^21 lanes before the split)
  // Small batches over the host API (mbt_env_step_host, the reference's own regime of N ~ 1000): the step kernel ITSELF
  // mirrors what the API returns - observation rows (n, D) and rewards (n), no pad rows - into pinned, device-mapped host
  // memory (posted PCIe writes) and the last workgroup to finish raises a flag there that the host spins on: one launch and
  // no interrupt per env.step() (round 3: a second launch for the export and a hipStreamSynchronize).  All nullptr otherwise.
  float* host_obs;
  float* host_reward;
  uint32_t* done_counter;  // device memory: workgroups of this launch that have finished (reset by the last one)
  uint32_t* host_flag;     // device-mapped host memory: receives flag_value once every workgroup's mirror stores are visible
  uint32_t flag_value;
  uint32_t reserved_pad;
  // host-callback plugins (Variant::HOST): what the host computed for this step
  const double* host_fill_p;   // (n_pad, 2) fill probabilities of this step's depths; speed dynamics with a host-callback impact model: (n_pad) price impacts
  const float* host_arrivals;  // (n_pad, 2) arrivals as 0.0f / 1.0f
  // ... and, for small batches, what the host needs back to call the user's update() / calculate() without a further round trip
  // (round 5; the MIRROR instantiation of a Variant::HOST kernel writes them into device-mapped host memory like the observation):
  float* host_state;           // (n, D) the un-normalised state rows (float32 roundings), or nullptr
  int32_t* host_resid;         // (n, RES) their int32 remainders (precise_state / exact intensities), or nullptr
  uint8_t* host_events;        // (n) the step's event bytes (arrivals, fills, market orders, clips), or nullptr
};

// ---- structure of the arithmetic --------------------------------------------------------------------------------
// (1) Everything that depends on noise and parameters only is precomputed into a LaneDraw while the loads are in
//     flight; (2) the post-load arithmetic uses explicit FMAs / v_med3 and carries nothing optional (event bytes are
//     assembled only when recording); (3) rare exact re-decisions live in cold blocks.  Explicit FMAs (the library is
//     built with -ffp-contract=off) also make the step and rollout kernels, which inline the same code, agree bit
//     for bit.  Measured (rocprofv3 PMC, 2^20 lanes, profiles/r01_pmc_sq.txt): 296 VALU instructions per wave; a wave is
//     resident for ~2500 quad-cycles of which 304 issue VALU - times the 8 resident waves of a SIMD that is most of the
//     residency window, so at this size arithmetic and the memory system co-limit the kernel.

// Per-lane quantities that depend on noise and parameters only.
struct LaneDraw {
  float arr_bid, arr_ask;  // Poisson: arrival indicators 1.0f / 0.0f (ARR:56); Hawkes: the raw uniforms (decided with lambda)
  float uf_bid, uf_ask;    // fill uniforms (FILL:33); only the exact re-decision and the exogenous model read them after the loads
  float lo_bid, hi_bid, lo_ask, hi_ask;  // depth thresholds of the exponential fill test, see fill_thresholds()
  float dz;                // drift_dt + vol_sqrt_dt * Z: the whole midprice increment of Brownian motion (MID:60-65)
};

__device__ __forceinline__ void fill_thresholds(float u, const StepParams& P, float& lo, float& hi);

template <class V>
__device__ __forceinline__ LaneDraw make_draw(const LaneNoise& nz, const StepParams& P) {
  LaneDraw d;
  if (V::ARR == kArrPoisson && !V::USER_ARRIVAL && !V::INJECT) {
    // Philox uniforms are u = (w >> 8) * 2^-24: the decision u < thr is an integer compare on the word w itself (StepParams) -
    // no shift, no conversion, no multiply (the float uniforms of the arrival side are then dead code in this instantiation)
    d.arr_bid = ((nz.wa_bid < P.arr_thr_w_bid) | (P.arr_always_bid != 0)) ? 1.0f : 0.0f;
    d.arr_ask = ((nz.wa_ask < P.arr_thr_w_ask) | (P.arr_always_ask != 0)) ? 1.0f : 0.0f;
  } else if (V::ARR == kArrPoisson && !V::USER_ARRIVAL) {  // strict '<' against thresholds rounded UP to float32: exact vs the float64 compare
    d.arr_bid = nz.ua_bid < P.arr_thr_bid ? 1.0f : 0.0f;
    d.arr_ask = nz.ua_ask < P.arr_thr_ask ? 1.0f : 0.0f;
  } else {
    d.arr_bid = nz.ua_bid;
    d.arr_ask = nz.ua_ask;
  }
  d.uf_bid = nz.uf_bid;
  d.uf_ask = nz.uf_ask;
  d.lo_bid = d.hi_bid = d.lo_ask = d.hi_ask = 0.0f;
  if (!V::EXO && !V::USER_FILL && !V::HOST_FILL && V::DYN != kDynTouch) {
    fill_thresholds(nz.uf_bid, P, d.lo_bid, d.hi_bid);
    fill_thresholds(nz.uf_ask, P, d.lo_ask, d.hi_ask);
  }
  d.dz = __builtin_fmaf(P.vol_sqrt_dt, nz.z, P.drift_dt);
  return d;
}

// ---- fill decisions --------------------------------------------------------------------------------------
// Exponential fill test (FILL:34, FILL:57-58), exact against float64:  u < exp(-kappa * depth)  <=>  depth < t(u),
// t(u) = -ln(u) / kappa.  The threshold depends on the draw only, so it is computed BEFORE the state/action loads are
// consumed (v_log_f32, one multiply) and the decision after the loads is two compares: every wave of a SIMD leaves
// the load wait at about the same time, so arithmetic after it is serialised across the resident waves - moving the
// exponential in front of the wait took the AS step from 8.37 to 8.16 us (2^20 lanes).  [lo, hi] brackets the exact
// threshold: relative 1e-6 (v_log_f32 is good to 1e-7 relative away from u ~ 1 - checked exhaustively over all 2^24
// uniforms, tests/test_gpu_rng.py - plus the rounding of a de-normalised depth) and absolute 2e-7 / kappa (u ~ 1).
// A depth inside the bracket (probability ~2e-6) is re-decided in double by the cold block.  u = 0 gives t = +inf and
// lo = NaN: every depth counts as inside the bracket and the double evaluation decides (exp underflow, FILL:58).
struct FillTest {
  bool fill;  // depth < t(u), certain unless `near`
  bool near;  // the depth lies inside the bracket: the fast decision cannot be trusted
};

__device__ __forceinline__ void fill_thresholds(float u, const StepParams& P, float& lo, float& hi) {
  const float t = __builtin_amdgcn_logf(u) * P.fill_depth_per_log2;  // -ln(u) / kappa  (v_log_f32 is log2)
  const float band = __builtin_fmaf(t, 1e-6f, P.fill_band_abs);  // t >= 0 for u in [0, 1]
  lo = t - band;
  hi = t + band;
}

__device__ __forceinline__ FillTest fill_test(float depth, float lo, float hi) {
  const bool sure = depth < lo;
  return FillTest{sure, !sure && !(depth > hi)};
}

// ExogenousMmFillProbabilityModel (FILL:159-163): p = 1 for a quote at or inside the exogenous best depth,
// base * exp(-kappa (depth - best)) beyond it.  `near` additionally covers a quote within rounding of the best depth.
__device__ __forceinline__ FillTest fill_test_exogenous(float u, float depth, int side, const StepParams& P) {
  const float best = P.exo_depth[side];
  const float x = depth - best;
  const float y = P.kappa_log2e_neg * x;
  const float p = P.exo_base * __builtin_amdgcn_exp2f(y);
  const float slack = (__builtin_fabsf(depth) + __builtin_fabsf(best)) * 2.4e-7f;  // roundings of depth, best and x
  const float band = __builtin_fmaf(p, __builtin_fmaf(__builtin_fabsf(y), 3e-7f, __builtin_fmaf(__builtin_fabsf(P.kappa_log2e_neg), slack, 4e-6f)), 1e-30f);
  const bool inside = x <= 0.0f;  // p = 1 exactly: the strict `u < 1` of FILL:34 holds for every uniform in [0, 1)
  const float d = u - (inside ? 1.0f : p);
  return FillTest{d < 0.0f, (!inside && __builtin_fabsf(d) <= band) || __builtin_fabsf(x) <= slack};
}

__device__ __forceinline__ float depth_of(float a, int side, bool norm, const StepParams& P) {
  return norm ? static_cast<float>((static_cast<double>(a) + 1.0) * P.act_grad[side] + P.act_lo[side]) : a;  // TE:124
}

// cold: exact re-decision of one quote (about 4e-6 of draws get here)
__device__ __attribute__((cold)) bool refine_fill_f64(float u, float a, int side, bool norm, const StepParams& P) {
  double depth = a;
  if (norm) depth = (static_cast<double>(a) + 1.0) * P.act_grad[side] + P.act_lo[side];
  return static_cast<double>(u) < exp(-P.kappa_f64 * depth);
}

__device__ __attribute__((cold)) bool refine_fill_exogenous_f64(float u, float a, int side, bool norm, const StepParams& P) {
  double depth = a;
  if (norm) depth = (static_cast<double>(a) + 1.0) * P.act_grad[side] + P.act_lo[side];
  const double best = P.exo_depth_f64[side];
  return static_cast<double>(u) < (depth > best ? P.exo_base_f64 * exp(-P.kappa_f64 * (depth - best)) : 1.0);
}

// ---- x ** p for the float32 tier (IMP:55-56 `action ** exponent`; RW:59-68, :101-104, :133-137 `inventory ** exponent`) ------------
// The library's powf is ~185 vector instructions a call, and the speed kernels call it for four lanes per thread: 1009 instructions per
// wave against 271 without, the vector pipe 84 % busy, 0.56 of the HBM line (profiles/r06_speed_precise_pmc.json) - while a bare
// exp2(p * log2(x)) on v_log_f32 / v_exp_f32 loses p * log2(x) * 2^-24 of relative accuracy (1e-6 at x = 100: the tier's whole reward
// tolerance).  This form spends the chip's float64 rate instead (half the float32 rate on CDNA4) and is accurate to 0.5001 ulp - the
// float64 power NumPy computes, rounded to float32 once (tools/microbench/pow_f32_model.py: 16 million arguments against long double):
//   |x| = 2^k m, m in [sqrt(1/2), sqrt(2));  l0 = v_log_f32(m), good to ~2^-23;  d = m * 2^(-l0) - 1 in double (2^t: a degree-8
//   polynomial, 1.5e-12 on |t| <= 0.52), so  log2(m) = l0 + d / ln 2  to ~1e-12 (d^2 < 1e-13);  y = p (k + log2 m);
//   x^p = 2^rint(y) * 2^(y - rint(y)), the same polynomial, v_ldexp_f64, one conversion.  About 30 float64 + 10 float32 instructions.
// Sign and zero as C's pow / NumPy's power have them (a negative base with an integral exponent keeps or loses its sign by the
// exponent's parity, with any other exponent it is NaN; 0 ** p is 0, 1 or inf); subnormal, infinite and NaN bases and p == 0 go to the
// library's pow out of line (pow_out_of_line below) - no configuration's actions or inventories are any of these.
static __device__ __attribute__((noinline, cold)) double pow_out_of_line(double q, double p);
__device__ __forceinline__ double exp2_on_half_unit(double t) {  // 2^t, |t| <= 0.52: Chebyshev interpolant of degree 8, relative error 1.5e-12
  double r = 0x1.63e65bcaf94f0p-20;
  r = __builtin_fma(r, t, 0x1.00f06b86a7b42p-16);
  r = __builtin_fma(r, t, 0x1.43089c4c30d66p-13);
  r = __builtin_fma(r, t, 0x1.5d87263a9a124p-10);
  r = __builtin_fma(r, t, 0x1.3b2ab71ee8c18p-7);
  r = __builtin_fma(r, t, 0x1.c6b08df2412e4p-5);
  r = __builtin_fma(r, t, 0x1.ebfbdff821419p-3);
  r = __builtin_fma(r, t, 0x1.62e42fef7972cp-1);
  return __builtin_fma(r, t, 1.0);
}
__device__ __forceinline__ float power_f32(float x, double p) {
  const float ax = __builtin_fabsf(x);
  if (__builtin_expect(!(ax < __builtin_inff()) || (ax != 0.0f && ax < 1.17549435e-38f) || p == 0.0, 0))
    return static_cast<float>(pow_out_of_line(static_cast<double>(x), p));
  float m = __builtin_amdgcn_frexp_mantf(ax);  // [1/2, 1)
  int k = __builtin_amdgcn_frexp_expf(ax);
  if (m < 0.70710678f) {
    m *= 2.0f;
    k -= 1;
  }
  const float l0 = __builtin_amdgcn_logf(m);  // log2, |l0| <= 1/2
  const double d = __builtin_fma(static_cast<double>(m), exp2_on_half_unit(-static_cast<double>(l0)), -1.0);
  const double log2_m = __builtin_fma(d, 1.4426950408889634, static_cast<double>(l0));
  double y = p * (static_cast<double>(k) + log2_m);
  y = __builtin_fmin(__builtin_fmax(y, -2000.0), 2000.0);  // (far beyond float32 either way: ldexp saturates to 0 / inf)
  const double whole = __builtin_rint(y);
#ifndef MBT_EXP_POW_VEXP
  const double magnitude = __builtin_ldexp(exp2_on_half_unit(y - whole), static_cast<int>(whole));
#else  // EXPERIMENT (tools/dbg/r06_pow_variants.sh): the last stage on v_exp_f32 / v_ldexp_f32 - nine float64 instructions fewer per lane; measured 7.83
       // instead of 8.18 us for the speed kernel, but 0.84 ulp and one result in sixteen not the correctly rounded one: the half ulp was kept
  const float magnitude = __builtin_ldexpf(__builtin_amdgcn_exp2f(static_cast<float>(y - whole)), static_cast<int>(whole));
#endif
  float r = ax != 0.0f ? static_cast<float>(magnitude) : (p > 0.0 ? 0.0f : __builtin_inff());
  if (__builtin_signbitf(x)) {  // (wave-uniform tests of the exponent; a lane-wise select of the result)
    const bool integral = p == __builtin_rint(p);
    const bool odd = integral && __builtin_fabs(p) < 0x1p53 && __builtin_rint(p * 0.5) * 2.0 != p;
    r = integral ? (odd ? -r : r) : (ax != 0.0f ? __builtin_nanf("") : r);
  }
  return r;
}

// numpy `q ** p` (RW:101-104, RW:133-137) of up to three inventories at once; p == 2 in every reference
// configuration.  The general case runs ONE inlined power_f32 in a rolled loop over register selects (no arrays, so no
// scratch memory), which keeps it out of the instruction stream of the common path, and only for the inventories the reward
// reads (`need`: bit 0 a, bit 1 b, bit 2 c - wave-uniform: a running penalty reads one of them, not three); b is raised to `p_b`
// (CjOe: the old inventory to p - 1, RW:65).
__device__ __forceinline__ void inventory_powers(float a, float b, float c, double p, double p_b, bool is_two, uint32_t need, float& pa, float& pb, float& pc) {
  if (__builtin_expect(is_two, 1)) {
    pa = a * a; pb = p_b == 2.0 ? b * b : b; pc = c * c;
    return;
  }
  pa = pb = pc = 0.0f;
#pragma unroll 1
  for (int i = 0; i < 3; ++i) {
    if (((need >> i) & 1u) == 0u) continue;
    const float y = power_f32(i == 0 ? a : (i == 1 ? b : c), i == 1 ? p_b : p);  // (as a call, out of line: 8.97 -> 9.93 us for the exponent-1.5 case, -1 % elsewhere)
    if (i == 0) pa = y; else if (i == 1) pb = y; else pc = y;
  }
}

// S' - S for the midprice models other than plain Brownian motion, branch-free: the host turns the model kind into
// coefficients of
//   dS = (mid_add + mid_mul * S) * dz - ou_speed * (S - ou_level) + jump_size * (n_ask - n_bid),   dz = mu dt + sigma sqrt(dt) Z
//   GBM (MID:95-103)  mid_mul 1        OU (MID:140-143)  mid_add 1, mu 0, ou_speed theta (the pull is NOT scaled by dt)
//   +jumps on the agent's own trades (MID:222-227, :264-270)        constant (MID:32-33)  all 0
__device__ __forceinline__ float midprice_increment(float mid, float dz, float n_bid, float n_ask, const StepParams& P) {
  const float scale = __builtin_fmaf(P.mid_mul, mid, P.mid_add);
  const float pull = P.ou_speed * (mid - P.ou_level);
  return __builtin_fmaf(P.jump_size, n_ask - n_bid, __builtin_fmaf(scale, dz, -pull));
}

// Everything a reward adds to the mark-to-market change `pnl` (RW:96-109, RW:128-138, RW:57-70, RW:156-163).
__device__ __forceinline__ float finish_reward(float pnl, float q_old, float q_new, float cash_new, float mid_new, float q_init,
                                               float speed, bool is_terminal, const StepParams& P) {
  float reward = pnl;
  if (P.reward_kind == kRewExpUtility) {  // exponential utility of terminal wealth; zero before the terminal step
    reward = is_terminal ? -__expf(-P.risk_aversion * (cash_new + q_new * mid_new)) : 0.0f;
  } else if (P.reward_kind != kRewPnl) {
    float qp, qp_old, qp_init;
    const bool oe = P.reward_kind == kRewCjOe;  // needs q^(p-1) instead of q^p for the old inventory (RW:65)
    // RW:133-137 reads q' ** p alone, RW:101-108 all three, RW:59-68 q' and q0 to p and q to p - 1
    inventory_powers(q_new, q_old, q_init, P.X.exponent, oe ? P.X.exponent - 1.0 : P.X.exponent, P.exponent_is_two != 0, P.reward_kind == kRewRunning ? 1u : 7u, qp, qp_old, qp_init);
    reward -= P.dt * P.phi * qp;
    if (P.reward_kind == kRewRunning) {
      reward -= is_terminal ? P.alpha * qp : 0.0f;
    } else if (!oe) {
      reward -= P.alpha * ((qp - qp_old) + P.dt_over_episode * qp_init);
    } else {  // the terminal term MULTIPLIES by the episode length in the reference (RW:67)
      reward -= P.dt * P.alpha * (P.exponent * speed * qp_old + qp_init * P.episode_length);  // (qp_old: q ** (p - 1) here)
    }
  }
  return reward * P.reward_scale;
}

// The same for the rewards every reference configuration uses - inventory exponent 2, no exponential utility - without a
// single transcendental: what the speed kernels inline four times per thread when the host knows the exponents
// (SpeedVariant::POWERS = false).  Rounds exactly like finish_reward's exponent_is_two paths.
__device__ __forceinline__ float finish_reward_squares(float pnl, float q_old, float q_new, float q_init, float speed, bool is_terminal,
                                                       const StepParams& P) {
  float reward = pnl;
  if (P.reward_kind != kRewPnl) {
    const float qp = q_new * q_new, qp_old = q_old * q_old, qp_init = q_init * q_init;
    reward -= P.dt * P.phi * qp;
    if (P.reward_kind == kRewRunning) {
      reward -= is_terminal ? P.alpha * qp : 0.0f;
    } else if (P.reward_kind != kRewCjOe) {
      reward -= P.alpha * ((qp - qp_old) + P.dt_over_episode * qp_init);
    } else {
      reward -= P.dt * P.alpha * (P.exponent * speed * q_old + qp_init * P.episode_length);
    }
  }
  return reward * P.reward_scale;
}

// ---- precise_state: the reference's float64 state, exactly, in 8 bytes per value ---------------------------------------
// A double x is kept as  hi = float32(x)  (round to nearest: what the observation shows, np.float32(x)) in the state row
// and  lo = (x - hi) * 2^(53 - e)  as an int32 in a side buffer, e = the exponent of hi.  x - hi is exact in double, at most
// half a float32 ulp (2^(e-24)) in magnitude and a multiple of 2^(e-53) (the last bit of a double in hi's binade or the one
// below it - the latter when hi rounded up to a power of two), so lo is an INTEGER of at most 2^29: the pair holds every
// double whose float32 rounding is a normal number, exactly (zero, denormal or non-finite hi: lo = 0, |x| < 2^-126 is lost).
// A float32 pair (hi, float32(x - hi)) would keep 48 of the 53 bits; this keeps all of them for the same bytes, which is
// what makes "the same draws give the same arrivals" a theorem for Hawkes intensities instead of a 1 - 1e-14 event.
__device__ __forceinline__ int f32_biased_exponent(float hi) { return static_cast<int>((__builtin_bit_cast(uint32_t, hi) >> 23) & 0xffu); }
__device__ __forceinline__ double exact_join(float hi, int32_t lo) {
  return static_cast<double>(hi) + __builtin_ldexp(static_cast<double>(lo), f32_biased_exponent(hi) - (127 + 53));
}
__device__ __forceinline__ void exact_split(double x, float& hi, int32_t& lo) {
  hi = static_cast<float>(x);
  const int e = f32_biased_exponent(hi);
  const double scaled = __builtin_ldexp(x - static_cast<double>(hi), (127 + 53) - e);
  lo = (e != 0 && e != 255) ? static_cast<int32_t>(scaled) : 0;
}

// numpy's `q ** p` for the exponents it special-cases (2: a multiplication, 1: the value itself) and pow() otherwise
__device__ __forceinline__ double numpy_power(double q, double p) { return p == 2.0 ? q * q : (p == 1.0 ? q : pow(q, p)); }
// The same with the library functions OUT OF LINE: one body of pow() / exp() per kernel instead of one per call site.  For kernels
// that advance several lanes per thread in double (the speed family's precise_state tier: four lanes, up to four pow() each - 47-59 KB
// of code and 131-133 registers inlined, round 5) and take these paths only for exponents no reference configuration uses.  Same
// code, same flags (-ffp-contract=off): the same bits as the inlined call.
static __device__ __attribute__((noinline, cold)) double pow_out_of_line(double q, double p) { return pow(q, p); }  // (declared above power_f32)
static __device__ __attribute__((noinline, cold)) double exp_out_of_line(double x) { return exp(x); }
__device__ __forceinline__ double numpy_power_out_of_line(double q, double p) { return p == 2.0 ? q * q : (p == 1.0 ? q : pow_out_of_line(q, p)); }

// One Euler step of the midprice in double, each model in the operation order of ITS update() (MID:60-65, :95-103,
// :140-143, :222-227, :264-270; MBT_MID_LINEAR_SDE: the order LinearSdeMidpriceModel documents).  n_bid / n_ask: 1.0 where
// the agent's quote was filled (MID:220-221).  The kind is wave-uniform.
__device__ __forceinline__ double midprice_step_exact(double s, double z, double n_bid, double n_ask, const PreciseParams& X) {
  const double noise = X.sigma_sqrt_dt * z;                           // (sigma * sqrt(dt)) * Z
  const double jump = X.jump_size * n_ask - X.jump_size * n_bid;      // MID:226, MID:269
  switch (X.mid_kind) {
    case kMidBrownian: return (s + X.mu_dt) + noise;
    case kMidOu: return s + (-X.ou_speed * (s - X.ou_level) + noise);  // the pull is NOT scaled by dt (MID:140-143)
    case kMidGbm: return (s + X.mu * s * X.mid_dt) + X.sigma * s * X.sqrt_mid_dt * z;
    case kMidBrownianJump: return ((s + X.mu_dt) + noise) + jump;
    case kMidOuJump: return ((s - X.ou_speed * (s - X.ou_level)) + noise) + jump;
    case kMidLinearSde: return ((s + (X.mid_add + X.mid_mul * s) * (X.mu_dt + noise)) - X.ou_speed * (s - X.ou_level)) + jump;
    default: return s;  // constant (MID:32-33)
  }
}

// `q ** p` where the HOST already knows that p is 1 or 2 (every reference configuration: mbt_env.hip: reward_weight /
// speed_powers pick the instantiation): numpy's two fast paths without pow()'s ~300 inlined instructions and its registers
__device__ __forceinline__ double numpy_power_1_or_2(double q, double p) { return p == 2.0 ? q * q : q; }

// RewardFunction.calculate in double, in the reference's order (RW:23-33, RW:96-109, RW:128-138, RW:57-70, RW:156-163).
// dt is the difference of the two TIME columns (RW:99, RW:131), i.e. of the accumulated float64 clock, not step_size.
// TIER (the float32 kernels' reward tiers): kRewardPnl - the mark-to-market change alone; kRewardQuadratic - the penalised rewards
// with inventory exponents of 1 or 2 and no exponential utility (no pow, no exp in the instruction stream); kRewardGeneral -
// everything.  The operations and their order are the SAME in every tier: which one runs changes no bit of the result.
// OUT_OF_LINE: pow() / exp() as calls (numpy_power_out_of_line), for kernels that inline this function several times.
template <int TIER = kRewardGeneral, bool OUT_OF_LINE = false>
__device__ __forceinline__ double reward_exact(double cash, double q, double mid, double cash_new, double q_new, double mid_new, double q_init,
                                               double speed, bool is_terminal, double t_now, double t_next, const StepParams& P) {
  const PreciseParams& X = P.X;
  const double wealth_new = cash_new + q_new * mid_new;
  const double pnl = wealth_new - (cash + q * mid);
  if (TIER == kRewardPnl) return X.reward_scale * pnl;
  const auto power = [](double base, double p) {
    return TIER == kRewardQuadratic ? numpy_power_1_or_2(base, p) : (OUT_OF_LINE ? numpy_power_out_of_line(base, p) : numpy_power(base, p));
  };
  double reward = pnl;
  if (TIER == kRewardGeneral && P.reward_kind == kRewExpUtility) {
    reward = is_terminal ? -(OUT_OF_LINE ? exp_out_of_line(-X.risk_aversion * wealth_new) : exp(-X.risk_aversion * wealth_new)) : 0.0;
  } else if (P.reward_kind != kRewPnl) {
    const double dt = t_next - t_now;
    const double qp = power(q_new, X.exponent);
    if (P.reward_kind == kRewRunning) {
      reward = (pnl - dt * X.phi * qp) - X.alpha * (is_terminal ? 1.0 : 0.0) * qp;
    } else if (P.reward_kind == kRewCjMm) {
      reward = (pnl - dt * X.phi * qp) - X.alpha * ((qp - power(q, X.exponent)) + dt / X.episode_length * power(q_init, X.exponent));
    } else {  // CjOe: the terminal term MULTIPLIES by the episode length (RW:67)
      reward = (pnl - dt * X.phi * qp) - dt * X.alpha * (X.exponent * speed * power(q, X.exponent - 1.0) + power(q_init, X.exponent) * X.episode_length);
    }
  }
  return X.reward_scale * reward;  // TE:128-129 (1.0 when rewards are not normalised: exact)
}

struct LaneResult {
  float4 core;
  float2 lam;
  int4 lo;  // precise_state: what float32 left of [cash, midprice, bid intensity, ask intensity] (exact_split)
  float reward;
  // what happened, kept as predicates (scalar masks); turned into the event byte only when someone asks for it
  bool arr_bid, arr_ask, fill_bid, fill_ask, mo_buy, mo_sell, clipped_q, clipped_c;
};

__device__ __forceinline__ uint32_t event_byte(const LaneResult& r) {
  return (r.arr_bid ? 1u : 0u) | (r.arr_ask ? 2u : 0u) | (r.fill_bid ? 4u : 0u) | (r.fill_ask ? 8u : 0u) | (r.mo_buy ? 16u : 0u) |
         (r.mo_sell ? 32u : 0u) | (r.clipped_q ? 64u : 0u) | (r.clipped_c ? 128u : 0u);
}

// what a host-callback plugin computed for this lane and step (Variant::HOST; zero otherwise)
struct HostStep {
  float arr_bid = 0.f, arr_ask = 0.f;  // ArrivalModel.get_arrivals() of the user's class (ARR:27-29), as 0 / 1
  double p_bid = 0.0, p_ask = 0.0;     // FillProbabilityModel._get_fill_probabilities(depths) of the user's class (FILL:22-34)
};

// The Bernoulli decisions of one lane-step - arrivals, fills after the max-inventory mask, market-order flags - which are
// the same (bit-exact against the float64 reference) in every tier; the tiers differ in how they carry the state.
struct Decisions {
  float arr_bid, arr_ask;  // 1.0f / 0.0f (ARR:56, ARR:83, ARR:123)
  float n_bid, n_ask;      // executed size on each side: arrival x fill (x posted size at the touch)
  float off_bid, off_ask;  // distance of the execution price from the midprice: the (de-normalised) depth, or the half spread
  bool fill_bid, fill_ask, mo_buy, mo_sell;
};

// lam_bid / lam_ask: the Hawkes intensities the comparison of ARR:123 is made with, as doubles (the float32 tiers pass their
// float32 state, the exact tier the reference's float64 value).
template <class V>
__device__ __forceinline__ Decisions decide(const float q, const float4 act, const LaneDraw& dr, const double lam_bid, const double lam_ask,
                                            const double t_now, const bool norm_act, const StepParams& P, const UserProcessState& ups,
                                            const HostStep& hs = HostStep{}) {
  Decisions D;
  // -- arrivals (ARR:54-56 / ARR:121-123), strict '<'
  float arr_bid = dr.arr_bid, arr_ask = dr.arr_ask;
  if (V::ARR == kArrHawkes) {
    arr_bid = static_cast<double>(dr.arr_bid) < lam_bid * P.arr_dt_f64 ? 1.0f : 0.0f;
    arr_ask = static_cast<double>(dr.arr_ask) < lam_ask * P.arr_dt_f64 ? 1.0f : 0.0f;
  }
  if (V::USER_ARRIVAL) {
#ifdef MBT_JIT_USER_CODE
    // the user's get_arrivals (ARR:27-29): u < p(t, side), in double, at the time stamp of the observation acted on
    arr_bid = static_cast<double>(dr.arr_bid) < mbt_user_arrival_probability(t_now, 0, P.arr_dt_f64, ups, P.user_arrival_p) ? 1.0f : 0.0f;
    arr_ask = static_cast<double>(dr.arr_ask) < mbt_user_arrival_probability(t_now, 1, P.arr_dt_f64, ups, P.user_arrival_p) ? 1.0f : 0.0f;
#endif
  }
  if (V::HOST_ARRIVAL) {  // the user's get_arrivals() ran on the host (its own generator, ARR:27-29): these ARE the arrivals
    arr_bid = hs.arr_bid;
    arr_ask = hs.arr_ask;
  }
  D.arr_bid = arr_bid;
  D.arr_ask = arr_ask;

  // -- fills masked by the PRE-update inventory (TE:323-327).  Limit orders: the exponential fill test
  //    (FILL:28-34, FILL:57-58, after the de-normalisation of TE:104); at the touch: the action itself (MD:156-157)
  const bool open_bid = !(q >= P.q_max), open_ask = !(q <= -P.q_max);
  if (V::DYN == kDynTouch) {
    const float f_bid = open_bid ? act.x : 0.0f, f_ask = open_ask ? act.y : 0.0f;  // `fills` is the posted size (0/1)
    D.fill_bid = f_bid != 0.0f;
    D.fill_ask = f_ask != 0.0f;
    D.n_bid = arr_bid * f_bid;
    D.n_ask = arr_ask * f_ask;
    D.off_bid = D.off_ask = P.half_spread;
  } else {
    D.off_bid = depth_of(act.x, 0, norm_act, P);
    D.off_ask = depth_of(act.y, 1, norm_act, P);
    bool fb = false, fa = false;
    if (V::HOST_FILL) {  // FILL:33-34 with the probabilities the user's _get_fill_probabilities returned for these depths
      fb = static_cast<double>(dr.uf_bid) < hs.p_bid;
      fa = static_cast<double>(dr.uf_ask) < hs.p_ask;
    } else if (V::USER_FILL) {
#ifdef MBT_JIT_USER_CODE
      // the user's _get_fill_probabilities (FILL:22-34), evaluated in double on the de-normalised depth: u < p(depth)
      const double depth_b = norm_act ? (static_cast<double>(act.x) + 1.0) * P.act_grad[0] + P.act_lo[0] : static_cast<double>(act.x);
      const double depth_a = norm_act ? (static_cast<double>(act.y) + 1.0) * P.act_grad[1] + P.act_lo[1] : static_cast<double>(act.y);
      fb = static_cast<double>(dr.uf_bid) < mbt_user_fill_probability(depth_b, 0, P.user_fill_p);
      fa = static_cast<double>(dr.uf_ask) < mbt_user_fill_probability(depth_a, 1, P.user_fill_p);
#endif
    } else {
      const FillTest tb = V::EXO ? fill_test_exogenous(dr.uf_bid, D.off_bid, 0, P) : fill_test(D.off_bid, dr.lo_bid, dr.hi_bid);
      const FillTest ta = V::EXO ? fill_test_exogenous(dr.uf_ask, D.off_ask, 1, P) : fill_test(D.off_ask, dr.lo_ask, dr.hi_ask);
      fb = tb.fill;
      fa = ta.fill;
      if (__builtin_expect(tb.near | ta.near, 0)) {
        if (V::EXO) {
          if (tb.near) fb = refine_fill_exogenous_f64(dr.uf_bid, act.x, 0, norm_act, P);
          if (ta.near) fa = refine_fill_exogenous_f64(dr.uf_ask, act.y, 1, norm_act, P);
        } else {
          if (tb.near) fb = refine_fill_f64(dr.uf_bid, act.x, 0, norm_act, P);
          if (ta.near) fa = refine_fill_f64(dr.uf_ask, act.y, 1, norm_act, P);
        }
      }
    }
    D.fill_bid = fb && open_bid;
    D.fill_ask = fa && open_ask;
    D.n_bid = D.fill_bid ? arr_bid : 0.0f;
    D.n_ask = D.fill_ask ? arr_ask : 0.0f;
  }
  D.mo_buy = D.mo_sell = false;
  if (V::DYN == kDynLimitAndMarket) {  // MD:209-210: scores above 0.5 are market orders
    if (norm_act) {
      D.mo_buy = (static_cast<double>(act.z) + 1.0) * P.act_grad[2] + P.act_lo[2] > 0.5;
      D.mo_sell = (static_cast<double>(act.w) + 1.0) * P.act_grad[3] + P.act_lo[3] > 0.5;
    } else {
      D.mo_buy = act.z > 0.5f;
      D.mo_sell = act.w > 0.5f;
    }
  }
  return D;
}

// One env-step of one lane: everything that needs the loaded state and action.
template <class V>
__device__ __forceinline__ LaneResult lane_step(const float4 core, const float2 lam, const float4 act, const LaneDraw& dr,
                                                const float q_init, const float t_next, const bool is_terminal,
                                                const StepParams& P, const float z = 0.f, const double t_now = 0.0, const float2 zu = make_float2(0.f, 0.f),
                                                const HostStep& hs = HostStep{}, const double t_next_f64 = 0.0,  // (t_next_f64: the advanced clock in double, for user state-update expressions)
                                                const int4 lo = make_int4(0, 0, 0, 0)) {  // (EXACT_LAM: z / w = what float32 left of the two intensities)
  const float cash = core.x, q = core.y, mid = core.w;
  LaneResult r;
  r.lo = make_int4(0, 0, 0, 0);
  const bool norm_act = V::NORM && P.norm_act;
  const UserProcessState ups{V::USER_STATE > 0 ? static_cast<double>(lam.x) : 0.0, V::USER_STATE > 1 ? static_cast<double>(lam.y) : 0.0, zu.x, zu.y};
  // the intensities ARR:123 compares with: the reference's float64 values (EXACT_LAM) or the float32 state
  const double lam_bid = V::EXACT_LAM ? exact_join(lam.x, lo.z) : static_cast<double>(lam.x), lam_ask = V::EXACT_LAM ? exact_join(lam.y, lo.w) : static_cast<double>(lam.y);
  const Decisions D = decide<V>(q, act, dr, lam_bid, lam_ask, t_now, norm_act, P, ups, hs);
  const float arr_bid = D.arr_bid, arr_ask = D.arr_ask, n_bid = D.n_bid, n_ask = D.n_ask;
  r.arr_bid = arr_bid != 0.0f;
  r.arr_ask = arr_ask != 0.0f;
  r.fill_bid = D.fill_bid;
  r.fill_ask = D.fill_ask;
  r.mo_buy = D.mo_buy;
  r.mo_sell = D.mo_sell;

  // -- cash / inventory with the OLD midprice (MD:82-84); market orders (MD:208-214) and limit fills (MD:108-116 /
  //    MD:215-222) or fills at the touch (MD:146-154) together: buying dq units in total costs dq * mid, and every
  //    trade earns its distance from the midprice (`gain`: + depth for a limit fill, - half spread for a market order)
  float dq = n_bid - n_ask;
  float gain = __builtin_fmaf(n_ask, D.off_ask, n_bid * D.off_bid);
  if (V::DYN == kDynLimitAndMarket) {
    const float mb = r.mo_buy ? 1.0f : 0.0f, ms = r.mo_sell ? 1.0f : 0.0f;
    dq += mb - ms;
    gain = __builtin_fmaf(-P.half_spread, mb + ms, gain);
  }
  const float q_new = q + dq;
  const float cash_new = __builtin_fmaf(-dq, mid, cash + gain);

  // -- clip (TE:283-289): v_med3_f32
  const float q_clip = __builtin_amdgcn_fmed3f(q_new, -P.q_max, P.q_max);
  const float c_clip = __builtin_amdgcn_fmed3f(cash_new, -P.c_max, P.c_max);
  const float dq_clip = q_clip - q_new;  // 0 unless the inventory clip fired
  const float dc_clip = c_clip - cash_new;
  r.clipped_q = dq_clip != 0.0f;
  r.clipped_c = dc_clip != 0.0f;

  // -- midprice, then the Hawkes intensities, which jump on arrivals, not on fills (ARR:110-119)
  float d_mid = V::BROWNIAN ? dr.dz : midprice_increment(mid, dr.dz, n_bid, n_ask, P);
  if (V::USER_MID) {
#ifdef MBT_JIT_USER_CODE
    d_mid = static_cast<float>(mbt_user_midprice_increment(mid, t_now, z, P.mid_dt_f64, n_bid, n_ask, ups, P.user_mid_p));
#endif
  }
  const float mid_new = mid + d_mid;
  r.lam = lam;
  if (V::EXACT_LAM) {  // ARR:110-119 in double, in the reference's order (= lane_step_exact's), split back into row + remainder
    const PreciseParams& X = P.X;
    const double lb = (lam_bid + X.hawkes_speed * (X.hawkes_base_bid - lam_bid) * X.arr_dt) + X.hawkes_jump * static_cast<double>(arr_bid);
    const double la = (lam_ask + X.hawkes_speed * (X.hawkes_base_ask - lam_ask) * X.arr_dt) + X.hawkes_jump * static_cast<double>(arr_ask);
    exact_split(lb, r.lam.x, r.lo.z);
    exact_split(la, r.lam.y, r.lo.w);
  } else if (V::ARR == kArrHawkes) {
    r.lam.x = __builtin_fmaf(P.hawkes_jump, arr_bid, lam.x + P.hawkes_speed * (P.hawkes_base_bid - lam.x) * P.arr_dt);
    r.lam.y = __builtin_fmaf(P.hawkes_jump, arr_ask, lam.y + P.hawkes_speed * (P.hawkes_base_ask - lam.y) * P.arr_dt);
  }
  if (V::USER_STATE > 0) {
#ifdef MBT_JIT_USER_CODE
    // the user processes' own update() (SP:8-53), each from the state BEFORE the step, in double; the columns are float32
    UserProcessState upn = ups;
    upn.S_next = mid_new, upn.t_next = t_next_f64, upn.q_next = q_clip, upn.cash_next = c_clip;
    r.lam.x = static_cast<float>(mbt_user_state_next(0, mid, t_now, P.mid_dt_f64, P.arr_dt_f64, z, arr_bid, arr_ask, n_bid, n_ask, upn, P.user_state_p));
    if (V::USER_STATE > 1) r.lam.y = static_cast<float>(mbt_user_state_next(1, mid, t_now, P.mid_dt_f64, P.arr_dt_f64, z, arr_bid, arr_ask, n_bid, n_ask, upn, P.user_state_p));
#endif
  }

  // -- reward: the mark-to-market change (c'+q'S') - (c+qS) of RW:27-33 from the step's increments, then the reward
  //    function's own terms
  const float pnl = __builtin_fmaf(dq_clip, mid, __builtin_fmaf(q_clip, d_mid, gain)) + dc_clip;
  if (V::HOST_REWARD) {
    r.reward = 0.0f;  // calculate() runs on the host (host_reward_kernel files its values)
  } else if (V::REWARD == kRewardPnl) {
    r.reward = pnl * P.reward_scale;
  } else if (V::REWARD == kRewardQuadratic) {
    // RunningInventoryPenalty (RW:128-138) and CjMmCriterion (RW:96-109) with exponent 2, branch-free: the host
    // routes alpha into exactly one of the two terms
    //   penalty = (dt phi + alpha_cjmm [+ alpha_running at the terminal step]) q'^2 - alpha_cjmm q^2 + alpha_cjmm dt/L q0^2
    // with the coefficients folded on the host (wave-uniform; the q0 term needs no loaded value when q0 is a scalar),
    // which leaves four dependent VALU operations after the loads
    const float c_new = P.quad_new + (is_terminal ? P.alpha_running : 0.0f);
    const float pen = __builtin_fmaf(-P.alpha_cjmm, q * q, __builtin_fmaf(c_new, q_clip * q_clip, P.quad_init * (q_init * q_init)));
    r.reward = (pnl - pen) * P.reward_scale;
  } else if (V::USER_REWARD) {
#ifdef MBT_JIT_USER_CODE
    UserRewardArgs u;  // RewardFunction.calculate(current_state, action, next_state, is_terminal_step) (RW:10-13), in double
    u.cash = cash; u.q = q; u.t = static_cast<double>(t_next) - static_cast<double>(P.dt); u.mid = mid;
    u.cash_next = c_clip; u.q_next = q_clip; u.t_next = t_next; u.mid_next = mid_new;
    u.a0 = act.x; u.a1 = act.y; u.a2 = act.z; u.a3 = act.w;
    u.pnl = pnl;
    u.dt = P.dt; u.is_terminal = is_terminal ? 1.0 : 0.0; u.q0 = q_init; u.episode_length = P.episode_length;
    r.reward = static_cast<float>(mbt_user_reward(u, P.user_reward_p) * static_cast<double>(P.reward_scale));
#endif
  } else {
    r.reward = finish_reward(pnl, q, q_clip, c_clip, mid_new, q_init, 0.0f, is_terminal, P);
  }
  r.core = make_float4(c_clip, q_clip, t_next, mid_new);
  return r;
}

// precise_state: the same lane-step on the reference's float64 state, in the reference's float64 arithmetic.  Every
// expression below is written in the operation order of the NumPy statement it cites (the library is compiled with
// -ffp-contract=off, IEEE double add / multiply / divide round like NumPy's), so cash, inventory, midprice, intensities and
// rewards ARE the reference's values; the row gets their float32 rounding, `lo` what the rounding left (exact_split).
// `lo` in: x cash, y midprice, z / w bid / ask intensity.  t_now / t_next: the float64 clock before / after the step.
template <class V>
__device__ __forceinline__ LaneResult lane_step_exact(const float4 core, const float2 lam, const int4 lo, const float4 act, const LaneDraw& dr,
                                                      const float q_init, const bool is_terminal, const StepParams& P, const float z,
                                                      const double t_now, const double t_next, const float2 zu = make_float2(0.f, 0.f),
                                                      const HostStep& hs = HostStep{}) {
  const PreciseParams& X = P.X;
  const double cash = exact_join(core.x, lo.x), mid = exact_join(core.w, lo.y), q = core.y;  // (order-book inventories are integers)
  const double lam_bid = V::EXTRA > 0 ? exact_join(lam.x, lo.z) : 0.0, lam_ask = V::EXTRA > 1 ? exact_join(lam.y, lo.w) : 0.0;  // (or the user state columns)
  LaneResult r;
  const bool norm_act = V::NORM && P.norm_act;
  const UserProcessState ups{V::USER_STATE > 0 ? lam_bid : 0.0, V::USER_STATE > 1 ? lam_ask : 0.0, zu.x, zu.y};
  const Decisions D = decide<V>(core.y, act, dr, lam_bid, lam_ask, t_now, norm_act, P, ups, hs);
  r.arr_bid = D.arr_bid != 0.0f;
  r.arr_ask = D.arr_ask != 0.0f;
  r.fill_bid = D.fill_bid;
  r.fill_ask = D.fill_ask;
  r.mo_buy = D.mo_buy;
  r.mo_sell = D.mo_sell;
  const double n_bid = D.n_bid, n_ask = D.n_ask;
  // the prices limit orders execute at: midprice -/+ depth, the depth de-normalised in double (TE:124) - the float32
  // action IS the reference's float64 action
  double depth_b = act.x, depth_a = act.y;
  if (V::DYN == kDynTouch) {
    depth_b = depth_a = X.half_spread;  // MD:146-154
  } else if (norm_act) {
    depth_b = (static_cast<double>(act.x) + 1.0) * P.act_grad[0] + P.act_lo[0];
    depth_a = (static_cast<double>(act.y) + 1.0) * P.act_grad[1] + P.act_lo[1];
  }
  double cash_new = cash, q_new = q;
  if (V::DYN == kDynLimitAndMarket) {  // MD:208-214: market orders first, at the touch of the OLD midprice
    const double mb = r.mo_buy ? 1.0 : 0.0, ms = r.mo_sell ? 1.0 : 0.0;
    cash_new = cash_new + (ms * (mid - X.half_spread) - mb * (mid + X.half_spread));
    q_new = q_new + (mb - ms);
  }
  // MD:108-116 / MD:146-154 / MD:215-222:  q += sum(arrivals fills -sign),  cash += sum(sign arrivals fills (mid + depth sign)),  sign = (-1, +1)
  q_new = q_new + (n_bid + -n_ask);
  cash_new = cash_new + (-n_bid * (mid + -depth_b) + n_ask * (mid + depth_a));
  // clip (TE:283-289)
  const double q_clip = fmin(fmax(q_new, -X.q_max), X.q_max), c_clip = fmin(fmax(cash_new, -X.c_max), X.c_max);
  r.clipped_q = q_clip != q_new;
  r.clipped_c = c_clip != cash_new;
  // processes in registry order (TE:206-211): midprice, then the intensities, which jump on arrivals (ARR:110-119)
  double mid_new;
  if (V::USER_MID) {
#ifdef MBT_JIT_USER_CODE
    mid_new = mid + mbt_user_midprice_increment(mid, t_now, z, P.mid_dt_f64, n_bid, n_ask, ups, P.user_mid_p);
#else
    mid_new = mid;
#endif
  } else if (V::BROWNIAN) {
    mid_new = (mid + X.mu_dt) + X.sigma_sqrt_dt * static_cast<double>(z);  // MID:60-65, = midprice_step_exact's kMidBrownian row
  } else {
    mid_new = midprice_step_exact(mid, z, n_bid, n_ask, X);
  }
  double lam_bid_new = 0.0, lam_ask_new = 0.0;
  if (V::ARR == kArrHawkes) {
    lam_bid_new = (lam_bid + X.hawkes_speed * (X.hawkes_base_bid - lam_bid) * X.arr_dt) + X.hawkes_jump * static_cast<double>(D.arr_bid);
    lam_ask_new = (lam_ask + X.hawkes_speed * (X.hawkes_base_ask - lam_ask) * X.arr_dt) + X.hawkes_jump * static_cast<double>(D.arr_ask);
  }
  if (V::USER_STATE > 0) {
#ifdef MBT_JIT_USER_CODE
    UserProcessState upn = ups;
    upn.S_next = mid_new, upn.t_next = t_next, upn.q_next = q_clip, upn.cash_next = c_clip;
    lam_bid_new = mbt_user_state_next(0, mid, t_now, P.mid_dt_f64, P.arr_dt_f64, z, D.arr_bid, D.arr_ask, n_bid, n_ask, upn, P.user_state_p);
    if (V::USER_STATE > 1) lam_ask_new = mbt_user_state_next(1, mid, t_now, P.mid_dt_f64, P.arr_dt_f64, z, D.arr_bid, D.arr_ask, n_bid, n_ask, upn, P.user_state_p);
#endif
  }
  // reward
  if (V::HOST_REWARD) {
    r.reward = 0.0f;  // calculate() runs on the host, on these very float64 states (mbt_env_get_state_f64_host)
  } else if (V::USER_REWARD) {
#ifdef MBT_JIT_USER_CODE
    UserRewardArgs u;  // RewardFunction.calculate(current_state, action, next_state, is_terminal_step) (RW:10-13) on the float64 states
    u.cash = cash; u.q = q; u.t = t_now; u.mid = mid;
    u.cash_next = c_clip; u.q_next = q_clip; u.t_next = t_next; u.mid_next = mid_new;
    u.a0 = act.x; u.a1 = act.y; u.a2 = act.z; u.a3 = act.w;
    u.pnl = (c_clip + q_clip * mid_new) - (cash + q * mid);
    u.dt = t_next - t_now; u.is_terminal = is_terminal ? 1.0 : 0.0; u.q0 = q_init; u.episode_length = X.episode_length;
    r.reward = static_cast<float>(X.reward_scale * mbt_user_reward(u, P.user_reward_p));
#else
    r.reward = 0.0f;
#endif
  } else {
    r.reward = static_cast<float>(reward_exact<V::REWARD>(cash, q, mid, c_clip, q_clip, mid_new, q_init, 0.0, is_terminal, t_now, t_next, P));
  }
  float c_hi, m_hi, lb_hi = 0.0f, la_hi = 0.0f;
  r.lo = make_int4(0, 0, 0, 0);
  exact_split(c_clip, c_hi, r.lo.x);
  exact_split(mid_new, m_hi, r.lo.y);
  if (V::EXTRA > 0) exact_split(lam_bid_new, lb_hi, r.lo.z);
  if (V::EXTRA > 1) exact_split(lam_ask_new, la_hi, r.lo.w);
  r.core = make_float4(c_hi, static_cast<float>(q_clip), static_cast<float>(t_next), m_hi);
  r.lam = make_float2(lb_hi, la_hi);
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

// (x - lo) / grad - 1 (TE:112-118), float32 division like the reference's float arithmetic on Box bounds
__device__ __forceinline__ float normalise_column(float x, int col, const StepParams& P) { return (x - P.obs_lo[col]) / P.obs_grad[col] - 1.0f; }

__device__ __forceinline__ void normalise_row(float4& core, float2& lam, const StepParams& P) {
  if (!P.norm_obs) return;
  core.x = normalise_column(core.x, 0, P);
  core.y = normalise_column(core.y, 1, P);
  core.z = normalise_column(core.z, 2, P);
  core.w = normalise_column(core.w, 3, P);
  lam.x = normalise_column(lam.x, 4, P);  // (dead code unless the row has Hawkes columns)
  lam.y = normalise_column(lam.y, 5, P);
}

// ---- lane <-> thread mapping --------------------------------------------------------------------------------------
// A workgroup of 256 threads owns a TILE of 512 consecutive lanes; thread j owns lanes tile*512 + j and tile*512 + 256 + j.
// Consecutive threads touch consecutive rows, so every wave-level load/store covers one contiguous span (a state row is
// one 16-byte vector per lane for D = 4).  Measured with a copy kernel moving the same 44 B/lane at 2^20 lanes
// (tools/microbench/mb_copy.hip): 7.2 us for this mapping against 8.3 us when a thread owns two ADJACENT rows - there
// each 16-byte-per-lane instruction strides 32 bytes and requests every cache line twice.  The two lanes of a thread
// form the PAIR that shares two Philox blocks (philox.hpp); buffers are padded to whole tiles, so no load is ever
// out of bounds and only the reductions need to know which lanes are real.
constexpr uint32_t kTileLanes = 2 * kBlockThreads;

struct LaneLoads {
  float4 core;  // [cash, inventory, time, midprice]
  float2 lam;   // Hawkes intensities
  float4 act;   // (bid depth, ask depth[, market buy, market sell])
  float2 ua, uf;  // injected noise
  float z;
  float2 zu;    // injected noise of user processes (z1, z2)
  float qi;     // per-lane initial inventory (CjMm)
  int4 lo;      // precise_state: the int32 remainders of [cash, midprice, bid intensity, ask intensity] (exact_join)
  HostStep hs;  // host-callback plugins: what the host computed for this lane
};

typedef float ld4_t __attribute__((ext_vector_type(4)));
typedef float ld2_t __attribute__((ext_vector_type(2)));
typedef int ldi4_t __attribute__((ext_vector_type(4)));
typedef int ldi2_t __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ float4 load4(const float* p) {
  const ld4_t v = NT ? __builtin_nontemporal_load(reinterpret_cast<const ld4_t*>(p)) : *reinterpret_cast<const ld4_t*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}
template <bool NT>
__device__ __forceinline__ float2 load2(const float* p) {
  const ld2_t v = NT ? __builtin_nontemporal_load(reinterpret_cast<const ld2_t*>(p)) : *reinterpret_cast<const ld2_t*>(p);
  return make_float2(v.x, v.y);
}

template <class V, bool NT = false>
__device__ __forceinline__ LaneLoads load_lane(const StepBuffers& B, const StepParams& P, uint32_t lane) {
  LaneLoads L;
  if (V::DIM == 8) {  // Hawkes + exogenous depths: 32-byte rows, the (constant) depth columns are not read
    L.core = load4<NT>(B.state_in + static_cast<size_t>(lane) * 8);
    L.lam = load2<NT>(B.state_in + static_cast<size_t>(lane) * 8 + 4);
  } else if (V::DIM == 6) {  // rows of 6 floats: 8-byte aligned
    const float* row = B.state_in + static_cast<size_t>(lane) * 6;
    const float2 a = load2<NT>(row), b = load2<NT>(row + 2);
    L.core = make_float4(a.x, a.y, b.x, b.y);
    L.lam = V::EXTRA == 2 ? load2<NT>(row + 4) : make_float2(0.f, 0.f);
  } else if (V::DIM == 5) {  // one user state column: rows of 20 bytes, 4-byte aligned (run-time compiled kernels only)
    const float* row = B.state_in + static_cast<size_t>(lane) * 5;
    L.core = make_float4(row[0], row[1], row[2], row[3]);
    L.lam = make_float2(row[4], 0.f);
  } else {
    L.core = load4<NT>(B.state_in + static_cast<size_t>(lane) * 4);
    L.lam = make_float2(0.f, 0.f);
  }
  if (V::DYN == kDynLimitAndMarket) {
    L.act = load4<NT>(B.action + static_cast<size_t>(lane) * 4);
  } else {
    const float2 a = load2<NT>(B.action + static_cast<size_t>(lane) * 2);
    L.act = make_float4(a.x, a.y, 0.f, 0.f);
  }
  if (V::INJECT) {
    L.ua = reinterpret_cast<const float2*>(B.u_arr)[lane];
    L.uf = reinterpret_cast<const float2*>(B.u_fill)[lane];
    L.z = B.z[lane];
  }
  L.zu = (V::INJECT && V::USER_DRAWS) ? reinterpret_cast<const float2*>(B.z_user)[lane] : make_float2(0.f, 0.f);
  L.qi = P.q_init_scalar;
  L.lo = make_int4(0, 0, 0, 0);
  if (V::RES == 4) {
    const ldi4_t* src = reinterpret_cast<const ldi4_t*>(B.resid + static_cast<size_t>(lane) * 4);
    const ldi4_t v = NT ? __builtin_nontemporal_load(src) : *src;
    L.lo = make_int4(v.x, v.y, v.z, v.w);
  } else if (V::RES == 2) {
    const ldi2_t* src = reinterpret_cast<const ldi2_t*>(B.resid + static_cast<size_t>(lane) * 2);
    const ldi2_t v = NT ? __builtin_nontemporal_load(src) : *src;
    L.lo = V::EXACT_LAM ? make_int4(0, 0, v.x, v.y) : make_int4(v.x, v.y, 0, 0);  // (z / w are the intensities' slots in either tier)
  }
  if (V::HOST_FILL) {
    const double* p = B.host_fill_p + static_cast<size_t>(lane) * 2;
    L.hs.p_bid = p[0];
    L.hs.p_ask = p[1];
  }
  if (V::HOST_ARRIVAL) {
    const float2 a = reinterpret_cast<const float2*>(B.host_arrivals)[lane];
    L.hs.arr_bid = a.x;
    L.hs.arr_ask = a.y;
  }
  return L;
}

// Per-lane initial inventories (CjMm with random initial inventories only).  Issued AFTER the state/action loads of
// both lanes: the pointer test is a scalar branch behind an s_waitcnt, and must not sit between those loads.
template <class V>
__device__ __forceinline__ void load_initial_inventories(const StepBuffers& B, uint32_t lane0, uint32_t lane1, float& qi0, float& qi1) {
  if (V::PENALISED && B.q_init != nullptr) {
    qi0 = B.q_init[lane0];
    qi1 = B.q_init[lane1];
  }
}

// Orders the schedule: every operand is an in/out of one empty asm, so the draws are complete before, and every
// consumer of the loaded state/action after, this point.  Costs no instruction.
__device__ __forceinline__ void tie_loads_to_draws(LaneLoads& a, LaneLoads& b, LaneDraw& da, LaneDraw& db) {
  asm volatile("; loads are first consumed below this line"
               : "+v"(a.core.x), "+v"(a.core.y), "+v"(a.core.z), "+v"(a.core.w), "+v"(b.core.x), "+v"(b.core.y), "+v"(b.core.z), "+v"(b.core.w),
                 "+v"(a.act.x), "+v"(a.act.y), "+v"(b.act.x), "+v"(b.act.y), "+v"(da.arr_bid), "+v"(da.arr_ask), "+v"(da.uf_bid), "+v"(da.uf_ask),
                 "+v"(da.dz), "+v"(db.arr_bid), "+v"(db.arr_ask), "+v"(db.uf_bid), "+v"(db.uf_ask), "+v"(db.dz), "+v"(da.lo_bid),
                 "+v"(da.hi_bid), "+v"(da.lo_ask), "+v"(da.hi_ask), "+v"(db.lo_bid), "+v"(db.hi_bid), "+v"(db.lo_ask), "+v"(db.hi_ask));
}

// ---- stores that write THROUGH the XCD's L2 -------------------------------------------------------------------------
// A step leaves ~21 MB of new state and rewards at 2^20 lanes.  With ordinary stores those lines sit dirty in the eight
// XCD-private L2s until the end-of-kernel release writes them back - serial time between two dependent launches.  With the
// system-scope bit (sc1) the data streams out to the Infinity Cache / HBM while the kernel is still running, and the next
// step (which reads it through a freshly invalidated L2 anyway) starts sooner: 6.76 -> 5.82 us for the copy kernel of the
// same traffic (tools/microbench/mb_copy.hip), the non-temporal bit (nt) instead costs 18 %.  The builtins offer no
// per-store scope, hence the inline assembly; stores have no consumer in the kernel, so the compiler's wait counters are
// unaffected.  One hazard the compiler handles for its own stores and cannot see through the asm: gfx9 reads the data
// registers of a store wider than 8 bytes over more than one cycle, so a VALU write to them needs a wait state after the
// store - without the s_nop the next address computation overwrote (cash, inventory) of rows in flight.  (The leading
// s_nop 0 keeps a transcendental result that might feed the store one wait state away, as the compiler would.)
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_through(float4* p, const float4 v) {
  const v4f_t t = {v.x, v.y, v.z, v.w};
  asm volatile("s_nop 0\n\tglobal_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void store_through(float2* p, const float2 v) {
  const v2f_t t = {v.x, v.y};
  asm volatile("s_nop 0\n\tglobal_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" : : "v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void store_through(float* p, const float v) {
  asm volatile("s_nop 0\n\tglobal_store_dword %0, %1, off sc1\n\ts_nop 0" : : "v"(p), "v"(v) : "memory");
}

// How a row leaves the kernel.  kStoreThrough: written through the L2 (what the NEXT launch reads: the step kernels' state).
// kStorePlain: ordinary write-back stores (the host mirror; rows that are not 16 bytes wide).  kStoreRecord: the policy of the
// fused rollout's trajectory recording - a stream nobody in the kernel reads back, 28 B per lane and step from one long-running
// kernel (MBT_RECORD_STORE_POLICY, measured in tools/microbench/mb_floor.hip `record` and mb_rollout.hip: 0 plain, 1 sc1, 2 nt).
#ifndef MBT_RECORD_STORE_POLICY
#define MBT_RECORD_STORE_POLICY 0
#endif
enum : int { kStorePlain = 0, kStoreThrough = 1, kStoreStream = 2, kStoreSystem = 3, kStoreThroughStream = 4, kStoreRecord = MBT_RECORD_STORE_POLICY };
// (inline assembly like store_through - the builtins offer neither a scope nor sc1 + nt - with the same gfx9 wait states around it)
#define MBT_STORE_ASM(WIDTH, BITS, NOP_AFTER) asm volatile("s_nop 0\n\tglobal_store_" WIDTH " %0, %1, off " BITS "\n\ts_nop " NOP_AFTER : : "v"(p), "v"(t) : "memory")
template <int STORE>
__device__ __forceinline__ void store_as(float4* p, const float4 v) {
  const v4f_t t = {v.x, v.y, v.z, v.w};
  if (STORE == kStoreThrough) MBT_STORE_ASM("dwordx4", "sc1", "1");
  else if (STORE == kStoreStream) MBT_STORE_ASM("dwordx4", "nt", "1");
  else if (STORE == kStoreSystem) MBT_STORE_ASM("dwordx4", "sc0 sc1", "1");
  else if (STORE == kStoreThroughStream) MBT_STORE_ASM("dwordx4", "sc1 nt", "1");
  else *p = v;
}
template <int STORE>
__device__ __forceinline__ void store_as(float2* p, const float2 v) {
  const v2f_t t = {v.x, v.y};
  if (STORE == kStoreThrough) MBT_STORE_ASM("dwordx2", "sc1", "0");
  else if (STORE == kStoreStream) MBT_STORE_ASM("dwordx2", "nt", "0");
  else if (STORE == kStoreSystem) MBT_STORE_ASM("dwordx2", "sc0 sc1", "0");
  else if (STORE == kStoreThroughStream) MBT_STORE_ASM("dwordx2", "sc1 nt", "0");
  else *p = v;
}
template <int STORE>
__device__ __forceinline__ void store_as(float* p, const float t) {
  if (STORE == kStoreThrough) MBT_STORE_ASM("dword", "sc1", "0");
  else if (STORE == kStoreStream) MBT_STORE_ASM("dword", "nt", "0");
  else if (STORE == kStoreSystem) MBT_STORE_ASM("dword", "sc0 sc1", "0");
  else if (STORE == kStoreThroughStream) MBT_STORE_ASM("dword", "sc1 nt", "0");
  else *p = t;
}

// one state row as given
// STORE: kStoreThrough for the state the NEXT launch reads; see above for the others (bool arguments of earlier rounds map onto
// kStorePlain = false / kStoreThrough = true)
template <class V, int STORE = kStoreThrough>
__device__ __forceinline__ void store_row_values(float* base, uint32_t lane, const float4 core, const float2 lam, const float2 best) {
  if (V::DIM == 8) {
    float4* row = reinterpret_cast<float4*>(base) + static_cast<size_t>(lane) * 2;
    row[0] = core;  // (rows wider than 16 bytes are written by several instructions that each cover PART of a cache line:
    row[1] = make_float4(lam.x, lam.y, best.x, best.y);  // written through, partial lines double the step time - 38 -> 80 us
                                                          // for Hawkes at 2^22 lanes - so these stay ordinary write-back stores)
  } else if (V::DIM == 6) {
    float2* row = reinterpret_cast<float2*>(base) + static_cast<size_t>(lane) * 3;
    row[0] = make_float2(core.x, core.y);
    row[1] = make_float2(core.z, core.w);
    row[2] = V::EXTRA == 2 ? lam : best;
  } else if (V::DIM == 5) {
    float* row = base + static_cast<size_t>(lane) * 5;
    row[0] = core.x; row[1] = core.y; row[2] = core.z; row[3] = core.w; row[4] = lam.x;
  } else {
    store_as<STORE>(reinterpret_cast<float4*>(base) + lane, core);
  }
}

// one state row (un-normalised, or normalised per TE:112-118 when `normalise`)
template <class V, int STORE = kStoreThrough>
__device__ __forceinline__ void store_row(float* base, uint32_t lane, float4 core, float2 lam, bool normalise, const StepParams& P) {
  if (normalise) normalise_row(core, lam, P);
  float2 best = make_float2(P.exo_depth[0], P.exo_depth[1]);  // the exogenous best depths never move (FILL:168-170)
  if (V::EXO && normalise && P.norm_obs) best = make_float2(normalise_column(best.x, V::EXO_COL, P), normalise_column(best.y, V::EXO_COL + 1, P));
  store_row_values<V, STORE>(base, lane, core, lam, best);
}

// precise_state: the normalised observation (TE:112-118) is formed from the float64 state like the reference forms it -
// (x - low) / gradient - 1 in double, the float32 Box bounds promoted - and rounded once.
__device__ __forceinline__ float normalise_column_exact(double x, int col, const StepParams& P) {
  return static_cast<float>((x - static_cast<double>(P.obs_lo[col])) / static_cast<double>(P.obs_grad[col]) - 1.0);
}

// the observation row of the precise_state tier: normalised (when the environment normalises) from the float64 state,
// `t`: the float64 clock the row's time column stands for
template <class V, int STORE = kStoreThrough>
__device__ __forceinline__ void store_row_exact(float* base, uint32_t lane, float4 core, float2 lam, const int4 lo, const double t, const StepParams& P) {
  float2 best = make_float2(P.exo_depth[0], P.exo_depth[1]);
  if (P.norm_obs) {
    core.x = normalise_column_exact(exact_join(core.x, lo.x), 0, P);
    core.y = normalise_column_exact(core.y, 1, P);
    core.z = normalise_column_exact(t, 2, P);
    core.w = normalise_column_exact(exact_join(core.w, lo.y), 3, P);
    if (V::EXTRA > 0) lam.x = normalise_column_exact(exact_join(lam.x, lo.z), 4, P);
    if (V::EXTRA > 1) lam.y = normalise_column_exact(exact_join(lam.y, lo.w), 5, P);
    if (V::EXO) best = make_float2(normalise_column_exact(P.exo_depth_f64[0], V::EXO_COL, P), normalise_column_exact(P.exo_depth_f64[1], V::EXO_COL + 1, P));
  }
  store_row_values<V, STORE>(base, lane, core, lam, best);
}

template <class V>
__device__ __forceinline__ void store_lo(int32_t* base, uint32_t lane, const int4 lo) {
  if (V::RES == 4) store_through(reinterpret_cast<float4*>(base) + lane, make_float4(__builtin_bit_cast(float, lo.x), __builtin_bit_cast(float, lo.y), __builtin_bit_cast(float, lo.z), __builtin_bit_cast(float, lo.w)));
  else if (V::EXACT_LAM) store_through(reinterpret_cast<float2*>(base) + lane, make_float2(__builtin_bit_cast(float, lo.z), __builtin_bit_cast(float, lo.w)));
  else store_through(reinterpret_cast<float2*>(base) + lane, make_float2(__builtin_bit_cast(float, lo.x), __builtin_bit_cast(float, lo.y)));
}

// Arithmetic and stores of one lane; returns its reward (0 for a pad lane) and counts a clip.
// MIRROR: the instantiation for small batches over the host API (see step_body) also writes what env.step() returns into host memory.
template <class V, bool MIRROR = false>
__device__ __forceinline__ float finish_lane(const StepBuffers& B, const StepParams& P, uint32_t lane, const LaneLoads& L,
                                             const LaneDraw& d, bool& clipped, float* staged_row, const float z = 0.f, const float2 zu = make_float2(0.f, 0.f)) {
  const LaneResult r = V::PRECISE ? lane_step_exact<V>(L.core, L.lam, L.lo, L.act, d, L.qi, P.is_terminal != 0, P, z, P.t_now, P.t_next_f64, zu, L.hs)
                                  : lane_step<V>(L.core, L.lam, L.act, d, L.qi, P.t_next, P.is_terminal != 0, P, z, P.t_now, zu, L.hs, P.t_next_f64, L.lo);
  if (V::RES != 0) store_lo<V>(B.resid, lane, r.lo);
  if (V::DIM == 4) {
    store_row<V>(B.state_out, lane, r.core, r.lam, false, P);
  } else if (V::DIM == 5) {  // (20-byte rows are only 4-byte aligned in LDS too)
    staged_row[0] = r.core.x; staged_row[1] = r.core.y; staged_row[2] = r.core.z; staged_row[3] = r.core.w; staged_row[4] = r.lam.x;
  } else {  // rows wider than 16 bytes go through LDS (see step_kernel): this lane's row, 8-byte pieces
    float2* row = reinterpret_cast<float2*>(staged_row);
    row[0] = make_float2(r.core.x, r.core.y);
    row[1] = make_float2(r.core.z, r.core.w);
    if (V::DIM == 6) {
      row[2] = V::EXTRA == 2 ? r.lam : make_float2(P.exo_depth[0], P.exo_depth[1]);
    } else {
      row[2] = r.lam;
      row[3] = make_float2(P.exo_depth[0], P.exo_depth[1]);
    }
  }
  store_through(B.reward + lane, r.reward);
  // -- optional outputs (wave-uniform branches)
  if (V::NORM && B.obs != nullptr) {
    if (V::PRECISE) store_row_exact<V>(B.obs, lane, r.core, r.lam, r.lo, P.t_next_f64, P);
    else store_row<V>(B.obs, lane, r.core, r.lam, true, P);
  }
  if (B.events != nullptr) B.events[lane] = static_cast<uint8_t>(event_byte(r));
  if (B.lane_returns != nullptr) B.lane_returns[lane] += r.reward;
  if (MIRROR && lane < P.n) {  // small-batch host API: the row and the reward as env.step() returns them, straight into host memory
    B.host_reward[lane] = r.reward;
    if (V::PRECISE) store_row_exact<V, kStorePlain>(B.host_obs, lane, r.core, r.lam, r.lo, P.t_next_f64, P);
    else store_row<V, kStorePlain>(B.host_obs, lane, r.core, r.lam, V::NORM, P);
    if (V::HOST != 0 && B.host_state != nullptr) {  // host-callback plugins: the float64 state (row + remainders) and the events, for update() / calculate()
      store_row<V, kStorePlain>(B.host_state, lane, r.core, r.lam, false, P);
      if (V::RES == 4) reinterpret_cast<int4*>(B.host_resid)[lane] = r.lo;
      else if (V::RES == 2) reinterpret_cast<int2*>(B.host_resid)[lane] = V::EXACT_LAM ? make_int2(r.lo.z, r.lo.w) : make_int2(r.lo.x, r.lo.y);
      B.host_events[lane] = static_cast<uint8_t>(event_byte(r));
    }
  }
  clipped = r.clipped_q | r.clipped_c;
  return r.reward;
}

// Small-batch host API: tell the host that this launch's mirror (StepBuffers::host_obs / host_reward) is complete, without a
// second launch and without an interrupt.  Every thread makes its stores to host memory visible (system-scope fence), the
// workgroup meets, one thread counts the workgroup in; the last workgroup of the launch re-arms the counter and writes the
// launch's sequence number where the host is spinning.  (Measured, tools/microbench/mb_sync.hip: hipStreamSynchronize costs
// ~12.5 us around one kernel, a flag written by a follow-up one-thread kernel 8.5; this needs neither.)
// A SEPARATE instantiation (step_body's MIRROR), not a run-time branch in every kernel: measured as a wave-uniform branch on a
// kernel argument, the mirror + this epilogue cost the benchmark kernel 0.8 us of its 6.65 (2^20 lanes, five workgroups per
// CU; 0.7 us for its precise_state twin, 0.1-0.3 for the heavier kernels: profiles/r04_experiments.txt) - code that never ran.
__device__ __forceinline__ void signal_host(const StepBuffers& B) {
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t arrived = __hip_atomic_fetch_add(B.done_counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (arrived == gridDim.x) {
      __hip_atomic_store(B.done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(B.host_flag, B.flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---- graph-capturable stepping: the clock on the device (mbt_env_device_clock_begin / mbt_env_step_device_captured) -------------
// A launch of step_kernel carries the step's clock as kernel ARGUMENTS the host computed (philox_step, t_next, is_terminal:
// mbt_env.hip, launch_step) - so a HIP graph that captured [policy, step] x k would replay the same k steps forever, and a consumer
// whose policy lives on the device (torch) pays 4-5 us of host time per enqueued launch: below ~2^19 lanes its loop is launch-bound.
// These instantiations read the clock from a block of device memory instead and the LAST workgroup of the launch advances it, with
// the host's arithmetic (t += dt in double, TE:216; terminal per TE:218-220; the Philox step counts on) - the kernel arguments are
// then the same for every step, and a captured graph replays correctly.  An episode's end is handled inside the same launch, the way
// mbt_env_step_many_device(auto_reset) handles it with two more launches (SBE:28-37): every workgroup overwrites the rows of its own
// tile with the reset row (after saving the terminal observation, if asked to), the last workgroup reduces the episode's return sums
// - in reduce_returns_kernel's order of additions: the same bits - into a log in the clock block and zeroes the accumulators.
// Results are those of the launch_step loop to the bit: same step_tile / speed_step_body, same counters, same clock arithmetic.
//
// Who may write the next step's clock, and when?  Every workgroup of a launch reads the clock, so whoever overwrites it has to know
// that all of them have - and on this device finding that out costs more than the arithmetic of a step: atomics on one cache line are
// serialised at ~12 ns each (8192 waves counting on one word: 97 us for a 6.6 us launch), a returning atomic per workgroup on 64 lines
// + a top-level word still adds 1.7 us at 2^20 lanes, fire-and-forget atomics polled by the last workgroup 4 us
// (tools/microbench/mb_captured.hip, profiles/r06_graph_step.txt).  So nobody counts: the clock has TWO slots, a launch reads slot
// `parity` and its workgroup 0 writes slot `parity ^ 1` - which no workgroup of the launch reads - whenever it gets there.  `parity`
// is a kernel argument that the HOST alternates from one mbt_env_step_device_captured call to the next; inside a captured graph it is
// therefore baked into the nodes, and the first call of every capture (and every call outside one) is preceded by a one-thread
// ALIGN kernel that copies the current slot into slot 0 - so a graph of any length, replayed any number of times in any order with
// other graphs or single calls, always starts from slot 0 with parity 0.  Cost on the hot path: one 16-byte SCALAR load per wave in
// front of the generator (+0.2 us at 2^20 lanes, +0.4 us where a launch is latency-bound) and a few stores in workgroup 0.
// Only the launch that ENDS an episode counts its waves (fire-and-forget atomics, polled by the workgroup dispatched last, which
// then files the episode's return sums): once per episode, 4 us.
// The state is stepped IN PLACE (state_out == state_in), so the observation a captured policy reads has one address.

// reset (TE:131-140): rows [initial_cash, q0, start_time, initial_price, process columns...]
struct ResetRow {
  float cash0, t0, s0, q0_scalar;
  float extra[4];  // columns 4..: Hawkes baselines (ARR:103), exogenous best depths (FILL:148-154), or the impact model's initial state (IMP:81, IMP:121)
  // precise_state: the same row as the reference's float64 values (column order of the state row; [1] is unused when the
  // initial inventories are per lane) and the int32 remainders of the residual columns (exact_split on the host)
  double exact[8];
  int32_t lo[4];
  int32_t res;  // residual columns per lane: 0, 2 or 4
  int32_t precise;  // precise_state: observations are normalised from the float64 values (res != 0 alone may be the float32 tier's exact intensities)
};

// one lane of a reset: its state row, its remainders, its (normalised) observation row, its running return
__device__ __forceinline__ void reset_lane(uint32_t i, float* state, float* obs, float* lane_returns, const float* q0, const ResetRow& row0, int dim,
                                           const StepParams& P, int32_t* resid) {
  if (resid != nullptr)
    for (int j = 0; j < row0.res; ++j) resid[static_cast<size_t>(i) * row0.res + j] = row0.lo[j];  // what float32 left of the initial values
  float* row = state + static_cast<size_t>(i) * dim;
  float* orow = obs != nullptr ? obs + static_cast<size_t>(i) * dim : nullptr;
  for (int j = 0; j < dim; ++j) {
    const float v = j == 0 ? row0.cash0 : j == 1 ? (q0 != nullptr ? q0[i] : row0.q0_scalar) : j == 2 ? row0.t0 : j == 3 ? row0.s0 : row0.extra[j - 4];
    row[j] = v;
    if (orow != nullptr) {
      if (row0.precise != 0) {  // precise_state: normalised from the float64 value, like the reference (TE:112-118)
        const double x = (j == 1 && q0 != nullptr) ? static_cast<double>(q0[i]) : row0.exact[j];
        orow[j] = P.norm_obs ? normalise_column_exact(x, j, P) : v;
      } else {
        orow[j] = P.norm_obs ? normalise_column(v, j, P) : v;
      }
    }
  }
  if (lane_returns != nullptr) lane_returns[i] = 0.0f;
}

constexpr uint32_t kClockLogSlots = 16;
// The clock block.  Its first 32 bytes are `struct mbt_device_clock` of include/mbt_env.h: a MIRROR of the current slot that no step
// kernel reads (what mbt_env_device_clock_read returns and what a device consumer may read through mbt_env_device_clock_ptr,
// e.g. `done` as a mask).
struct ClockSlot {
  double time;             // the clock at the beginning of the next step (TE:216)
  uint32_t episode_step;   // steps since the last (explicit or automatic) reset
  uint32_t philox_step;    // Philox counter word 2 of the next step
  uint32_t steps;          // steps taken since mbt_env_device_clock_begin
  uint32_t episodes;       // steps among them that ended an episode
  int32_t done;            // the last step ended an episode (TE:218-220)
  uint32_t reserved;
};
struct DeviceClock {
  ClockSlot shown;         // the mirror; `shown.reserved` is mbt_device_clock::log_count: entries written to `log` since begin (entry k sits in slot k % kClockLogSlots)
  ClockSlot slot[2];       // a launch reads slot[parity] (the first 16 bytes, as one scalar load), its workgroup 0 writes slot[parity ^ 1]
  uint32_t current;        // the slot the NEXT launch is to read (captured_align_kernel brings it to 0)
  uint32_t faults;         // episode-end launches whose last workgroup gave up waiting for the others (never, on a healthy device): mbt_env_device_clock_read reports it
  uint32_t reserved[6];
  double log[kClockLogSlots][3];  // [sum R, sum R^2 (NaN unless per-lane returns are tracked), lanes] of the newest finished episodes
};
struct CapturedParams {
  DeviceClock* clock;
  uint32_t* counters;      // an episode-end launch counts its waves: word 16 g those of the g-th group of 32 workgroups that have finished (a 64-byte line each)
  uint32_t parity;         // the clock slot this launch reads (it writes the other)
  uint32_t reserved_pad;
  double dt_f64, terminal_time, t_start;  // the host's clock arithmetic (mbt_env.hip: launch_step); where an automatic reset restarts
  int32_t auto_reset;      // an episode's end resets the lanes and logs the return sums inside the launch
  int32_t dim;
  uint32_t tile_lanes;     // lanes per workgroup: 512 (order book) or 1024 (speed dynamics)
  uint32_t n_waves;        // slots of wave_sums
  float* obs;              // the normalised observation buffer (nullptr: the state row is the observation)
  float* terminal_obs;     // (n_pad, D): receives the observation of an episode's last step before the reset overwrites it; or nullptr
  const float* q0;         // per-lane initial inventories of the last explicit reset, or nullptr
  ResetRow row0;           // the row an automatic reset writes
};

struct CapturedStep {  // the clock of the step a launch takes
  double t, t_next;
  uint32_t philox_step;
  bool terminal;
};

// The clock of this step, wave-uniform (SGPRs), in two halves.  captured_clock_issue: ONE scalar load of the slot's first 16 bytes
// [time, episode step, Philox step], issued before the state / action loads of the tile.  captured_prologue: waits for it (scalar
// loads have a counter of their own - lgkmcnt - so the vector loads stay in flight, which a wait for a vector load issued behind
// them would not allow: vmcnt counts in order) and sets P's per-step fields from it.  The slot was written by the PREVIOUS launch
// (its workgroup 0, or the align kernel) and nobody writes it during this one: the scalar cache, invalid at the launch's start like
// every cache, cannot hold a stale copy.
typedef uint32_t clock_words_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ clock_words_t captured_clock_issue(const ClockSlot* slot) {
  clock_words_t w;
  asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(w) : "s"(slot) : "memory");
  return w;
}
__device__ __forceinline__ CapturedStep captured_prologue(clock_words_t w, const CapturedParams& C, StepParams& P) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w) : : "memory");
  CapturedStep s;
  s.t = __builtin_bit_cast(double, (static_cast<uint64_t>(w.y) << 32) | w.x);
  s.t_next = s.t + C.dt_f64;                                      // TE:216, exactly as launch_step advances the host's clock
  s.terminal = s.t_next >= C.terminal_time - C.dt_f64 / 2;        // TE:218-220
  s.philox_step = w.w;
  P.philox_step = w.w;
  P.is_terminal = s.terminal ? 1 : 0;
  P.t_next = static_cast<float>(s.t_next);
  P.t_now = s.t;
  P.t_next_f64 = s.t_next;
  return s;
}

// reduce_returns_kernel's sums by ONE wave, in that kernel's order of additions: its thread t accumulates the elements t, t + 256,
// ... and a tree over shared memory folds thread t + s into thread t for s = 128, 64, ..., 1.  Here lane l of the wave plays the
// threads l, l + 64, l + 128, l + 192 (four accumulators), the folds s = 128 and s = 64 are additions between its own accumulators,
// the folds s = 32 ... 1 shuffles within the wave.  `zero`: the accumulators are cleared behind the reads (an automatic reset).
__device__ __forceinline__ void captured_reduce_returns(double* wave_sums, uint32_t n_waves, float* lane_returns, uint32_t n, double out[3]) {
  // (the wave sums are written by device-scope atomics of waves on every XCD and read here the same way - relaxed device-scope loads,
  // EIGHT in flight per accumulator, added in index order; the per-lane returns are plain memory behind the caller's acquire fence)
  const uint32_t l = threadIdx.x & 63u;
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  constexpr uint32_t kAhead = 8;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    for (uint32_t i0 = l + 64u * v; i0 < n_waves; i0 += 256u * kAhead) {
      double x[kAhead];
#pragma unroll
      for (uint32_t k = 0; k < kAhead; ++k) {
        const uint32_t i = i0 + 256u * k;
        x[k] = i < n_waves ? __hip_atomic_load(wave_sums + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
#pragma unroll
      for (uint32_t k = 0; k < kAhead; ++k) {
        const uint32_t i = i0 + 256u * k;
        if (i < n_waves) {
          a[v] += x[k];
          __hip_atomic_store(wave_sums + i, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (lane_returns != nullptr)
      for (uint32_t i0 = l + 64u * v; i0 < n; i0 += 256u * kAhead) {
        float r[kAhead];
#pragma unroll
        for (uint32_t k = 0; k < kAhead; ++k) {
          const uint32_t i = i0 + 256u * k;
          r[k] = i < n ? lane_returns[i] : 0.0f;
        }
#pragma unroll
        for (uint32_t k = 0; k < kAhead; ++k)
          if (i0 + 256u * k < n) b[v] += static_cast<double>(r[k]) * r[k];
      }
  }
  a[0] += a[2]; a[1] += a[3]; b[0] += b[2]; b[1] += b[3];  // s = 128
  a[0] += a[1]; b[0] += b[1];                              // s = 64
  for (int s = 32; s > 0; s >>= 1) {
    a[0] += __shfl_down(a[0], s, 64);
    b[0] += __shfl_down(b[0], s, 64);
  }
  out[0] = a[0];
  out[1] = lane_returns != nullptr ? b[0] : __builtin_nan("");
  out[2] = static_cast<double>(n);
}

// Behind the step of a tile: the episode's end (every workgroup, its own rows), the next step's clock (workgroup 0), and - at an
// episode's end - the count and the episode's log entry (the workgroup dispatched last, once every wave of the launch is in).
__device__ __forceinline__ void captured_epilogue(const StepBuffers& B, const StepParams& P, const CapturedParams& C, const CapturedStep s) {
  const bool episode_end = s.terminal && C.auto_reset != 0;
  if (__builtin_expect(episode_end, 0)) {
    // This workgroup's own stores of the step are complete and visible to all ITS threads before any of its rows is rewritten: a
    // release / barrier / acquire at WORKGROUP scope (the stores have left for the L2 the workgroup shares; its L1 is dropped).  Not at
    // device scope: that writes the whole L2 back, once per wave - 8192 waves x 3 fences made the episode-end launch 320 us at 2^20
    // lanes (profiles/r06_graph_step.txt).  Nothing here is read by another workgroup of this launch - except the return accumulators, below.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t first = blockIdx.x * C.tile_lanes;
    float* state = B.state_out;
    if (C.terminal_obs != nullptr) {
      const float* shown = C.obs != nullptr ? C.obs : state;
      const size_t base = static_cast<size_t>(first) * C.dim, count = static_cast<size_t>(C.tile_lanes) * C.dim;
      for (size_t k = threadIdx.x; k < count; k += kBlockThreads) C.terminal_obs[base + k] = shown[base + k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
    }
    // (the running returns are cleared by the last workgroup, which needs them all for the sum of squares first)
    for (uint32_t i = first + threadIdx.x; i < first + C.tile_lanes; i += kBlockThreads) reset_lane(i, state, C.obs, nullptr, C.q0, C.row0, C.dim, P, B.resid);
    // What the episode's filer (the workgroup dispatched last, possibly on another XCD) reads of this workgroup: its waves' return sums -
    // device-scope atomics, coherent by themselves, which only have to be COMPLETE before the wave counts itself in (a workgroup-scope
    // release is the wait for that) - and, when they are tracked, the per-lane returns, plain stores that do need the L2 written back.
    if (B.lane_returns != nullptr) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
  }
  DeviceClock* clock = C.clock;
  // workgroup 0 writes the next step's clock into the slot no workgroup of this launch reads, and its mirror
  if (blockIdx.x == 0u && threadIdx.x == 0u) {
    const ClockSlot& mine = clock->slot[C.parity];
    ClockSlot next;
    next.time = episode_end ? C.t_start : s.t_next;
    next.episode_step = episode_end ? 0u : mine.episode_step + 1u;
    next.philox_step = s.philox_step + 1u;
    next.steps = mine.steps + 1u;
    next.episodes = mine.episodes + (s.terminal ? 1u : 0u);
    next.done = s.terminal ? 1 : 0;
    next.reserved = 0u;
    clock->slot[C.parity ^ 1u] = next;
    clock->current = C.parity ^ 1u;
    clock->shown.time = next.time; clock->shown.episode_step = next.episode_step; clock->shown.philox_step = next.philox_step;
    clock->shown.steps = next.steps; clock->shown.episodes = next.episodes; clock->shown.done = next.done;  // (shown.reserved = log_count: the episode filer's, below)
  }
  if (!episode_end) return;
  // An episode's end, once in n_steps launches: every wave counts itself in (fire and forget) behind its stores ...
  constexpr uint32_t kWaves = kBlockThreads / 64;
  if ((threadIdx.x & 63u) == 0u) (void)__hip_atomic_fetch_add(C.counters + 16u * (blockIdx.x >> 5), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x + 1u != gridDim.x || threadIdx.x >= 64u) return;
  // ... the first wave of the workgroup that is dispatched last stays until all the others are in (one counter line per lane and poll;
  // a wall-clock bound like every wait on the device: a launch that lost a workgroup must not leave this one spinning), re-arms the
  // counters ...
  {
    const uint32_t groups = (gridDim.x + 31u) >> 5;
    const uint64_t t0 = wall_clock64();
    for (;;) {
      bool all_in = true;
      for (uint32_t g = threadIdx.x; g < groups; g += 64u) {
        const uint32_t members = gridDim.x - (g << 5) < 32u ? gridDim.x - (g << 5) : 32u;
        all_in &= __hip_atomic_load(C.counters + 16u * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members * kWaves;
      }
      if (__builtin_amdgcn_ballot_w64(!all_in) == 0ull) break;
      if (wall_clock64() - t0 > 200000000ull) {  // 2 s of the 100 MHz wall clock
        if (threadIdx.x == 0u) clock->faults += 1u;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    for (uint32_t g = threadIdx.x; g < groups; g += 64u) __hip_atomic_store(C.counters + 16u * g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ... and, alone on the launch now, files the episode: its return sums into the log, the accumulators back to zero
  if (B.lane_returns != nullptr) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (the per-lane returns are plain memory: see above)
  double sums[3];
  captured_reduce_returns(B.wave_sums, C.n_waves, B.lane_returns, P.n, sums);
  if (B.lane_returns != nullptr)
    for (uint32_t i = threadIdx.x; i < 2u * P.n_pairs; i += 64u) B.lane_returns[i] = 0.0f;
  if (threadIdx.x == 0u) {
    const uint32_t log_count = clock->shown.reserved;
    double* entry = clock->log[log_count % kClockLogSlots];
    entry[0] = sums[0]; entry[1] = sums[1]; entry[2] = sums[2];
    clock->shown.reserved = log_count + 1u;
  }
}

// STREAM: the state / action loads carry the non-temporal bit.  Chosen by the host (mbt_env.hip: tune_for_size) when one
// launch's working set exceeds the Infinity Cache - nothing read now is still cached at the next step, so it should not
// displace lines on its way in: 128.3 -> 118.7 us for the 44-byte copy kernel at 2^24 lanes (profiles/r01_microbench.txt) -
// and a LOSS where the working set does fit (2^20..2^22 lanes), hence two instantiations rather than one policy.  (A
// run-time branch around the two load sequences does not survive the optimiser: it merges the branches' loads and drops
// the hint.)
// MIRROR: small batches over the host API (mbt_env_step_host, N <= 65536): the kernel also mirrors its outputs into
// device-mapped host memory and raises a completion flag there (signal_host) - ONE launch per env.step(), no interrupt.
// `tile`: the 512-lane tile this workgroup steps (blockIdx.x for the ordinary kernels: one tile per workgroup; the resident
// small-batch kernel walks several tiles per workgroup and step)
// CAPTURED: the graph-capturable instantiation - the step's clock comes from device memory (`C`, see above), `P_in` holds what
// does not change from step to step; `captured` receives the step's clock for captured_epilogue.
template <class V, bool STREAM = false, bool MIRROR = false, bool CAPTURED = false>
__device__ __forceinline__ void step_tile(const StepBuffers& B, const StepParams& P_in, const uint32_t tile, const CapturedParams* C = nullptr,
                                          CapturedStep* captured = nullptr) {
  static_assert(!(STREAM && MIRROR), "streaming loads are for launches beyond the Infinity Cache, the mirror for small batches");
  static_assert(!(CAPTURED && (MIRROR || V::INJECT || V::HOST != 0)), "a captured step has no host in its loop");
  const uint32_t lane0 = tile * kTileLanes + threadIdx.x, lane1 = lane0 + kBlockThreads;
  clock_words_t clock_words = {0u, 0u, 0u, 0u};
  if (CAPTURED) clock_words = captured_clock_issue(&C->clock->slot[C->parity]);
  LaneLoads L0 = load_lane<V, STREAM>(B, P_in, lane0), L1 = load_lane<V, STREAM>(B, P_in, lane1);  // issue every load ...
  load_initial_inventories<V>(B, lane0, lane1, L0.qi, L1.qi);
  StepParams P_step;  // (CAPTURED only: the kernel arguments with this step's clock filled in)
  if (CAPTURED) {
    P_step = P_in;
    *captured = captured_prologue(clock_words, *C, P_step);
  }
  // (Reading the parameters from the kernel-argument segment only HERE, behind the loads - a scheduling barrier and the segment's pointer
  // passed through an empty asm: 6.39 -> 5.98 us in the micro-benchmark - makes every kernel slower, AS 6.63 -> 6.86 us: the generator
  // below needs its key and counters at once, and its ~120 instructions are what hides the loads.  profiles/r06_kernarg_layout.txt.)
  const StepParams& P = CAPTURED ? P_step : P_in;
  const uint64_t pair = P.pair_offset + tile * kBlockThreads + threadIdx.x;
  LaneNoise nz0, nz1;
  LaneDraw d0, d1;
  float2 zu0 = L0.zu, zu1 = L1.zu;
  if (V::USER_DRAWS && !V::INJECT) philox_pair_user_noise(pair, P.philox_step, P.key0, P.key1, zu0.x, zu0.y, zu1.x, zu1.y);
  if (V::INJECT) {
    nz0 = LaneNoise{L0.ua.x, L0.ua.y, L0.uf.x, L0.uf.y, L0.z};
    nz1 = LaneNoise{L1.ua.x, L1.ua.y, L1.uf.x, L1.uf.y, L1.z};
    d0 = make_draw<V>(nz0, P);
    d1 = make_draw<V>(nz1, P);
  } else {
    philox_pair_noise(pair, P.philox_step, P.key0, P.key1, nz0, nz1);  // ... draw while they fly
    d0 = make_draw<V>(nz0, P);
    d1 = make_draw<V>(nz1, P);
    tie_loads_to_draws(L0, L1, d0, d1);
  }
  // Rows of 24 or 32 bytes (Hawkes intensities, exogenous depths) would leave a thread as several stores that each cover
  // part of a cache line, which cannot be written through the L2 (store_through).  The workgroup assembles its 512 rows
  // in LDS instead (12 / 16 KB) and writes them out as contiguous whole-line float4 - through the L2.
  __shared__ __attribute__((aligned(16))) float staged_rows[V::DIM > 4 ? kTileLanes * V::DIM : 4];
  bool clipped0, clipped1;
  float r0 = finish_lane<V, MIRROR>(B, P, lane0, L0, d0, clipped0, staged_rows + threadIdx.x * V::DIM, nz0.z, zu0);
  float r1 = finish_lane<V, MIRROR>(B, P, lane1, L1, d1, clipped1, staged_rows + (threadIdx.x + kBlockThreads) * V::DIM, nz1.z, zu1);
  if (V::DIM > 4) {
    __syncthreads();
    constexpr int kTileVectors = kTileLanes * V::DIM / 4;  // float4 per tile: 3 (D = 6) or 4 (D = 8) per thread; 2.5 for D = 5
    const float4* staged = reinterpret_cast<const float4*>(staged_rows);
    float4* out = reinterpret_cast<float4*>(B.state_out) + static_cast<size_t>(tile) * kTileVectors;
#pragma unroll
    for (int k = 0; k * kBlockThreads < kTileVectors; ++k)
      if ((k + 1) * kBlockThreads <= kTileVectors || threadIdx.x + k * kBlockThreads < kTileVectors)
        store_through(out + threadIdx.x + k * kBlockThreads, staged[threadIdx.x + k * kBlockThreads]);
  }
  if ((tile + 1u) * kTileLanes > P.n) {  // only the last tile can hold pad lanes: computed, never reported
    const bool real0 = lane0 < P.n, real1 = lane1 < P.n;
    r0 = real0 ? r0 : 0.0f;
    r1 = real1 ? r1 : 0.0f;
    clipped0 &= real0;
    clipped1 &= real1;
  }
  const float r_sum = r0 + r1;
  // clipped lanes of this wave, counted on the scalar unit (two ballots + popcounts)
  const uint32_t clips = __builtin_popcountll(__builtin_amdgcn_ballot_w64(clipped0)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(clipped1));
  // -- per-wave running sum of rewards (numerator of the mean episode return): one slot per wave, one
  //    fire-and-forget hardware fp64 atomic per wave, no contention
  const float total = wave_sum(r_sum);
  if ((threadIdx.x & 63u) == 0u) {
    const uint32_t wave = tile * (kBlockThreads / 64) + (threadIdx.x >> 6);
    unsafeAtomicAdd(&B.wave_sums[wave], static_cast<double>(total));
    if (__builtin_expect(clips != 0u, 0)) atomicAdd(&B.clip_count[wave & (kClipSlots - 1u)], static_cast<unsigned long long>(clips));
  }
}

template <class V, bool STREAM = false, bool MIRROR = false>
__device__ __forceinline__ void step_body(const StepBuffers& B, const StepParams& P) {
  step_tile<V, STREAM, MIRROR>(B, P, blockIdx.x);
  if (MIRROR) signal_host(B);
}

// (the body is a device function so that the run-time compiled kernels of mbt_env_create_jit - plain extern "C" entry
// points around one instantiation - share it with the ahead-of-time instantiations)
template <class V, bool STREAM = false, bool MIRROR = false>
__global__ __launch_bounds__(kBlockThreads) void step_kernel(const StepBuffers B, const StepParams P) {
  step_body<V, STREAM, MIRROR>(B, P);
}

template <class V, bool STREAM = false>
__device__ __forceinline__ void captured_step_body(const StepBuffers& B, const StepParams& P0, const CapturedParams& C) {
  CapturedStep s = {0.0, 0.0, 0u, false};
  step_tile<V, STREAM, false, true>(B, P0, blockIdx.x, &C, &s);
  captured_epilogue(B, P0, C, s);
}

template <class V, bool STREAM = false>
__global__ __launch_bounds__(kBlockThreads) void captured_step_kernel(const StepBuffers B, const StepParams P, const CapturedParams C) {
  captured_step_body<V, STREAM>(B, P, C);
}

// ---- resident small-batch stepping (opt-in: MBT_RESIDENT_STEP=1; mbt_env.hip: resident_step) ---------------------------------
// In the reference's own regime (N ~ 1000) one env.step() is ONE launch of the MIRROR instantiation, and what is left of its 14 us is
// mostly the ~6 us between the launch call and the kernel's first instruction.  This kernel removes it by staying on the device:
// at most four workgroups (each walks its share of the tiles) wait for the host to ring a doorbell - a sequence number in a MAILBOX
// line that also carries where this step's observation rows and rewards are to be mirrored - step every lane exactly like
// step_kernel<V, false, true> (same step_tile, same Philox counters, same clock arithmetic as the host: t += dt in double, TE:216;
// terminal per TE:218-220), raise the completion flag (signal_host) and go back to waiting.  It leaves by itself when the episode
// ends, when told to (kResidentExit: any other call on the environment), after `idle_ticks` without a doorbell and after
// `life_ticks` in any case - every wait on the device has a wall-clock bound (100 MHz wall_clock64): a host that died cannot leave
// the device spinning.  On leaving without having handled step `seq` it says so in host memory (`host_exit`), so a host that rang
// the doorbell in the same instant starts another kernel at that step instead of waiting for an answer that will not come.
// Measured (tools/microbench/mb_resident.hip, profiles/r05_resident_step.txt): 16.1 -> 9.6 us per mock step at N = 1000 with the
// mailbox and the actions in device memory the host writes through the PCIe BAR, 12.1 us with both in host memory; a bandwidth-bound
// kernel on ANOTHER stream runs 20-27 % slower beside it - which is why this is opt-in.
constexpr uint32_t kResidentExit = 0xFFFFFFFFu;
struct ResidentMailbox {  // one 64-byte line, written by the host as a whole (write-combined) before every step
  uint64_t host_obs;      // where this step's (n, D) observation rows go (the device's address of device-visible host memory)
  uint64_t host_reward;   // ... and its (n) rewards
  uint32_t seq;           // the step's sequence number = the value its completion flag will take; kResidentExit: leave
  uint32_t reserved[11];
};
struct ResidentParams {
  const ResidentMailbox* mailbox;  // device memory the host can write (fine-grained, through the BAR) or device-mapped host memory
  uint32_t* host_exit;             // host memory: the sequence number of the step the kernel left BEFORE handling
  uint32_t* control;               // device memory: workgroup 0's decision for the current step, read by the other workgroups
  uint32_t first_seq, n_tiles;
  double t_start, dt_f64, terminal_time;  // the host's clock (mbt_env.hip: launch_step), advanced here with the same arithmetic
  uint64_t idle_ticks, life_ticks;
  int32_t ping_pong;
  uint32_t generation;             // of this launch (the host counts them): tags the "leave" word in `control`
};

template <class V>
__device__ __forceinline__ void resident_body(const StepBuffers& B0, const StepParams& P0, const ResidentParams& R) {
  StepBuffers B = B0;
  StepParams P = P0;
  __shared__ uint32_t s_cmd;
  __shared__ uint64_t s_out[2];
  double t = R.t_start;
  const uint64_t born = wall_clock64();
  // ONE decision per step for the whole grid: workgroup 0 watches the mailbox and publishes "go(seq)" or "leave" in device memory
  // (`control`), the others watch that word, and a step is taken by every workgroup or by none.  (Every workgroup deciding for itself
  // was wrong: one timed out while its neighbour saw the doorbell that was rung in the same microsecond, and half a step ran.)  The leave token carries the launch's generation, so a
  // word left behind by an earlier kernel is never mistaken for this one's.
  const uint32_t leave = 0x80000000u | (R.generation & 0x7FFFFFFFu);
  // (sequence numbers count up and wrap, skipping kResidentExit - the one value that means "leave" - exactly as the host's counter does,
  // mbt_env.hip: resident_step: a kernel alive across the wrap, after 4.29e9 steps, keeps answering the number the host waits for)
  for (uint32_t seq = R.first_seq;; seq = (seq + 1u == kResidentExit) ? seq + 2u : seq + 1u) {
    if (threadIdx.x == 0) {
      const uint64_t t0 = wall_clock64();
      uint32_t cmd;
      if (blockIdx.x == 0) {
        // `control` changes only once EVERY workgroup has finished the previous step (the last to arrive zeroes the counter,
        // signal_host; this workgroup has arrived, so the counter is non-zero until then) - and the idle clock below starts only
        // then.  Without this, on a busy device: this workgroup finishes step n and waits for doorbell n + 1, which the host rings
        // only after flag n, which needs a neighbour that has not even seen "go(n)" yet; the idle bound runs out, "leave" replaces
        // "go(n)", the neighbour leaves without its tiles, the flag never rises and a third of the lanes is a step behind (found
        // by a soak with eight processes in resident mode, profiles/r05_soak.txt; the lifetime bound and the host's answer
        // time-out had the same hole).  In the ordinary course this load finds 0 while the host is still reading the results.
        if (seq != R.first_seq)
          while (__hip_atomic_load(B.done_counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u && wall_clock64() - born <= 2 * R.life_ticks) __builtin_amdgcn_s_sleep(1);
        for (;;) {
          cmd = __hip_atomic_load(&R.mailbox->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
          if (cmd == kResidentExit || static_cast<int32_t>(cmd - seq) >= 0) break;
          const uint64_t now = wall_clock64();
          if (now - t0 > R.idle_ticks || now - born > R.life_ticks) { cmd = kResidentExit; break; }
          __builtin_amdgcn_s_sleep(2);
        }
        __hip_atomic_store(R.control, cmd == kResidentExit ? leave : (seq & 0x7FFFFFFFu), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // ("go" tokens live in the lower half, "leave" tokens in the upper)
      } else {
        for (;;) {
          const uint32_t word = __hip_atomic_load(R.control, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          if (word == (seq & 0x7FFFFFFFu)) { cmd = seq; break; }
          if (word == leave || wall_clock64() - born > 2 * R.life_ticks) { cmd = kResidentExit; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      s_cmd = cmd;
      if (cmd != kResidentExit) {  // (the line was written as a whole and stays as it is until this step's flag is up)
        s_out[0] = __hip_atomic_load(&R.mailbox->host_obs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_out[1] = __hip_atomic_load(&R.mailbox->host_reward, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __syncthreads();
    const uint32_t cmd = s_cmd;
    B.host_obs = reinterpret_cast<float*>(s_out[0]);
    B.host_reward = reinterpret_cast<float*>(s_out[1]);
    __syncthreads();
    // Every wave drops what its CU's vector cache still holds: the actions the host has just rewritten, and the state rows this
    // very kernel wrote a step ago (a kernel boundary does this for the one-launch-per-step path).
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (cmd == kResidentExit) {
      if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(R.host_exit, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    const double t_next = t + R.dt_f64;  // TE:216, exactly as launch_step advances the host's clock
    const bool terminal = t_next >= R.terminal_time - R.dt_f64 / 2;  // TE:218-220
    P.is_terminal = terminal ? 1 : 0;
    P.t_next = static_cast<float>(t_next);
    P.t_now = t;
    P.t_next_f64 = t_next;
    B.flag_value = seq;
    for (uint32_t tile = blockIdx.x; tile < R.n_tiles; tile += gridDim.x) {
      step_tile<V, false, true>(B, P, tile);
      if (V::DIM > 4) __syncthreads();  // (the next tile re-uses the staged rows)
    }
    signal_host(B);
    if (terminal) return;  // the episode is over: the host (which computed the same flag) resets before it steps again
    t = t_next;
    P.philox_step += 1;
    if (R.ping_pong) {
      float* previous = const_cast<float*>(B.state_in);
      B.state_in = B.state_out;
      B.state_out = previous;
    }
  }
}

template <class V>
__global__ __launch_bounds__(kBlockThreads) void resident_step_kernel(const StepBuffers B, const StepParams P, const ResidentParams R) {
  resident_body<V>(B, P, R);
}

// ---- fused rollout -----------------------------------------------------------------------
// Many consecutive env-steps in ONE launch with an on-device closed-form policy: the pair's state stays in
// registers, noise comes from the same Philox stream the step kernel would draw (philox step = first + k), so a
// rollout is bit-identical to the equivalent sequence of step() calls.  The caller's per-time-step Python loop
// (generate_trajectory.py:21-34) disappears; HBM is touched only to record the trajectory (optional, time-major
// so that every store is a coalesced float4) and once at the end for the final state.
enum : int { kPolicyFixed = 0, kPolicyAvellanedaStoikov = 1, kPolicyTable = 2, kPolicyTimeTable = 3, kPolicyBuffer = 4 };

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

#ifdef MBT_JIT_USER_CODE
struct LearnedPolicyParams {};  // (learned policies are not part of the run-time compiled translation unit)
#endif

// A loaded value is waited for HERE, inside the branch that loaded it.  On gfx9 loads and stores share one counter (vmcnt), so a
// wait for a load also waits for every store issued before it.  The rollout's policy is a run-time switch: two of its branches load
// (a tabulated policy, the action buffer), and the wait the compiler put where the branches MERGE - inside the step loop, in front
// of the first use of the action - made the closed-form policies, which load nothing, wait every step for the previous step's
// recording stores to be acknowledged: waves parked on s_waitcnt 38 % (2^18 lanes) / 47 % (2^20) of their cycles
// (profiles/r05_pmc_rollout.json).  With the wait in the loading branch the loop of the other policies has no vmcnt wait at all.
__device__ __forceinline__ void settle_load(float4& v) { asm volatile("; loaded value settled in its own branch" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// LEARNED: the policy is a linear map or an MLP evaluated on the matrix cores (policy_mlp.hpp) instead of a closed form;
// separate instantiations, so that the closed-form rollouts keep their register budget.
template <class V, bool LEARNED = false>
__device__ __forceinline__ void rollout_body(const StepBuffers& B, const StepParams& P, const RolloutParams& R, const LearnedPolicyParams* LP = nullptr) {
  static_assert(!V::INJECT, "rollouts draw their own noise");
  constexpr int A = (V::DYN == kDynLimitAndMarket) ? 4 : 2;
  const uint32_t lanes[2] = {blockIdx.x * kTileLanes + threadIdx.x, blockIdx.x * kTileLanes + threadIdx.x + kBlockThreads};
  const uint64_t pair = P.pair_offset + blockIdx.x * kBlockThreads + threadIdx.x;
  const size_t n_pad = static_cast<size_t>(P.n_pairs) * 2;
  float4 core[2];
  float2 lam[2];
  int4 lo[2];
  float qi[2], ret[2] = {0.f, 0.f};
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const LaneLoads L = load_lane<V>(B, P, lanes[l]);  // (the action slot of L is ignored: the policy acts here)
    core[l] = L.core;
    lam[l] = L.lam;
    qi[l] = L.qi;
    lo[l] = L.lo;
    if (R.obs_traj != nullptr) {
      if (V::PRECISE) store_row_exact<V, kStoreRecord>(R.obs_traj, lanes[l], core[l], lam[l], lo[l], R.t_start, P);
      else store_row<V, kStoreRecord>(R.obs_traj, lanes[l], core[l], lam[l], V::NORM, P);
    }
  }
  load_initial_inventories<V>(B, lanes[0], lanes[1], qi[0], qi[1]);
  float4 held[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  if (R.policy == kPolicyBuffer) {  // action repeat: each lane keeps the row of the action buffer it was given
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      if (A == 4) held[l] = reinterpret_cast<const float4*>(B.action)[lanes[l]];
      else { const float2 a = reinterpret_cast<const float2*>(B.action)[lanes[l]]; held[l] = make_float4(a.x, a.y, 0.f, 0.f); }
      settle_load(held[l]);
    }
  }
  uint32_t wave_clips = 0;  // clipped lane-steps of this WAVE: a popcount of a ballot, i.e. scalar-unit work (a per-thread counter cost two
                            // vector instructions per lane and step in a loop that is bound by vector issue)
  const bool real[2] = {lanes[0] < P.n, lanes[1] < P.n};
  double t = R.t_start;
  float last_reward[2] = {0.f, 0.f};  // what step() leaves behind for the final step: its rewards and event bytes
  uint32_t last_events[2] = {0u, 0u};
#ifndef MBT_JIT_USER_CODE
  constexpr int kMlpLdsWaveBytesPerBlock = (kBlockThreads / 64) * kMlpLdsBytesPerWave;
  __shared__ __attribute__((aligned(16))) char policy_lds[LEARNED ? kMlpLdsWaveBytesPerBlock + kMlpLdsWeightBytes : 16];
  static_assert(!LEARNED || kBlockThreads == 256, "stage_mlp_weights copies with 256 threads");
  if (LEARNED && !LP->is_linear) stage_mlp_weights(LP->w, policy_lds + kMlpLdsWaveBytesPerBlock);  // (uniform branch: the barrier inside is reached by all or none)
#endif
  const bool recording = R.obs_traj != nullptr || R.act_traj != nullptr || R.rew_traj != nullptr;
  for (uint32_t k = 0; k < R.n_steps; ++k) {
    LaneNoise nz[2];
    if (!LEARNED) philox_pair_noise(pair, P.philox_step + k, P.key0, P.key1, nz[0], nz[1]);
    float2 zu[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
    if (V::USER_DRAWS) philox_pair_user_noise(pair, P.philox_step + k, P.key0, P.key1, zu[0].x, zu[0].y, zu[1].x, zu[1].y);
    float4 act[2];
    if (LEARNED) {
#ifndef MBT_JIT_USER_CODE
      // the observation the agent would be handed (normalised per TE:112-118 when the environment normalises)
      float o[2][8], a[2][4];
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        float4 c = core[l];
        float2 m = lam[l];
        if (V::NORM) normalise_row(c, m, P);
        o[l][0] = c.x; o[l][1] = c.y; o[l][2] = c.z; o[l][3] = c.w; o[l][4] = m.x; o[l][5] = m.y; o[l][6] = 0.f; o[l][7] = 0.f;
      }
      if (LP->is_linear) {
        linear_forward(*LP, o[0], a[0]);
        linear_forward(*LP, o[1], a[1]);
      } else {
        // (the network lives in LDS, staged before the loop; the tile loop streams its fragments from there - policy_mlp.hpp)
        mlp_forward_wave(policy_lds + kMlpLdsWaveBytesPerBlock, *LP, o, a, policy_lds + (threadIdx.x >> 6) * kMlpLdsBytesPerWave);
      }
      explore_and_clip(*LP, pair, P.philox_step + k, P.key0, P.key1, a);
#pragma unroll
      for (int l = 0; l < 2; ++l) act[l] = make_float4(a[l][0], a[l][1], A == 4 ? a[l][2] : 0.f, A == 4 ? a[l][3] : 0.f);
      philox_pair_noise(pair, P.philox_step + k, P.key0, P.key1, nz[0], nz[1]);  // after the network: the draws are not live across it
#endif
    } else if (R.policy == kPolicyFixed) {
      act[0] = act[1] = make_float4(R.action[0], R.action[1], R.action[2], R.action[3]);
    } else if (R.policy == kPolicyBuffer) {
      act[0] = held[0];
      act[1] = held[1];
    } else if (R.policy == kPolicyTimeTable) {  // open-loop schedule over time steps
      const float* row = reinterpret_cast<const float*>(R.table) + static_cast<size_t>(min(R.table_row0 + k, R.table_rows - 1u)) * A;
      act[0] = make_float4(row[0], row[1], A == 4 ? row[2] : 0.f, A == 4 ? row[3] : 0.f);
      settle_load(act[0]);
      act[1] = act[0];
    } else if (R.policy == kPolicyTable) {  // quotes tabulated over (time step, inventory), e.g. Cartea-Jaimungal
      const uint32_t row = min(R.table_row0 + k, R.table_rows - 1u);
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int col = min(max(static_cast<int>(core[l].y) + R.table_q_offset, 0), static_cast<int>(R.table_cols) - 1);
        const float2 d = R.table[static_cast<size_t>(row) * R.table_cols + col];
        act[l] = make_float4(d.x, d.y, 0.f, 0.f);
        settle_load(act[l]);
      }
    } else {  // Avellaneda-Stoikov quotes from (inventory, time) of the current observation
      const float tau = static_cast<float>(R.terminal_time - t);
      const float half = 0.5f * (R.as_c1 * tau + R.as_c2);
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const float shift = core[l].y * R.as_c1 * tau;
        act[l] = make_float4(shift + half, -shift + half, 0.f, 0.f);
      }
    }
    const double t_now = t;
    t += R.dt_f64;
    const bool terminal = (k + 1 == R.n_steps) && R.last_is_terminal != 0;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      // (Advancing the pair TOGETHER on 2-vectors - v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 for everything after the Bernoulli
      // decisions - was built and measured in round 3: 15 fewer arithmetic instructions per pair and step, 23 more v_mov to put
      // lane values into adjacent registers and SGPR spills from the longer live ranges: 3.48e11 instead of 3.62e11 env-steps/s at
      // 2^20 lanes.  profiles/r03_experiments.txt.  The compiler already packs the two SIDES of a lane where that is free.)
      const LaneResult r = V::PRECISE ? lane_step_exact<V>(core[l], lam[l], lo[l], act[l], make_draw<V>(nz[l], P), qi[l], terminal, P, nz[l].z, t_now, t, zu[l])
                                      : lane_step<V>(core[l], lam[l], act[l], make_draw<V>(nz[l], P), qi[l], static_cast<float>(t), terminal, P, nz[l].z, t_now, zu[l], HostStep{}, t, lo[l]);
      core[l] = r.core;
      lam[l] = r.lam;
      lo[l] = r.lo;
      ret[l] += r.reward;
      wave_clips += static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(real[l] && (r.clipped_q | r.clipped_c))));
      last_reward[l] = r.reward;
      if (B.events != nullptr) last_events[l] = event_byte(r);
      // (ONE scalar branch per lane and step in the returns-only rollout, which is bound by instruction issue, instead of three pointer
      // tests: 1.08 -> 1.04 us per step at 2^18 lanes, profiles/r06_mb_rollout.txt.  What round 5 suspected of the RECORDED rollout at
      // 2^18 lanes - six stores issued back to back stall a wave that has only one neighbour on its SIMD - was built and measured this
      // round: the step's rows kept in registers and written one store after every third Philox round of the NEXT step, 1.32-1.36 ->
      // 1.39-1.40 us per step.  Rejected; same file.)
      if (!recording) continue;
      if (R.obs_traj != nullptr) {
        float* slice = R.obs_traj + static_cast<size_t>(k + 1) * n_pad * V::DIM;
        if (V::PRECISE) store_row_exact<V, kStoreRecord>(slice, lanes[l], core[l], lam[l], lo[l], t, P);
        else store_row<V, kStoreRecord>(slice, lanes[l], core[l], lam[l], V::NORM, P);
      }
      if (R.act_traj != nullptr) {
        float* dst = R.act_traj + static_cast<size_t>(k) * n_pad * A;
        if (A == 2) store_as<kStoreRecord>(reinterpret_cast<float2*>(dst) + lanes[l], make_float2(act[l].x, act[l].y));
        else store_as<kStoreRecord>(reinterpret_cast<float4*>(dst) + lanes[l], act[l]);
      }
      if (R.rew_traj != nullptr) store_as<kStoreRecord>(R.rew_traj + static_cast<size_t>(k) * n_pad + lanes[l], r.reward);
    }
  }
  float ret_sum = 0.0f;
#pragma unroll
  for (int l = 0; l < 2; ++l) {  // what step() leaves behind: final state, last rewards (and events) of the final step
    store_row<V>(B.state_out, lanes[l], core[l], lam[l], false, P);
    if (V::RES != 0) store_lo<V>(B.resid, lanes[l], lo[l]);
    if (V::NORM && B.obs != nullptr) {
      if (V::PRECISE) store_row_exact<V>(B.obs, lanes[l], core[l], lam[l], lo[l], t, P);
      else store_row<V>(B.obs, lanes[l], core[l], lam[l], true, P);
    }
    if (R.n_steps > 0) {
      B.reward[lanes[l]] = last_reward[l];
      if (B.events != nullptr) B.events[lanes[l]] = static_cast<uint8_t>(last_events[l]);
    }
    if (B.lane_returns != nullptr) B.lane_returns[lanes[l]] += ret[l];
    ret_sum += lanes[l] < P.n ? ret[l] : 0.0f;
  }
  const float total = wave_sum(ret_sum);
  if ((threadIdx.x & 63u) == 0u) {
    const uint32_t wave = blockIdx.x * (kBlockThreads / 64) + (threadIdx.x >> 6);
    unsafeAtomicAdd(&B.wave_sums[wave], static_cast<double>(total));
    if (__builtin_expect(wave_clips != 0u, 0)) atomicAdd(&B.clip_count[wave & (kClipSlots - 1u)], static_cast<unsigned long long>(wave_clips));
  }
}

template <class V>
__global__ __launch_bounds__(kBlockThreads) void rollout_kernel(const StepBuffers B, const StepParams P, const RolloutParams R) {
  rollout_body<V>(B, P, R);
}

#ifndef MBT_JIT_USER_CODE
template <class V>
__global__ __launch_bounds__(kBlockThreads, 4) void learned_rollout_kernel(const StepBuffers B, const StepParams P, const RolloutParams R,
                                                                         const LearnedPolicyParams LP) {
  rollout_body<V, true>(B, P, R, &LP);
}

#ifndef MBT_KERNEL_TU  // (a non-template kernel: defined once, in the host translation unit - see kernel_table.hpp)
// The same policy as a kernel of its own, for a step loop: observation buffer (n_pad, D) -> action buffer (n_pad, A), in
// the step kernel's lane <-> thread mapping (so the wave-level MLP sees the same rows in the same places as the rollout).
__global__ __launch_bounds__(kBlockThreads) void policy_kernel(const float* obs, float* action, int dim, int act_dim, const LearnedPolicyParams LP,
                                                              uint64_t pair_offset, uint32_t philox_step, uint32_t key0, uint32_t key1) {
  constexpr int kMlpLdsWaveBytesPerBlock = (kBlockThreads / 64) * kMlpLdsBytesPerWave;
  __shared__ __attribute__((aligned(16))) char policy_lds[kMlpLdsWaveBytesPerBlock + kMlpLdsWeightBytes];
  const uint32_t lanes[2] = {blockIdx.x * kTileLanes + threadIdx.x, blockIdx.x * kTileLanes + threadIdx.x + kBlockThreads};
  float o[2][8], a[2][4];
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const float* row = obs + static_cast<size_t>(lanes[l]) * dim;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[l][c] = c < dim ? row[c] : 0.0f;
  }
  if (LP.is_linear) {
    linear_forward(LP, o[0], a[0]);
    linear_forward(LP, o[1], a[1]);
  } else {
    stage_mlp_weights(LP.w, policy_lds + kMlpLdsWaveBytesPerBlock);
    mlp_forward_wave(policy_lds + kMlpLdsWaveBytesPerBlock, LP, o, a, policy_lds + (threadIdx.x >> 6) * kMlpLdsBytesPerWave);
  }
  // exploration noise of the step that is about to be taken: the same (pair, philox step) the fused rollout would use
  explore_and_clip(LP, pair_offset + blockIdx.x * kBlockThreads + threadIdx.x, philox_step, key0, key1, a);
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    float* row = action + static_cast<size_t>(lanes[l]) * act_dim;
    for (int c = 0; c < act_dim; ++c) row[c] = a[l][c];
  }
}
#endif  // MBT_KERNEL_TU
#endif

// (neither the run-time compiled translation unit nor the translation units that only instantiate step / rollout kernels
// - csrc/kernels_*.hip - need what follows: the helper kernels are defined once, in mbt_env.hip)
#if !defined(MBT_JIT_USER_CODE) && !defined(MBT_KERNEL_TU)
// ---- small helper kernels ----------------------------------------------------------------------------------

// reset (TE:131-140): rows [initial_cash, q0, start_time, initial_price, process columns...] (reset_lane), zeroed accumulators.
__global__ void reset_kernel(float* state, float* obs, float* lane_returns, double* wave_sums, const float* q0, const ResetRow row0,
                             uint32_t n_pad, uint32_t n_waves, int dim, const StepParams P, int32_t* resid) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_waves) wave_sums[i] = 0.0;
  if (i >= n_pad) return;
  reset_lane(i, state, obs, lane_returns, q0, row0, dim, P, resid);
}

// Graph-capturable stepping: in front of the first step of every capture, and of every step outside one - the current clock slot
// becomes slot 0, so that the first launch behind it reads slot 0 (parity 0) whatever ran before.  One thread.
__global__ void captured_align_kernel(DeviceClock* clock) {
  if (clock->current != 0u) {
    clock->slot[0] = clock->slot[1];
    clock->current = 0u;
  }
}

// x ** p of the float32 tier on its own (mbt_power_f32_device): what tests measure power_f32's half-ulp claim on
__global__ __launch_bounds__(kBlockThreads) void power_f32_kernel(const float* x, double p, float* out, uint32_t n) {
  const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
  if (i < n) out[i] = power_f32(x[i], p);
}

// MEASUREMENT (mbt_env_record_floor_device): the fused rollout's recording and nothing else - the same lane <-> thread mapping, the same
// time-major slices, the same store instructions (kStoreRecord), 28 B per lane and step for D = 4, A = 2, no arithmetic worth the name -
// so that bench.py can time the write-only floor of a recording in the SAME process, against the SAME buffers, as the rollout it compares
// with it (the floor moves by +-15 % with where an allocation lands: a figure from another process is another allocation).
__global__ __launch_bounds__(kBlockThreads) void record_floor_kernel(float* obs_traj, float* act_traj, float* rew_traj, uint32_t n_pad, uint32_t steps, int dim, int act_dim) {
  const uint32_t lanes[2] = {blockIdx.x * kTileLanes + threadIdx.x, blockIdx.x * kTileLanes + threadIdx.x + kBlockThreads};
  float4 row[2] = {make_float4(static_cast<float>(lanes[0]), 1.f, 0.f, 100.f), make_float4(static_cast<float>(lanes[1]), -1.f, 0.f, 100.f)};
  for (uint32_t k = 0; k < steps; ++k) {
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      row[l].z = static_cast<float>(k);
      row[l].x += 1.0f;
      if (obs_traj != nullptr && dim == 4) store_as<kStoreRecord>(reinterpret_cast<float4*>(obs_traj + static_cast<size_t>(k + 1) * n_pad * 4) + lanes[l], row[l]);
      if (act_traj != nullptr) {
        float* dst = act_traj + static_cast<size_t>(k) * n_pad * act_dim;
        if (act_dim == 2) store_as<kStoreRecord>(reinterpret_cast<float2*>(dst) + lanes[l], make_float2(row[l].w, row[l].x));
        else store_as<kStoreRecord>(reinterpret_cast<float4*>(dst) + lanes[l], row[l]);
      }
      if (rew_traj != nullptr) store_as<kStoreRecord>(rew_traj + static_cast<size_t>(k) * n_pad + lanes[l], row[l].x);
    }
  }
}

// Small batches over the host API for kernels WITHOUT a mirror instantiation (injected-noise mode): observation rows and rewards
// copied into pinned, device-mapped host memory by a second launch (instead of two DMA copies, ~25 us each at any size).
__global__ void export_step_kernel(const float* obs, const float* reward, float* host_obs, float* host_reward, uint32_t n_obs, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (host_obs != nullptr && i < n_obs) host_obs[i] = obs[i];
  if (host_reward != nullptr && i < n) host_reward[i] = reward[i];
}

// Holds the environment's stream until the host writes `value` to a word of device-mapped host memory - so that a burst of
// launches can be ENQUEUED behind it and then run back to back whatever the host's cost per launch is (a tracer's, say:
// mbt_env_set_launch_gate).  Gives up by itself after `timeout_ticks` of the 100 MHz wall clock: a host that died cannot
// leave the device spinning.
__global__ void gate_kernel(const uint32_t* flag, uint32_t value, uint64_t timeout_ticks) {
  const uint64_t t0 = wall_clock64();
  // (the host counts bursts up and may already have opened LATER gates when this kernel gets to run: at-or-past, not equal)
  while (static_cast<int32_t>(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
    if (wall_clock64() - t0 > timeout_ticks) break;
    __builtin_amdgcn_s_sleep(64);
  }
}

// ---- host-callback plugins (Variant::HOST): the device side of what surrounds the user's NumPy code ---------------------
// The depths the user's _get_fill_probabilities(depths) is asked about (TE:104 + MD `_limit_depths`): the first two action
// columns, de-normalised in double exactly as decide<V>() de-normalises them (TE:124) - float64 (n, 2), for the host.
__global__ void host_depths_kernel(const float* action, int act_dim, uint32_t n, const StepParams P, double* depths) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int side = 0; side < 2; ++side) {
    const double a = action[static_cast<size_t>(i) * act_dim + side];
    depths[static_cast<size_t>(i) * 2 + side] = P.norm_act ? (a + 1.0) * P.act_grad[side] + P.act_lo[side] : a;
  }
}

// The state columns host-callback processes OWN (SP:8-53; MBT_MID_HOST: from the midprice column on), after their update() ran on the host: float64 (n, d) values into
// the state row (their float32 rounding; with precise_state the int32 remainder too, so state64 hands the user's own values
// back) and into the normalised observation row, the way TE:206-211 copies process.current_state into the state matrix.
__global__ void host_columns_kernel(const double* columns, uint32_t n, int d, int dim, int first, float* state, int32_t* resid, int res, int speed,
                                    float* obs, const StepParams P) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int j = 0; j < d; ++j) {
    const double x = columns[static_cast<size_t>(i) * d + j];
    float hi = static_cast<float>(x);
    // remainder columns: order book [cash, midprice (, x0, x1)] - state column 3 + k <-> remainder 1 + k; speed dynamics [cash,
    // inventory, midprice, impact state] - state column 3 + k <-> remainder 2 + k
    const int slot = (first - (speed ? 1 : 2)) + j;
    if (resid != nullptr && slot < res) {
      int32_t lo;
      exact_split(x, hi, lo);
      resid[static_cast<size_t>(i) * res + slot] = lo;
    }
    state[static_cast<size_t>(i) * dim + first + j] = hi;
    if (obs != nullptr) obs[static_cast<size_t>(i) * dim + first + j] = !P.norm_obs ? hi : (resid != nullptr ? normalise_column_exact(x, first + j, P) : normalise_column(hi, first + j, P));
  }
}

// The rewards the user's calculate() returned for the step that just ran (float64, host-computed from the float64 states):
// scaled (TE:128-129), rounded once to float32 like every reward this library hands out, filed where the step kernel would
// have filed its own - the reward buffer, the per-lane returns, the running sums behind the episode statistics.
// `replace`: the step kernel has no host-reward form (speed dynamics) and filed its own reward already - the difference goes in.
__global__ void host_reward_kernel(const double* rewards, double scale, uint32_t n, float* reward, float* lane_returns, double* wave_sums,
                                   uint32_t n_waves, int replace) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const float r = i < n ? static_cast<float>(scale * rewards[i]) : 0.0f;
  float add = r;
  if (i < n) {
    if (replace) add = r - reward[i];
    reward[i] = r;
    if (lane_returns != nullptr) lane_returns[i] += add;
  }
  const float total = wave_sum(add);
  if ((threadIdx.x & 63u) == 0u) unsafeAtomicAdd(&wave_sums[(i >> 6) % n_waves], static_cast<double>(total));
}

// un-normalised state rows -> normalised observation rows (after set_state)
// (`exact`: the precise_state tier normalises in double, like the reference; a state set from float32 rows has no remainders)
__global__ void normalise_rows_kernel(const float* state, float* obs, uint32_t n_pad, int dim, const StepParams P, int exact) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  const float* r = state + static_cast<size_t>(i) * dim;
  float* o = obs + static_cast<size_t>(i) * dim;
  for (int j = 0; j < dim; ++j) o[j] = !P.norm_obs ? r[j] : (exact ? normalise_column_exact(r[j], j, P) : normalise_column(r[j], j, P));
}

// [sum of wave_sums, sum of lane_returns^2 (NaN when per-lane returns are not tracked), lane count] -> out[0..2]; one block.
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
    out[1] = lane_returns != nullptr ? s_sq[0] : __builtin_nan("");
    out[2] = static_cast<double>(n);
  }
}

// RewardFunction.calculate on caller-supplied state matrices (RW:23-33, RW:96-109, RW:128-138), in DOUBLE and in
// the reference's order of operations, so host code that calls `calculate()` on stored trajectories (and the
// reference's own unit tests) gets the reference's float64 values without a CPU implementation.
__global__ void reward_calculate_kernel(int kind, const double* cur, const double* nxt, int dim, uint32_t n, int is_terminal,
                                        double phi, double alpha, double p, const double* q_init, const double* episode_length,
                                        const double* action, double risk_aversion, double* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* c = cur + static_cast<size_t>(i) * dim;
  const double* x = nxt + static_cast<size_t>(i) * dim;
  const double pnl = (x[0] + x[1] * x[3]) - (c[0] + c[1] * c[3]);
  double r = pnl;
  if (kind == kRewExpUtility) {  // RW:156-163
    r = is_terminal ? -exp(-risk_aversion * (x[0] + x[1] * x[3])) : 0.0;
  } else if (kind == kRewCjOe) {  // RW:57-70
    const double dt = x[2] - c[2];
    const double qp = (p == 2.0) ? x[1] * x[1] : pow(x[1], p);
    const double qpm1 = (p == 2.0) ? c[1] : pow(c[1], p - 1.0);
    const double qip = (p == 2.0) ? q_init[i] * q_init[i] : pow(q_init[i], p);
    r = pnl - dt * phi * qp - dt * alpha * (p * action[i] * qpm1 + qip * episode_length[i]);
  } else if (kind != kRewPnl) {
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

// StochasticProcessModel.update / ArrivalModel.get_arrivals / FillProbabilityModel.get_fills for HOST callers of the plugin
// objects outside an environment (SP:33-35, ARR:27-29, FILL:28-34): the arithmetic of one call on caller-supplied float64
// arrays, in the reference's order of operations - so a process object seeded like the reference's (its NumPy generator
// supplies the draws on the host, as in the reference) walks the reference's path without a CPU implementation of the maths.
enum : int { kProcessMidpriceUpdate = 0, kProcessHawkesUpdate = 1, kProcessArrivals = 2, kProcessFills = 3 };
__global__ void process_evaluate_kernel(int op, int arrival_kind, int fill_kind, const PreciseParams X, double thr_bid, double thr_ask, double kappa,
                                        double exo_bid, double exo_ask, double exo_base, const double* a, const double* b, const double* c, const double* d,
                                        uint32_t n, double* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (op == kProcessMidpriceUpdate) {  // a: S (n), b: z (n), c / d: the agent's bid / ask fills (n) or null
    out[i] = midprice_step_exact(a[i], b[i], c != nullptr ? c[i] : 0.0, d != nullptr ? d[i] : 0.0, X);
    return;
  }
  for (int side = 0; side < 2; ++side) {
    const size_t j = static_cast<size_t>(i) * 2 + side;
    if (op == kProcessHawkesUpdate) {  // a: intensities (n, 2), b: arrivals (n, 2) as 0 / 1   (ARR:110-119)
      const double base = side == 0 ? X.hawkes_base_bid : X.hawkes_base_ask;
      out[j] = (a[j] + X.hawkes_speed * (base - a[j]) * X.arr_dt) + X.hawkes_jump * b[j];
    } else if (op == kProcessArrivals) {  // a: uniforms (n, 2), b: intensities (n, 2) for Hawkes   (ARR:56, ARR:83, ARR:123)
      const double threshold = arrival_kind == kArrHawkes ? b[j] * X.arr_dt : (side == 0 ? thr_bid : thr_ask);
      out[j] = a[j] < threshold ? 1.0 : 0.0;
    } else {  // fills - a: uniforms (n, 2), b: depths (n, 2)   (FILL:34, FILL:57-58, FILL:159-163)
      const double best = side == 0 ? exo_bid : exo_ask;
      const double p = fill_kind == 2 ? (b[j] > best ? exo_base * exp(-kappa * (b[j] - best)) : 1.0) : exp(-kappa * b[j]);
      out[j] = a[j] < p ? 1.0 : 0.0;
    }
  }
}

// The production noise, written out in the step kernel's own lane <-> pair mapping (tests pin the generator and tie
// Philox mode to injected mode with it).  One 256-thread block per tile.
__global__ void rng_fill_kernel(uint64_t pair_offset, uint32_t step, uint32_t k0, uint32_t k1, float* u_arr, float* u_fill, float* z) {
  const uint32_t lane0 = blockIdx.x * kTileLanes + threadIdx.x, lane1 = lane0 + kBlockThreads;
  LaneNoise a, b;
  philox_pair_noise(pair_offset + blockIdx.x * kBlockThreads + threadIdx.x, step, k0, k1, a, b);
  if (u_arr != nullptr) {
    reinterpret_cast<float2*>(u_arr)[lane0] = make_float2(a.ua_bid, a.ua_ask);
    reinterpret_cast<float2*>(u_arr)[lane1] = make_float2(b.ua_bid, b.ua_ask);
  }
  if (u_fill != nullptr) {
    reinterpret_cast<float2*>(u_fill)[lane0] = make_float2(a.uf_bid, a.uf_ask);
    reinterpret_cast<float2*>(u_fill)[lane1] = make_float2(b.uf_bid, b.uf_ask);
  }
  if (z != nullptr) {
    z[lane0] = a.z;
    z[lane1] = b.z;
  }
}

// the extra normals of user processes (Variant::USER_DRAWS), in the step kernel's lane <-> pair mapping: z_user (n_pad, 2)
__global__ void rng_fill_user_kernel(uint64_t pair_offset, uint32_t step, uint32_t k0, uint32_t k1, float* z_user) {
  const uint32_t lane0 = blockIdx.x * kTileLanes + threadIdx.x, lane1 = lane0 + kBlockThreads;
  float2 a, b;
  philox_pair_user_noise(pair_offset + blockIdx.x * kBlockThreads + threadIdx.x, step, k0, k1, a.x, a.y, b.x, b.y);
  reinterpret_cast<float2*>(z_user)[lane0] = a;
  reinterpret_cast<float2*>(z_user)[lane1] = b;
}

__global__ void philox_kat_kernel(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
  const PhiloxWords w = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
  out[0] = w.w0; out[1] = w.w1; out[2] = w.w2; out[3] = w.w3;
}

#endif  // !MBT_JIT_USER_CODE && !MBT_KERNEL_TU

}  // namespace mbt
