/*
 * mbt_env.h - C ABI of libmbtenv: the MI355X-native (gfx950) TradingEnvironment.step() hot path.
 *
 * The reference (JJJerome/mbt_gym) is pure Python/NumPy and has no FFI; the boundary it offers is the
 * Python duck type gym.Env / StochasticProcessModel / RewardFunction.  This header is the C boundary a
 * maintainer would bind (ctypes) underneath those Python classes.  Every entry point
 * names the reference code it replaces (paths relative to /root/reference/mbt_gym):
 *   TE   gym/TradingEnvironment.py      MD   gym/ModelDynamics.py     RW  rewards/RewardFunctions.py
 *   MID  stochastic_processes/midprice_models.py    ARR  stochastic_processes/arrival_models.py
 *   FILL stochastic_processes/fill_probability_models.py   SP  stochastic_processes/StochasticProcessModel.py
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary; every call returns 0 or a negative
 *     mbt_status and leaves a message for mbt_last_error() (thread local).
 *   - the library owns all device memory of an environment; callers own every pointer they pass.
 *   - "*_host" entry points take any host pointer (pinned memory - mbt_host_alloc, or the caller's own - is read and
 *     written by DMA directly, pageable memory through a bounce buffer) and are synchronous (NumPy semantics);
 *     "*_device" entry points take device pointers (or NULL = the library's own buffers), only enqueue
 *     work on the environment's HIP stream and return immediately.
 *   - one host thread per environment handle at a time.
 *   - state rows are float32 [cash, inventory, time, midprice] (index_names.py:1-4); a Hawkes arrival
 *     model adds [bid intensity, ask intensity], a price-impact model with state adds [impact] (TE:311-318).
 *     Side 0 = bid, 1 = ask (index_names.py:6-7).
 *   - there is NO CPU fallback: without a gfx950 device mbt_env_create fails with MBT_ERR_NO_DEVICE.
 */
#ifndef MBT_ENV_H
#define MBT_ENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MBT_ABI_VERSION 8u

typedef enum mbt_status {
  MBT_OK = 0,
  MBT_ERR_INVALID = -1,     /* bad argument / unsupported plugin combination */
  MBT_ERR_NO_DEVICE = -2,   /* no HIP device, or device is not gfx950 */
  MBT_ERR_HIP = -3,         /* a HIP runtime call failed */
  MBT_ERR_STATE = -4,       /* call order violated (e.g. step before reset, noise not set) */
  MBT_ERR_ABI = -5          /* mbt_config.abi_version mismatch */
} mbt_status;

/* plugin kinds: the reference classes that have a device implementation
 * (IMP = stochastic_processes/price_impact_models.py) */
enum {
  MBT_MID_BROWNIAN = 0 /* MID:36-68 */, MBT_MID_OU = 1 /* MID:114-146 */, MBT_MID_GBM = 2 /* MID:71-111 */,
  MBT_MID_BROWNIAN_JUMP = 3 /* MID:193-230 */, MBT_MID_OU_JUMP = 4 /* MID:233-273 */, MBT_MID_CONSTANT = 5 /* MID:12-33 */,
  /* The family all of the above are members of, for user-defined MidpriceModel subclasses (SP:8-53) that are linear
   * SDEs - one Euler step of
   *   dS = (mid_coef_add + mid_coef_mul * S) * (drift * dt + volatility * sqrt(dt) * Z) - ou_speed * (S - ou_level)
   *        + jump_size * (ask fills - bid fills)
   * e.g. CEV-free local-volatility mixtures, drifting OU, GBM with trade impact.  The built-in kinds are the rows
   * (1,0,theta=0) BM, (1,0,theta) OU [drift forced to 0], (0,1) GBM, +jump_size for the jump variants, (0,0) constant. */
  MBT_MID_LINEAR_SDE = 6,
  MBT_MID_USER = 7 /* a user-defined one-column MidpriceModel subclass (SP:8-53) with an arbitrary increment: mbt_env_create_jit only */,
  MBT_MID_HOST = 8 /* a MidpriceModel subclass that only has HOST code: update() runs in the caller AFTER the launch, see "host-callback
                    * plugins" (needs MBT_REW_HOST: the reward of a step holds the midprice the caller is about to compute) */
};
enum {
  MBT_ARR_POISSON = 0 /* ARR:32-56 */, MBT_ARR_HAWKES = 1 /* ARR:86-126 */, MBT_ARR_POISSON_NONLINEAR = 2 /* ARR:59-83 */,
  MBT_ARR_NONE = 3 /* speed dynamics: no order flow (MD:47-48) */,
  MBT_ARR_USER = 4 /* a user-defined, stateless ArrivalModel subclass (ARR:9-29): mbt_env_create_jit only */,
  MBT_ARR_HOST = 5 /* a stateless ArrivalModel subclass that only has HOST code: get_arrivals() runs in the caller, see "host-callback plugins" */
};
enum {
  MBT_FILL_EXPONENTIAL = 0 /* FILL:42-65 */, MBT_FILL_NONE = 1 /* at-the-touch and speed dynamics */,
  MBT_FILL_EXOGENOUS_MM = 2 /* FILL:126-170: certain inside an exogenous best depth, exponential beyond; adds two state columns */,
  MBT_FILL_USER = 3 /* a user-defined FillProbabilityModel subclass (FILL:9-39): mbt_env_create_jit only */,
  MBT_FILL_HOST = 4 /* a FillProbabilityModel subclass that only has HOST code: _get_fill_probabilities runs in the caller */
};
enum {
  MBT_DYN_LIMIT = 0 /* MD:87-131 */, MBT_DYN_LIMIT_AND_MARKET = 1 /* MD:179-240 */, MBT_DYN_AT_THE_TOUCH = 2 /* MD:134-176 */,
  MBT_DYN_SPEED = 3 /* MD:243-275 */
};
enum {
  MBT_REW_PNL = 0 /* RW:20-36 */, MBT_REW_RUNNING_PENALTY = 1 /* RW:116-143 */, MBT_REW_CJ_MM = 2 /* RW:77-113 */,
  MBT_REW_EXP_UTILITY = 3 /* RW:149-163 */, MBT_REW_CJ_OE = 4 /* RW:39-74, speed dynamics */,
  MBT_REW_USER = 5 /* a user-defined RewardFunction subclass (RW:8-17): mbt_env_create_jit only */,
  MBT_REW_HOST = 6 /* a RewardFunction subclass that only has HOST code: calculate() runs in the caller (every dynamics kind) */
};
enum {
  MBT_IMPACT_NONE = -1, MBT_IMPACT_TEMPORARY_POWER = 0 /* IMP:34-61 */, MBT_IMPACT_TEMPORARY_AND_PERMANENT = 1 /* IMP:64-96 */,
  MBT_IMPACT_TEMPORARY_AND_TRANSIENT = 2 /* IMP:99-139 */, MBT_IMPACT_TRANSIENT = 3 /* IMP:142-179 */,
  /* a PriceImpactModel subclass (IMP:9-31) that only has HOST code: get_impact(action) runs in the caller before every step
   * (mbt_env_set_host_impacts), see "host-callback plugins"; _STATE: the model owns the state column y (initial value
   * initial_transient_impact), advanced by its own update() on the host (mbt_env_set_host_state_columns).  precise_state only. */
  MBT_IMPACT_HOST = 4, MBT_IMPACT_HOST_STATE = 5
};
enum {
  MBT_NOISE_PHILOX = 0,   /* counter-based Philox4x32-10 drawn inside the kernel (production) */
  MBT_NOISE_INJECTED = 1  /* noise supplied by mbt_env_set_noise_* (parity tests vs. the reference) */
};

/* Everything the constructors of TradingEnvironment (TE:27-94), the ModelDynamics (MD:18-40) and the plugin
 * classes hold, flattened.  Plain data; versioned by abi_version. */
typedef struct mbt_config {
  uint32_t abi_version;        /* MBT_ABI_VERSION */
  int32_t device;              /* HIP device ordinal */
  uint64_t num_trajectories;   /* lanes owned by this handle (TE:41) */
  uint64_t trajectory_offset;  /* global id of lane 0 when the trajectory axis is sharded; a multiple of 512 (1024 for speed dynamics) */
  uint32_t n_steps;            /* TE:30 */
  uint32_t reserved0;
  double terminal_time;        /* TE:29; step_size = terminal_time / n_steps (TE:49) */

  int32_t midprice_kind;
  int32_t arrival_kind;
  int32_t fill_kind;
  int32_t dynamics_kind;
  int32_t reward_kind;
  int32_t noise_mode;

  double drift, volatility, initial_price;   /* MID:39-41 / MID:117-120 */
  double ou_level, ou_speed;                 /* MID:117-118 (speed is NOT multiplied by dt, MID:140-143) */
  double intensity[2];                       /* Poisson rate (ARR:35) or Hawkes baseline (ARR:89), bid/ask */
  double hawkes_jump, hawkes_speed;          /* ARR:91-92 */
  double fill_exponent;                      /* FILL:44 */
  double market_half_spread;                 /* MD:189 */
  double phi, alpha, inventory_exponent;     /* RW:119-122 / RW:83-86 */
  double initial_cash;                       /* TE:33 */
  double initial_inventory;                  /* TE:34, used by reset when no per-lane array is given */
  double max_inventory;                      /* TE:36 */
  double max_cash;                           /* TE:37, TE:229-230 */
  double reward_scale;                       /* TE:128-129; 1.0 when rewards are not normalised */

  uint64_t seed;                             /* Philox key */
  int32_t normalise_observation;             /* TE:44, TE:112-118 */
  int32_t normalise_action;                  /* TE:43, TE:120-126 */
  float obs_lo[8], obs_hi[8];                /* float32 Box bounds (TE:232-241) */
  float act_lo[4], act_hi[4];                /* float32 Box bounds (MD:118-121, MD:224-231, MD:269-271) */

  /* every process keeps its OWN step_size constructor argument (SP:21) and the environment never synchronises
   * them; 0 = terminal_time / n_steps */
  double midprice_step_size;                 /* MID:63-64; also the volume of speed dynamics (MD:265) */
  double arrival_step_size;                  /* ARR:56, ARR:83, ARR:115-123 */
  double jump_size;                          /* MID:199, MID:240 */
  double risk_aversion;                      /* RW:150 */
  int32_t impact_kind;                       /* MBT_IMPACT_* (speed dynamics) */
  int32_t reserved1;
  double temporary_impact, impact_exponent;  /* IMP:37-38 */
  double permanent_impact;                   /* IMP:68 */
  double transient_impact, resilience, initial_transient_impact, kernel_coefficient; /* IMP:103-106 */
  double impact_step_size;                   /* IMP:75: the impact model's own terminal_time / n_steps; 0 = env's */

  /* MBT_FILL_EXOGENOUS_MM: the (bid, ask) best depths = initial states of the two depth processes (FILL:148-154).
   * The reference never copies the processes' updates into the fill model's state (FILL:168-170), so the two state
   * columns and the depths of FILL:159-163 hold these values for the whole episode. */
  double exogenous_depth[2];
  double base_fill_probability;              /* FILL:132 */

  /* ---- ABI 4 (ABI 5 added entry points only) ---- */
  double reward_terminal_time;               /* CjMmCriterion / CjOeCriterion keep their OWN terminal_time (RW:88, RW:74, RW:113); 0 = terminal_time */
  double mid_coef_add, mid_coef_mul;         /* MBT_MID_LINEAR_SDE only */
  /* Numerics tier.
   * 0 (default): float32 state, the 44 B / env-step kernels.  Decisions (arrivals, fills, inventory) are bit-exact against
   *   the float64 reference on the same draws - Hawkes arrivals included: the intensities they are decided on are held exactly
   *   (hawkes_float32_intensities below; 76 B / env-step for Hawkes rows).  Rewards, with U = 2^-24
   *   (one float32 rounding) and S_err, c_err = |float32 state - reference state| of midprice and cash BEFORE the step:
   *       |r - r_ref| <= 1e-5 + 1e-6 max(|r|, |q' dS|)                       (the contract + float32's own output rounding)
   *                    + |q'| L (S_err + 2 U |S|)                            (L = 0 for Brownian / jump / constant midprices - the
   *                                                                           increment does not read the state; theta for OU;
   *                                                                           |dS / S| for GBM)
   *                    + [the clip of TE:283-289 fired] (4 (S_err + 2 U |S|) + c_err + 4 U |cash|)
   *   i.e. 1e-5-class wherever no float32 state LEVEL enters the reward, and exactly the state error marked to market where
   *   one does (a clip; a state-proportional increment).  The state columns themselves drift like a sum of independent
   *   roundings: <= 6 sqrt(k / 6) ulp32(|x|max) after k steps (six sigma), cash additionally by what it inherits from the
   *   midprice through - dq S.  tests/float32_tier_bounds.py is this formula, tests/test_gpu_random_configs.py asserts it.
   *   Real-valued inventory (speed dynamics) is float32 state too: its error shows times the price move, and times the
   *   price where the clip fires.
   * 1: the reference's float64 state, EXACTLY - every real-valued column (cash, midprice, Hawkes intensities, the inventory
   *   and impact state of speed dynamics) is its float32 rounding in the state row plus an int32 remainder in a side
   *   buffer (mbt_exact_split: the same 8 bytes as a float32 pair, all 53 bits) - stepped in double in the reference's own
   *   order of operations: state, rewards and every decision ARE the reference's on the same draws (rewards rounded once
   *   to float32; a transcendental in the path - a fractional impact exponent, exponential utility, a user expression -
   *   agrees to an ulp instead of to the bit).  Every plugin family, user-defined (mbt_env_create_jit) included.
   *   +8 B per remainder column and env-step: 60 B instead of 44 (limit orders), 92 instead of 60 (Hawkes), 80 instead of
   *   48 (speed dynamics with an impact state).  Observations are unchanged in layout: float32 rows. */
  int32_t precise_state;
  /* The Hawkes intensity recursion lambda += speed (base - lambda) dt + jump (ARR:110-119) is a contraction only for
   * hawkes_speed * arrival_step_size < 1; from 1 it oscillates and from 2 it diverges (in the float64 reference as well),
   * and float32 state no longer tracks float64 state.  mbt_env_create rejects such a configuration unless this is 1. */
  int32_t allow_stiff_hawkes;
  /* ---- ABI 7 ---- */
  /* Hawkes intensities in the float32 tier (precise_state == 0; ignored otherwise and without a Hawkes arrival model).
   * 0 (default): the two intensity columns - and only they - are held EXACTLY (float32 rounding in the state row, what the
   *   observation shows, + an int32 remainder each in a side buffer: mbt_exact_split), the recursion lambda += speed (base -
   *   lambda) dt + jump arrivals and the threshold lambda dt are evaluated in double in the reference's order (ARR:110-123): the
   *   arrivals - hence fills and inventories - ARE the float64 reference's on the same draws, like every other decision of this
   *   tier.  Cash and midprice stay float32 with the increment-form reward.  +16 B per env-step: 76 instead of 60.
   * 1: float32 intensities, 60 B per env-step - what SURVEY section 8d prices; a draw with |u - lambda dt| <= (2e-5 + 3e-7
   *   lambda) dt can then decide differently from the reference (probability 2 (2e-5 + 3e-7 lambda) dt per draw and side:
   *   ~1e-6 per lane-step at lambda ~ 40, dt ~ 1/40), after which that lane's path is another sample of the same process. */
  int32_t hawkes_float32_intensities;
  /* 1: small batches (up to 4096 lanes; float32 tier, built-in order-book models, production noise - ignored otherwise) step through a
   * RESIDENT kernel: see mbt_env_step_host.  The environment variable MBT_RESIDENT_STEP=1 does the same for every environment a process
   * creates.  Costs kernels on other streams of the device 20-27 % while it is there, hence 0 by default. */
  int32_t resident_step;
} mbt_config;

typedef struct mbt_env mbt_env; /* opaque: device state, buffers, stream */

/* ---- library ---------------------------------------------------------------------------------- */
uint32_t mbt_abi_version(void);
size_t mbt_config_sizeof(void); /* sizeof(mbt_config) as compiled, for binding self-checks */
/* sha256 (hex) of the sources this library was compiled from (every .hip / .hpp under csrc plus include/mbt_env.h, in
 * name order), baked in at build time: lets a binding detect a stale prebuilt library next to newer sources. */
const char* mbt_source_hash(void);
const char* mbt_last_error(void);
int mbt_device_count(void);
/* Writes the device name ("gfx950...") into buf; returns MBT_OK or an error. */
int mbt_device_name(int device, char* buf, size_t buf_len);

/* ---- pinned host memory for the "*_host" entry points ------------------------------------------ */
/* The "*_host" entry points move actions in and observations / rewards out with DMA copies, which read and write PINNED
 * host memory directly; pageable memory has to be staged (by the library through a pinned bounce buffer: one more pass
 * over the bytes, and page faults if the destination is fresh).  A caller that steps large batches through the host API
 * allocates its action / observation / reward buffers here once and re-uses them: the copies then run at the PCIe rate
 * (28 B per lane and step for the Avellaneda-Stoikov workload).  Every host pointer is accepted either way - the library
 * recognises these blocks (and foreign pinned memory, through hipPointerGetAttributes).  Returns NULL on failure. */
void* mbt_host_alloc(size_t bytes);
void mbt_host_free(void* ptr);

/* ---- lifetime --------------------------------------------------------------------------------- */
/* Replaces TradingEnvironment.__init__ (TE:27-94) for the numeric part: validates the plugin combination,
 * allocates the (N, D) state and the output buffers in HBM, creates the stream. */
int mbt_env_create(const mbt_config* cfg, mbt_env** out);
void mbt_env_destroy(mbt_env* env);

/* ---- user-defined plugins: the device route for subclasses of the reference's plugin contract -------------------
 * The reference composes arbitrary StochasticProcessModel / RewardFunction subclasses (SP:8-53, FILL:9-39, RW:8-17).
 * There is no CPU path to run their NumPy code, so a subclass states its numerics as a C++ device EXPRESSION and the
 * library compiles the step (and fused rollout) kernel around it at run time with hiprtc - the same kernel source as
 * the built-in tiers (general tier: any midprice kind incl. MBT_MID_LINEAR_SDE, Poisson / Hawkes arrivals, limit or
 * limit + market dynamics, normalisation), with the user's function inlined where the built-in one would be:
 *   fill_probability   replaces FillProbabilityModel._get_fill_probabilities (FILL:22-34) when cfg.fill_kind ==
 *                      MBT_FILL_USER: an expression of type double in `depth` (double: the de-normalised quote depth),
 *                      `side` (int: 0 bid, 1 ask) and the named parameters.  A fill happens when u < expression, with
 *                      u the lane's uniform, compared in double.
 *   midprice_increment replaces MidpriceModel.update (MID:60-65 and its siblings) when cfg.midprice_kind == MBT_MID_USER: an
 *                      expression of type double for S' - S in `S` (the midprice before the step), `t` (the time at the
 *                      beginning of the step), `z` (this lane's standard normal draw), `dt` (the midprice model's step
 *                      size), `fills_bid`, `fills_ask` (1.0 where the agent's bid / ask quote was filled this step,
 *                      MID:220-221) and the named parameters - e.g. CEV: "mu * S * dt + sigma * pow(S, gamma) * sqrt(dt) * z".
 *                      One state column, one normal per step; evaluated in double, the state itself is float32.
 *   arrival_probability replaces ArrivalModel.get_arrivals (ARR:27-29) when cfg.arrival_kind == MBT_ARR_USER: an expression
 *                      of type double in `t` (double: the time at the BEGINNING of the step, i.e. the time stamp of the
 *                      observation the agent acted on), `side` (0 = a sell order arriving at the bid, 1 = a buy order at
 *                      the ask, ARR:9-13), `dt` (the arrival model's step size) and the named parameters - e.g. a
 *                      time-of-day intensity profile.  An arrival happens when u < expression.  The model is stateless
 *                      (no state columns; what it needs of the past it must get from `t`).
 *   reward             replaces RewardFunction.calculate (RW:10-13) when cfg.reward_kind == MBT_REW_USER: an expression of
 *                      type double in  cash, q, t, mid  (current state)  cash_next, q_next, t_next, mid_next  (next state)
 *                      a0, a1, a2, a3 (the action as given)  pnl (the mark-to-market change, from the step's increments)
 *                      dt, is_terminal (1.0 on the terminal step), q0 (initial inventory), episode_length (T - t_start)
 *                      and the named parameters.  The value is multiplied by cfg.reward_scale and rounded to float32.
 * Parameter names are comma-separated C identifiers bound, in order, to the params arrays (at most 8 each).  Math
 * functions are HIP's device library in double (exp, log, pow, sqrt, fabs, fmin, fmax, tanh, ...).  Compilation takes
 * a few seconds per distinct (configuration, code) pair and is cached for the life of the process; diagnostics of a
 * failed compilation are returned by mbt_jit_log().  Everything else about the handle is as mbt_env_create makes it. */
typedef struct mbt_user_code {
  const char* fill_probability;   /* NULL unless cfg.fill_kind == MBT_FILL_USER */
  const char* fill_param_names;   /* e.g. "k,alpha"; NULL or "" = none */
  double fill_params[8];
  const char* reward;             /* NULL unless cfg.reward_kind == MBT_REW_USER */
  const char* reward_param_names;
  double reward_params[8];
  const char* arrival_probability; /* NULL unless cfg.arrival_kind == MBT_ARR_USER */
  const char* arrival_param_names;
  double arrival_params[8];
  const char* midprice_increment;  /* NULL unless cfg.midprice_kind == MBT_MID_USER */
  const char* midprice_param_names;
  double midprice_params[8];
  /* ---- ABI 5: user processes that OWN state (SP:8-53: a subclass carries its own (N, d) current_state) ----------------
   * state_columns = 1 or 2 adds that many columns x0 (, x1) to the state row right after the midprice, in the reference's
   * registry order (a second midprice factor first, then the arrival model's columns, TE:303-318); Poisson-type or user
   * arrivals and no exogenous-depth fill model (the columns take the place of the built-in Hawkes intensities).  Each
   * column is advanced by its own expression  state_update[j]  for the NEXT value, of type double, in
   *     S, t, dt (the step size of the process that owns the column, state_owner), x0, x1 (the columns BEFORE the step), z (the lane's midprice normal), z1, z2,
   *     arr_bid, arr_ask (1.0 where an order arrived), fills_bid, fills_ask (1.0 where the agent's quote was executed),
   *     S_next, t_next, q_next, cash_next (what the state matrix the reference hands update() holds at that point, TE:206-211:
   *     cash / inventory after the agent's update and the clip, the advanced clock, the midprice already advanced)
   * and the named state parameters - x0, x1, S, t as the step found them, like the process's own state in the reference's update().  The
   * midprice_increment and arrival_probability expressions may read x0, x1 (and z1, z2) too: a two-factor midprice
   * (dS = alpha dt + sigma dW, alpha its own OU process), a Hawkes variant with cross-excitation, a stochastic intensity.
   * extra_normals = 1 draws z1, z2 - two more standard normals per lane and step, a third Philox block per pair of lanes
   * (mbt_rng_fill_user_host exports them; injected-noise mode takes them through mbt_env_set_user_noise_host) - otherwise
   * they read 0.  state_initial: the columns after reset (SP:30-31); their observation bounds travel in cfg.obs_lo / obs_hi. */
  int32_t state_columns;
  int32_t extra_normals;
  const char* state_update[2];
  const char* state_param_names;
  double state_params[8];
  double state_initial[2];
  int32_t state_owner[2];  /* 0 = the midprice model's column (dt = its step size), 1 = the arrival model's, 2 = a host-callback fill model's (MBT_FILL_HOST: no update expression) */
} mbt_user_code;
int mbt_env_create_jit(const mbt_config* cfg, const mbt_user_code* code, mbt_env** out);
const char* mbt_jit_log(void);    /* thread local; "" when the last compilation had nothing to say */
/* Compiles the kernels for (cfg, code) and discards them: "do these expressions compile?" - needs no GPU (hiprtc
 * cross-compiles), so a plugin can be checked where it is written. */
int mbt_jit_check(const mbt_config* cfg, const mbt_user_code* code);
/* ---- host-callback plugins: subclasses of the plugin contract that only have HOST code ----------------------------------
 * A user of the reference writes FillProbabilityModel._get_fill_probabilities(depths) (FILL:22-34), ArrivalModel.get_arrivals()
 * (ARR:27-29) or RewardFunction.calculate(current_state, action, next_state, is_terminal_step) (RW:10-13) in NumPy.  Such a
 * class needs no device expression: with MBT_FILL_HOST / MBT_ARR_HOST / MBT_REW_HOST the caller's code keeps running on the
 * host BETWEEN launches and the step kernel takes its results for the step - everything else of the step (max-inventory mask,
 * cash / inventory, clip, midprice and Hawkes updates, normalisation, the draws themselves) stays on the device.  The slow
 * path: one or two extra host round trips per step.  mbt_env_create serves these kinds (the kernel is still compiled at run
 * time: one instantiation); they combine with built-in kinds and with device expressions of the OTHER families.  Per step:
 *   1. MBT_FILL_HOST:  mbt_env_host_depths(action) -> the (N, 2) float64 depths the quotes stand for (the action
 *      de-normalised in double exactly as the kernel does, TE:104 / TE:124); the caller evaluates its
 *      _get_fill_probabilities(depths) and hands the (N, 2) float64 result to mbt_env_set_host_fill_probabilities: a fill is
 *      u < p with u the lane's uniform (Philox, or injected), compared in double (FILL:33-34).
 *   2. MBT_ARR_HOST:  the caller's get_arrivals() (which draws from the caller's own generator, ARR:55) -> (N, 2) 0.0f / 1.0f
 *      -> mbt_env_set_host_arrivals.  A model that owns state columns keeps them on the host and files them after its update():
 *      mbt_env_set_host_state_columns.
 *   3. mbt_env_step_* as usual (refused with MBT_ERR_STATE if an input of this step is missing).
 *   4. MBT_REW_HOST:  the step reports rewards of 0; the caller reads the float64 states (mbt_env_get_state_f64_host - before
 *      and after the step; with precise_state they ARE the reference's float64 states), evaluates calculate() and files the
 *      (N) float64 result with mbt_env_set_host_rewards: multiplied by cfg.reward_scale (TE:128-129), rounded once to float32,
 *      written to the reward buffer, added to the episode-return sums; reward_out (N float32, may be NULL) receives what
 *      env.step() returns.  The next step is refused until this happened.  (Trading-with-speed dynamics: the speed kernels are
 *      built ahead of time and have no host-reward form - the step files the mark-to-market change and this call REPLACES it,
 *      in the reward buffer and in the return sums.)
 *   5. MBT_MID_HOST:  the kernel holds the midprice column(s) still; after the launch the caller's update(arrivals, fills, action,
 *      state) advances them on the host (MID: the reference calls it first of the processes, TE:206-211) and files them with
 *      mbt_env_set_host_state_columns - BEFORE step 4, whose `next_state` shows them: the reward function (built-in classes
 *      included: mbt_reward_calculate_host evaluates them on the float64 matrices) is then a host callback by construction, which
 *      is why MBT_MID_HOST is only accepted together with MBT_REW_HOST.  With trading-with-speed dynamics: the price column alone.
 *   6. MBT_IMPACT_HOST / MBT_IMPACT_HOST_STATE (trading-with-speed dynamics, precise_state): the caller's get_impact(action) on the
 *      de-normalised action (MD:263; float64 (N)) -> mbt_env_set_host_impacts before the step: execution price = midprice + that
 *      value.  A model that owns the impact-state column keeps it on the host and files it after its update() like the others.
 * Fused rollouts are not available for such an environment (MBT_ERR_INVALID): the host is consulted every step. */
int mbt_env_host_depths(mbt_env* env, const float* action_host, double* depths_host);
int mbt_env_set_host_fill_probabilities(mbt_env* env, const double* probabilities_host);
int mbt_env_set_host_arrivals(mbt_env* env, const float* arrivals_host);
int mbt_env_set_host_rewards(mbt_env* env, const double* rewards_host, float* reward_out_host);
/* What TE:206-211 hands a process's update() and RW:10-13 a reward's calculate() AFTER the step - the state matrix, the step's arrivals
 * and fills - without a device round trip.  Small batches (up to 65536 lanes), order-book dynamics, host-callback plugins: the step kernel also mirrors its un-normalised rows,
 * their int32 remainders and the event bytes into host memory, so what the caller's update() / calculate() are handed needs no
 * further round trip.  state_host: (N, D) float64 as mbt_env_get_state_f64_host assembles it (may be NULL); events_host: (N) bytes as
 * mbt_env_get_events_host returns them (may be NULL).  MBT_ERR_STATE when the last call on the environment was not such a step
 * (use the two functions named instead). */
int mbt_env_host_step_outputs(mbt_env* env, double* state_host, uint8_t* events_host);
int mbt_env_set_host_impacts(mbt_env* env, const double* impacts_host);
/* The state columns host-callback processes OWN (SP:8-53), ONE contiguous block in registry order (TE:303-318): with
 * MBT_MID_HOST the midprice column 3 and the midprice model's further columns (declared through mbt_user_code.state_columns /
 * state_initial / state_owner = 0 with NULL update expressions), then the columns of a host-callback arrival model that owns
 * state (state_owner = 1, NULL update expressions; mbt_env_create_jit), then a host-callback fill model's (state_owner = 2) or, with speed dynamics, the impact-state column of
 * MBT_IMPACT_HOST_STATE - d columns in all.  The kernel carries them through the
 * step unchanged; after the caller's update(arrivals, fills, action, state) calls ran, the new (N, d) float64 values are filed
 * with this call - their float32 rounding into the state row (TE:206-211: the reference copies process.current_state into the
 * state matrix), the int32 remainders too under precise_state, the normalised observation row if there is one.  Also valid
 * right after a reset (a process whose reset() sets per-lane initial values).  Follow it with mbt_env_get_obs_host for the
 * observation env.step() / env.reset() returns. */
int mbt_env_set_host_state_columns(mbt_env* env, const double* columns_host);

/* Every launch and copy of an environment is ordered on ONE stream: its own (created non-blocking, so it does not
 * synchronise with the null stream) until this call hands it another hipStream_t, e.g. torch's current stream.
 * mbt_env_create returns with all buffers allocated, zero-filled and idle.  A caller that reads or writes the device
 * buffers (mbt_env_*_device pointers) from a different stream orders the two itself - mbt_env_synchronize, or an
 * event - or shares its stream through this call. */
int mbt_env_set_stream(mbt_env* env, void* hip_stream);
/* Waits for everything enqueued on the environment's stream (polls for the first ~200 us, then blocks). */
int mbt_env_synchronize(mbt_env* env);
/* TradingEnvironment.step_size setter (TE:158-167): the clock, the done rule (TE:218-220) and EVERY process (each is
 * given the new value, TE:161-163) continue with `step_size`; n_steps, terminal_time and max_cash are untouched, exactly
 * like the reference.  Host-side parameters only: no reallocation, takes effect with the next step. */
int mbt_env_set_step_size(mbt_env* env, double step_size);

/* ---- seeding (TE:345-348, SP:37-39) ----------------------------------------------------------- */
/* Re-keys the Philox generator and restarts its step counter.  Like the reference, reset() does not reseed:
 * successive episodes continue the stream. */
int mbt_env_seed(mbt_env* env, uint64_t seed);

/* ---- reset (TE:96-101, TE:131-140, TE:257-281, RW:111-113) ------------------------------------ */
/* start_time must already be quantised to a step multiple (TE:266-268).  q0 is an optional per-lane initial
 * inventory array (N floats, host memory) - NULL means cfg.initial_inventory for every lane. */
int mbt_env_reset(mbt_env* env, double start_time, const float* q0_host);
/* Same, then copies the (normalised) observation (N, D) row-major into obs_host (may be NULL). */
int mbt_env_reset_host(mbt_env* env, double start_time, const float* q0_host, float* obs_host);

/* ---- step (TE:103-110 and everything it calls: MD:108-131, MD:208-240, ARR:54-56, ARR:110-123,
 *      FILL:28-34, FILL:57-58, TE:198-220, TE:283-289, TE:323-327, MID:60-65, MID:140-143, RW:23-33,
 *      RW:96-109, RW:128-138, TE:112-129) --------------------------------------------------------- */
/* action: (N, A) float32 row-major, A = 2 (limit) or 4 (limit + market).  Outputs: obs (N, D), reward (N),
 * done (scalar; lane-invariant, TE:218-220).  Any output pointer may be NULL.
 * Batches of up to 65536 lanes (the reference's own regime is N ~ 1000) take ONE launch and no interrupt: the step kernel
 * reads the actions from, and mirrors observation rows and rewards into, pinned device-mapped host memory and raises a
 * completion flag there that this call spins on.
 * Opt-in, mbt_config.resident_step or the environment variable MBT_RESIDENT_STEP=1 at creation (batches of up to 4096 lanes, float32 tier, built-in order-book
 * models, production noise): the first such call of an episode starts a RESIDENT kernel of at most four workgroups that stays on the
 * device; this call then only writes the actions and a 64-byte mailbox line (into device memory through the PCIe BAR where the
 * platform allows, host memory otherwise) and spins on the flag - no launch per step: 13.3-13.9 -> 8.7-9.3 us per env.step() at
 * N = 1000.  Results are the one-launch path's to the bit (same kernel code, counters and clock arithmetic).  The kernel leaves at
 * the episode's end, when any other entry point is called on the environment (which waits for it), after 2 ms without a call
 * (MBT_RESIDENT_IDLE_US) and after 30 s in any case; while it is there, kernels on OTHER streams of the device run 20-27 % slower
 * (profiles/r05_resident_step.txt) - which is why it is not the default.  A latency optimisation, not a contract: a step the kernel
 * does not answer within 200 ms (MBT_RESIDENT_ANSWER_MS) is taken by a launch instead - the kernel takes a step whole or not at all -
 * and the environment keeps to one launch per step from there on (one line on stderr says so). */
int mbt_env_step_host(mbt_env* env, const float* action_host, float* obs_host, float* reward_host, int32_t* done);
/* action_device == NULL uses the buffer returned by mbt_env_action_ptr().  Asynchronous. */
int mbt_env_step_device(mbt_env* env, const float* action_device, int32_t* done);
/* k consecutive mbt_env_step_device calls with the same action buffer in ONE host call (k launches; a consumer in an
 * interpreted language pays its call overhead once).  With auto_reset, an episode that ends inside the batch is handled
 * the way SB3's VecEnv contract prescribes (SBE:28-37) without draining the stream: the episode's return sums are
 * reduced on the device (and all-reduced over the communicator of mbt_env_set_communicator, if any) into the episode
 * log, and the lanes are reset with the start time and initial inventories of the last mbt_env_reset*.  Without
 * auto_reset the batch stops at the end of the episode.  steps_done / episodes_ended may be NULL. */
int mbt_env_step_many_device(mbt_env* env, uint32_t k, const float* action_device, int32_t auto_reset, uint32_t* steps_done,
                             uint32_t* episodes_ended);
/* Launch gate for mbt_env_step_many_device: with burst > 0 the launches of a call are enqueued in bursts of `burst` behind a
 * one-thread kernel that waits for a word of host memory, which the library writes once the burst is queued - the kernels
 * of a burst then run back to back however long the HOST needs per launch.  For runs under a tracer (rocprofv3 raises the
 * host's cost per launch to ~11 us, above a 7 us kernel: the queue runs dry and kernels that start on an idle chip take
 * 0.5-1.8 us longer than in the untraced run being profiled); not for production - the device idles while a burst is
 * queued.  The gate kernel gives up by itself after 1 s.  burst <= 4096; 0 = off (default). */
int mbt_env_set_launch_gate(mbt_env* env, uint32_t burst);

/* ---- graph-capturable stepping: the clock on the device (ABI 8) -----------------------------------
 * For consumers whose POLICY lives on the device (torch; the consumer the reference trains through
 * gym/StableBaselinesTradingEnvironment.py:25-37): the loop "policy forward -> action buffer -> mbt_env_step_device" costs
 * 4-5 us of host time per enqueued launch (profiles/r05_timed_region.json), more than a step takes below ~2^19 lanes.  A HIP
 * graph removes that - but mbt_env_step_device hands the step kernel its clock (time, terminal flag, Philox step) as kernel
 * ARGUMENTS computed on the host (TE:216-220), so a captured graph would replay one and the same step.  Between
 * mbt_env_device_clock_begin and mbt_env_device_clock_end the clock lives in device memory instead: the step kernel reads it
 * there and its last workgroup advances it with the host's arithmetic (t += dt in double; done = t >= T - dt/2), the launch
 * arguments no longer depend on the step, and
 *     mbt_env_set_stream(env, s); mbt_env_device_clock_begin(env, MBT_CLOCK_AUTO_RESET);
 *     hipStreamBeginCapture(s) / torch.cuda.graph(g, stream=s):  k x [policy forward into mbt_env_action_ptr(), mbt_env_step_device_captured(env, NULL)]
 *     hipGraphLaunch(...) as often as wanted;  mbt_env_device_clock_end(env)
 * steps the environment k steps per replay - results identical, bit for bit, to the same number of mbt_env_step_device calls
 * (with MBT_CLOCK_AUTO_RESET: to mbt_env_step_many_device(auto_reset = 1), the episode log included).  The state is stepped in
 * place in this mode: mbt_env_obs_ptr() / mbt_env_action_ptr() / mbt_env_reward_ptr() keep ONE address each from begin to end.
 * Every plugin family that steps without the host: built-in models of every tier, speed dynamics, device expressions
 * (mbt_env_create_jit); not injected noise, not host-callback plugins (MBT_ERR_STATE / MBT_ERR_INVALID).
 * While the mode is on, entry points that read or advance the HOST's clock, or change what a launch is handed (reset, step_host,
 * step_device, rollouts, seed, set_step_size, set_state, record_events, track_lane_returns ...), answer MBT_ERR_STATE; buffer
 * pointers, mbt_env_synchronize, the timers, clip_count, return_sums, set_action_host, get_obs_host / get_state_host stay
 * available, and mbt_env_get_clock reads the device's clock (waiting for the stream).  A graph captured in the mode must not be
 * replayed after mbt_env_device_clock_end (the host's clock has taken over again and would know nothing of those steps). */
enum {
  /* SB3's VecEnv contract (SBE:28-37): the launch that ends an episode also logs the episode's return sums (device side, popped by
   * mbt_env_episode_log_pop after mbt_env_device_clock_end: the 16 newest) and resets every lane with the start time and initial
   * inventories of the last mbt_env_reset* - the observation buffer then holds the first observation of the new episode.
   * Without it the clock runs on past the terminal time like the reference's (done stays 1). */
  MBT_CLOCK_AUTO_RESET = 1,
  /* ... and the observation of the episode's LAST step (SBE:32 `terminal_observation`) is kept in the (padded lanes, D) buffer of
   * mbt_env_terminal_obs_ptr() until the next episode ends. */
  MBT_CLOCK_TERMINAL_OBSERVATION = 2
};
typedef struct mbt_device_clock {   /* the first 32 bytes of the clock block, as device code lays them out */
  double time;             /* the clock at the beginning of the next step (TE:216) */
  uint32_t episode_step;   /* steps since the last reset */
  uint32_t philox_step;    /* Philox counter of the next step */
  uint32_t steps;          /* steps taken since mbt_env_device_clock_begin */
  uint32_t episodes;       /* steps among them that ended an episode */
  int32_t done;            /* the last step ended an episode (TE:218-220) */
  uint32_t log_count;      /* episodes logged since begin (MBT_CLOCK_AUTO_RESET) */
} mbt_device_clock;
int mbt_env_device_clock_begin(mbt_env* env, uint32_t flags);
/* Enqueues ONE step on the environment's stream and touches no host state: safe inside a stream capture, valid for every replay.
 * action_device == NULL: the buffer of mbt_env_action_ptr(). */
int mbt_env_step_device_captured(mbt_env* env, const float* action_device);
/* Waits for the environment's stream and reads the clock block. */
int mbt_env_device_clock_read(mbt_env* env, mbt_device_clock* out);
/* The block itself (device memory, laid out as struct mbt_device_clock), for device code that wants `done` or the step count
 * without a host round trip - e.g. a bootstrap mask in a captured training step.  Read-only for callers. */
void* mbt_env_device_clock_ptr(mbt_env* env);
float* mbt_env_terminal_obs_ptr(mbt_env* env);  /* NULL until a mbt_env_device_clock_begin asked for MBT_CLOCK_TERMINAL_OBSERVATION */
/* Waits for the stream, hands the clock back to the host (mbt_env_get_clock, mbt_env_step_device ... continue from where the
 * graph left off) and files the episodes that ended in the mode in the episode log (all-reduced over the communicator of
 * mbt_env_set_communicator, if any).  No-op when the mode is off. */
int mbt_env_device_clock_end(mbt_env* env);

/* ---- fused rollout: many steps in one launch with an on-device closed-form policy ---------------
 * Replaces the caller's per-time-step loop (gym/helpers/generate_trajectory.py:21-34) for policies that are closed
 * forms of the observation.  Bit-identical to the equivalent sequence of mbt_env_step_* calls (same Philox
 * counters).  Runs until the episode ends or max_steps, whichever comes first.
 * Trajectory buffers are optional (NULL = not recorded) and TIME-MAJOR so that device stores coalesce:
 *   obs_traj (steps+1, N, D) with row 0 = the observation before the first step, act_traj (steps, N, A),
 *   rew_traj (steps, N).  generate_trajectory's (N, D, steps+1) layout (GT:11-15) is the transpose (2,0,1)... i.e.
 *   np.transpose(obs_traj, (1, 2, 0)); the Python layer returns that view. */
enum {
  MBT_POLICY_FIXED = 0,               /* params[0..A) = the action every lane takes every step (agents/BaselineAgents.py:25-42) */
  MBT_POLICY_AVELLANEDA_STOIKOV = 1,  /* params[0] = risk aversion gamma (agents/BaselineAgents.py:52-83); needs un-normalised actions */
  MBT_POLICY_TIME_INVENTORY_TABLE = 2, /* (bid, ask) depths looked up by (time step, inventory): any policy that is a function
                                         of (t, q) tabulated by the host, e.g. the Cartea-Jaimungal optimal quotes
                                         (agents/BaselineAgents.py:86-170).  table[(row * table_cols + col) * 2 + side], host
                                         memory, row = round(t / dt), col = clamp(q + table_q_offset, 0, table_cols - 1) */
  MBT_POLICY_TIME_TABLE = 3,          /* open-loop schedule: table[row * A + j], row = round(t / dt), table_cols = A; e.g. the
                                         Cartea-Jaimungal optimal-execution speed (agents/BaselineAgents.py:173-210) */
  MBT_POLICY_ACTION_BUFFER = 4,       /* every lane repeats ITS row of the (N, A) action buffer (mbt_env_action_ptr / set_action_host)
                                         for max_steps steps: "action repeat" - k env.step(action) calls of a consumer that acts
                                         every k-th step, in one launch */
  /* LEARNED policies, evaluated inside the kernel on the observation the environment would hand out (normalised when it
   * normalises, TE:112-118) and clipped to the action space the agent acts in, as Stable-Baselines3 does before env.step -
   * the consumer the reference trains (agents/SbAgent.py, experiments/helpers.py:63-96).  Every dynamics with
   * real-valued actions (limit, limit + market, trading speed).  One fused kernel for the float32 tiers of the built-in
   * order-book models; with the exogenous-depth fill model, precise_state, user-defined plugins (mbt_env_create_jit) or
   * speed dynamics the same call runs the policy as a kernel of its own in front of each step (same recording, same exploration draws: bit-identical to the caller doing so itself).
   * Weights are float32 in host memory in torch.nn.Linear layout (out x in, row-major), concatenated: */
  /* Both: params[1] = 1 clips the action to the action space (SB3 before env.step), 0 passes it on as computed (the
   * reference's PolicyGradientAgent, agents/PolicyGradientAgent.py:34-47); params[2..2+A) = exploration std per action
   * component (0 = deterministic): action = mean + std * eps with eps ~ N(0, 1) from Philox blocks of their own
   * (independent of the environment's noise, reproducible from the seed) - for consumers that COLLECT training data with
   * a stochastic policy; the recorded action trajectory holds the action as applied. */
  MBT_POLICY_LINEAR = 5,              /* mean = W obs + b: table = [W (A x D) | b (A)], table_rows = 0, table_cols = A D + A */
  MBT_POLICY_MLP = 6                  /* two hidden layers of width H <= 64 (SB3's MlpPolicy actor is [64, 64] tanh):
                                         table = [W1 (H x D) | b1 (H) | W2 (H x H) | b2 (H) | W3 (A x H) | b3 (A)], table_rows = H,
                                         table_cols = the number of floats; params[0] = activation (0 tanh, 1 relu).  Runs on the
                                         matrix cores (v_mfma_f32_16x16x16_f16: operands rounded to fp16, fp32 accumulation). */
};
typedef struct mbt_policy {
  int32_t kind;
  int32_t reserved;
  double params[8];
  const float* table;      /* MBT_POLICY_TIME_INVENTORY_TABLE only */
  uint32_t table_rows, table_cols;
  int32_t table_q_offset;
  int32_t reserved1;
} mbt_policy;
/* mbt_env_rollout_host keeps its HBM staging (up to 8 GiB per recorded array: a 2^20-lane, 200-step recording is 5.9 GB, and
 * allocating and freeing it every episode cost as much as copying it out) and mbt_env_step_host its pinned bounce buffer until
 * mbt_env_destroy.  A long-lived environment that recorded once and will not again gives them back with this call (they are
 * re-created on demand). */
int mbt_env_release_staging(mbt_env* env);
/* Device variant: trajectory pointers are device memory sized for the PADDED lane count mbt_env_padded_lanes(). */
int mbt_env_rollout_device(mbt_env* env, const mbt_policy* policy, uint32_t max_steps, float* obs_traj, float* act_traj,
                           float* rew_traj, uint32_t* steps_done, int32_t* done);
/* Host variant: trajectory pointers are host memory with exactly N lanes per time slice; synchronous. */
int mbt_env_rollout_host(mbt_env* env, const mbt_policy* policy, uint32_t max_steps, float* obs_traj, float* act_traj,
                         float* rew_traj, uint32_t* steps_done, int32_t* done);
/* Evaluates a LEARNED policy (MBT_POLICY_LINEAR / MBT_POLICY_MLP) on the current observation into the buffer of
 * mbt_env_action_ptr(): the step-loop form of what the fused rollout does in-kernel - "mbt_env_policy_device, then
 * mbt_env_step_device(NULL)" repeated is bit-identical to one mbt_env_rollout_device with the same policy.  Asynchronous. */
int mbt_env_policy_device(mbt_env* env, const mbt_policy* policy);
uint64_t mbt_env_padded_lanes(mbt_env* env); /* N rounded up to whole 512-lane tiles (quads for speed dynamics):
                                                lanes per time slice of device trajectories */

/* ---- injected noise (parity mode; replaces the three numpy Generators of SP:27) ---------------- */
/* u_arr, u_fill: (N, 2) float32 in [0, 1) (ignored, may be NULL, for speed dynamics); z: (N) float32.
 * Consumed by the next step. */
int mbt_env_set_noise_host(mbt_env* env, const float* u_arr, const float* u_fill, const float* z);
/* z_user: (N, 2) float32, the two extra normals of user processes (mbt_user_code.extra_normals); consumed by the next step,
 * together with the arrays of mbt_env_set_noise_host. */
int mbt_env_set_user_noise_host(mbt_env* env, const float* z_user);

/* ---- device buffers (zero-copy consumers) ------------------------------------------------------ */
/* (N, A) staging buffer a device policy may write.  After a small-batch mbt_env_step_host the newest actions sit in the
 * library's host stage; THIS call files them into the buffer first, so a writer must (re-)fetch the pointer after a host step
 * rather than write through one it fetched before (the library cannot see such a write and would file the stage over it at
 * the next mbt_env_step_device(NULL)).  Passing an explicit action pointer to step_device / step_many_device is always safe. */
float* mbt_env_action_ptr(mbt_env* env);
/* (N, D) observation of the last reset/step; without normalisation this IS the state, which the next step updates IN PLACE: the
 * rows are valid until the next step / rollout / reset is enqueued on the environment's stream (copy them there to keep them). */
float* mbt_env_obs_ptr(mbt_env* env);
/* Which of the two regimes this environment steps in (mbt_env.hip: mbt_env::state): 1 = the state is updated IN PLACE - launches that
 * move 80 MB or more, every environment that normalises observations (one observation buffer), and every environment while its clock
 * is on the device - so the pointer above is the same after every step and the rows of step k are overwritten by step k + 1 (a
 * consumer that keeps `obs` next to `next_obs`, e.g. a replay buffer, copies the former on the environment's stream first); 0 = two
 * buffers alternate and the rows of step k stay valid until step k + 2 is enqueued.  MBT_PING_PONG_STATE = 0 / 1 forces a regime. */
int mbt_env_state_in_place(mbt_env* env);
/* (N) rewards of the last step.  After a HOST step (mbt_env_step_host, mbt_env_set_host_rewards) the buffer is complete when this
 * returns; after a device step it is complete once the environment's stream has reached that step (mbt_env_synchronize). */
float* mbt_env_reward_ptr(mbt_env* env);
int mbt_env_obs_dim(mbt_env* env);
int mbt_env_action_dim(mbt_env* env);

/* ---- state access / checkpoint (TE:142-144 `state`) -------------------------------------------- */
/* Un-normalised state (N, D) row-major float32. */
int mbt_env_get_state_host(mbt_env* env, float* state_host);
/* The same as float64, the dtype of the reference's `state` (TE:142-144).  With precise_state these ARE the reference's
 * float64 values (float32 rounding + int32 remainder, joined on the host; the TIME column is the host's float64 clock);
 * without it, the float32 state widened. */
int mbt_env_get_state_f64_host(mbt_env* env, double* state_host);
/* The representation precise_state keeps a float64 value x in (8 bytes, the same as a float32 pair, but EXACT):
 * hi = float32(x), rounded to nearest - what the observation shows - and lo = (x - hi) * 2^(53 - e) as an int32, e the
 * exponent of hi.  x - hi is at most half a float32 ulp and a multiple of 2^(e-53), so lo is an integer of magnitude
 * <= 2^29 and join(split(x)) == x for every double whose float32 rounding is a normal number (zero, denormal and
 * non-finite hi carry lo = 0).  Host-side restatements of the device functions, for bindings and tests. */
void mbt_exact_split(double x, float* hi, int32_t* lo);
double mbt_exact_join(float hi, int32_t lo);
/* x ** p as the float32 tier's kernels evaluate it (IMP:55-56 `action ** exponent`, RW:59-68 `inventory ** exponent` for exponents other than
 * 1 and 2; step_kernel.hpp: power_f32): NumPy's float64 power rounded to float32 once, to within 0.5001 ulp - n values on `device`, host arrays
 * in and out.  For tests and for checking the tier's arithmetic; no environment needed. */
int mbt_power_f32_device(int device, const float* x_host, double p, float* out_host, uint32_t n);
/* Observation of the last reset/step as the API returns it ((N, D), normalised when configured, TE:112-118). */
int mbt_env_get_obs_host(mbt_env* env, float* obs_host);
/* Upload (N, A) actions into the buffer of mbt_env_action_ptr() (e.g. a fixed quote for step_device loops). */
int mbt_env_set_action_host(mbt_env* env, const float* action_host);
int mbt_env_set_state_host(mbt_env* env, const float* state_host, double time, uint32_t philox_step);
int mbt_env_get_clock(mbt_env* env, double* time, uint32_t* episode_step, uint32_t* philox_step);

/* ---- diagnostics ------------------------------------------------------------------------------- */
/* When enabled the step kernel also writes one byte per lane: bit0/1 arrival bid/ask (ARR:56), bit2/3 fill
 * bid/ask after the max-inventory mask (TE:323-327), bit4/5 market buy/sell (MD:209-210), bit6 inventory
 * clipped, bit7 cash clipped (TE:283-289). */
int mbt_env_record_events(mbt_env* env, int enabled);
int mbt_env_get_events_host(mbt_env* env, uint8_t* events_host);
/* Number of lane-steps on which the clip of TE:283-289 changed a value since create (the reference prints). */
int mbt_env_clip_count(mbt_env* env, uint64_t* count);

/* ---- episode-return statistics (the quantity SB3's VecMonitor / plotting.py:96-108 report) ------ */
/* sums[0] = sum over lanes of the rewards since the last reset, sums[1] = sum of squared per-lane returns
 * (NaN unless per-lane tracking is on), sums[2] = number of lanes.  These three doubles are what the
 * multi-GPU all-reduce carries. */
int mbt_env_track_lane_returns(mbt_env* env, int enabled);
int mbt_env_return_sums(mbt_env* env, double sums[3]);
/* The same in two halves, so that an episode boundary does not drain the stream: _begin enqueues the reduction of the
 * episode that just ended (then reset and keep stepping), _end waits for that reduction alone and returns the sums.
 * One request in flight at a time. */
int mbt_env_return_sums_begin(mbt_env* env);
int mbt_env_return_sums_end(mbt_env* env, double sums[3]);
/* Episode log of mbt_env_step_many_device(auto_reset): pops the OLDEST finished episode's [sum R, sum R^2, lanes] (global
 * over all ranks when a communicator is set).  Returns 1 and fills sums if one was popped, 0 if the log is empty - or,
 * with wait == 0, if the oldest entry's reduction has not completed yet.  The log holds the 16 most recent episodes: when
 * a 17th ends before the oldest has been popped, the library waits for the oldest entry's reduction and DROPS it. */
int mbt_env_episode_log_pop(mbt_env* env, double sums[3], int32_t wait);

/* ---- multi-GPU: the trajectory axis is sharded, one handle per GPU; this is the ONLY collective on the path -------
 * (replaces the concatenation of MultiprocessTradingEnv workers, gym/MultiprocessTradingEnv.py:74-80,112-116, for the one
 * statistic a run reports).  RCCL is bound at run time (dlopen of librccl.so.1 - the copy already loaded in the process,
 * e.g. PyTorch's, if there is one), so the library has no link-time dependency on it. */
/* In place, blocking: sums[3] (host; this rank's [sum R, sum R^2, count]) -> the sums over all ranks of `nccl_comm`
 * (an ncclComm_t created on this environment's device).  One 24-byte ncclAllReduce on the environment's stream. */
int mbt_env_allreduce_returns(mbt_env* env, void* nccl_comm, double sums[3]);
/* Communicator used by the episode log of mbt_env_step_many_device (NULL = none: local sums). */
int mbt_env_set_communicator(mbt_env* env, void* nccl_comm);
/* Thin wrappers so that a binding needs no second FFI: ncclGetUniqueId (id_out: 128 bytes), ncclCommInitRank on
 * `device`, ncclCommDestroy.  The id travels from rank 0 to the others by whatever the launcher offers. */
#define MBT_COMM_ID_BYTES 128
int mbt_comm_unique_id(void* id_out);
int mbt_comm_init_rank(int device, int n_ranks, const void* id, int rank, void** comm_out);
/* ncclCommCount: how many ranks the communicator spans, as RCCL itself reports it (evidence for a multi-GPU run's log). */
int mbt_comm_count(void* comm, int* n_ranks);
int mbt_comm_destroy(void* comm);

/* ---- RewardFunction.calculate on caller-supplied matrices (RW:23-33, RW:96-109, RW:128-138) ---------------------
 * cur, nxt: (n, dim) row-major float64 state matrices; q_init, episode_length: (n) float64, only for MBT_REW_CJ_MM /
 * MBT_REW_CJ_OE (what their reset() captured, RW:72-74, RW:111-113); out: (n) float64.  Evaluated on the device in double, in the
 * reference's order of operations. */
int mbt_reward_calculate_host(int device, int reward_kind, double phi, double alpha, double inventory_exponent,
                              const double* cur, const double* nxt, int dim, uint64_t n, int is_terminal,
                              const double* q_init, const double* episode_length, const double* action /* (n), CJ_OE */,
                              double risk_aversion /* EXP_UTILITY */, double* out);

/* ---- StochasticProcessModel.update / get_arrivals / get_fills for host callers (SP:33-35, ARR:27-29, FILL:28-34) ----
 * The plugin objects of the reference can be driven on their own, outside an environment (a midprice path, an arrival
 * stream).  There is no CPU implementation of their arithmetic here either: one call evaluates one such method on
 * caller-supplied float64 arrays on the device, in double and in the reference's order of operations, with the draws the
 * CALLER supplies (the Python descriptors draw them from the same numpy Generator the reference's classes own, so a seeded
 * process object walks the reference's path).  cfg: only the fields of the process in question are read (kinds, parameters,
 * *_step_size).  Arrays are row-major; booleans come back as 0.0 / 1.0.
 *   MBT_PROCESS_MIDPRICE_UPDATE  a = S (n), b = z (n), c / d = the agent's bid / ask fills (n) or NULL   -> out = S' (n)
 *   MBT_PROCESS_HAWKES_UPDATE    a = intensities (n, 2), b = arrivals (n, 2)                              -> out (n, 2)
 *   MBT_PROCESS_ARRIVALS         a = uniforms (n, 2), b = intensities (n, 2) for Hawkes, else NULL       -> out (n, 2)
 *   MBT_PROCESS_FILLS            a = uniforms (n, 2), b = depths (n, 2)                                   -> out (n, 2) */
enum { MBT_PROCESS_MIDPRICE_UPDATE = 0, MBT_PROCESS_HAWKES_UPDATE = 1, MBT_PROCESS_ARRIVALS = 2, MBT_PROCESS_FILLS = 3 };
int mbt_process_evaluate_host(int device, int op, const mbt_config* cfg, uint64_t n, const double* a, const double* b, const double* c,
                              const double* d, double* out);

/* ---- the generator itself (so tests can pin it) ------------------------------------------------ */
/* Writes the noise lane ids [trajectory_offset, trajectory_offset + n) would draw at philox step `step`
 * under `seed` into host arrays (any may be NULL): u_arr (n,2), u_fill (n,2), z (n). */
int mbt_rng_fill_host(int device, uint64_t seed, uint64_t trajectory_offset, uint32_t step, uint64_t n,
                      float* u_arr, float* u_fill, float* z);
/* The extra normals of user processes (mbt_user_code.extra_normals): z_user (n, 2); offset a multiple of 512. */
int mbt_rng_fill_user_host(int device, uint64_t seed, uint64_t trajectory_offset, uint32_t step, uint64_t n, float* z_user);
/* Same for the speed-dynamics stream (one normal per lane, drawn per quad of lanes 256 apart in a 1024-lane tile): z (n); offset multiple of 1024. */
int mbt_rng_fill_quad_host(int device, uint64_t seed, uint64_t trajectory_offset, uint32_t step, uint64_t n, float* z);
/* Raw Philox4x32-10 block function on the device: out[4] = philox(ctr[4], key[2]) (known-answer tests). */
int mbt_philox4x32_10_host(int device, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* MEASUREMENT: the write-only floor of a recording - one launch that writes `steps` time slices into device trajectory buffers laid out as
 * for mbt_env_rollout_device (any may be NULL), with the rollout kernel's thread mapping and store instructions and no other work; enqueued
 * on the environment's stream, so mbt_env_timer_* bracket it like a rollout.  For bench.py: floor and rollout timed in one process against
 * the same buffers (16-byte observation rows only).  Touches nothing of the environment. */
int mbt_env_record_floor_device(mbt_env* env, uint32_t steps, float* obs_traj, float* act_traj, float* rew_traj);

/* ---- timing on the environment's stream (HIP events) ------------------------------------------- */
int mbt_env_timer_begin(mbt_env* env);
int mbt_env_timer_end(mbt_env* env, float* elapsed_ms); /* = stop + elapsed: synchronises */
/* The same in two halves, for a caller whose own clock brackets the work: `stop` only records the closing event (no
 * wait: nothing but the launches sits between the caller's two synchronisation points), `elapsed` waits for it and
 * reads the device time between the two events - to be called once the caller's clock has stopped. */
int mbt_env_timer_stop(mbt_env* env);
int mbt_env_timer_elapsed(mbt_env* env, float* elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* MBT_ENV_H */
