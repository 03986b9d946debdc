/* The C ABI without Python: one Avellaneda-Stoikov episode (BASELINE.json configs[0]: N = 1000 trajectories, 200 steps)
 * stepped from plain C with a host-side policy, then the same episode as ONE fused rollout launch.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/as_episode.c -Lmbt_gym_amd -lmbtenv -lm -Wl,-rpath,'$ORIGIN/../mbt_gym_amd' -o examples/as_episode
 *
 * What a C (or cgo / JNI / FFI) caller of the reference's hot path needs: mbt_config, create, reset_host, step_host,
 * return_sums, destroy - every function returns 0 or a negative mbt_status and mbt_last_error() explains. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mbt_env.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    int rc_ = (call);                                                      \
    if (rc_ < 0) {                                                         \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mbt_last_error());     \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main(void) {
  enum { N = 1000, STEPS = 200 };
  const double gamma = 0.1, sigma = 2.0, kappa = 1.5, T = 1.0;

  mbt_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = MBT_ABI_VERSION;
  cfg.num_trajectories = N;
  cfg.n_steps = STEPS;
  cfg.terminal_time = T;
  cfg.midprice_kind = MBT_MID_BROWNIAN;      /* BrownianMotionMidpriceModel(volatility=2, initial_price=100) */
  cfg.volatility = sigma;
  cfg.initial_price = 100.0;
  cfg.arrival_kind = MBT_ARR_POISSON;        /* PoissonArrivalModel(intensity=[140, 140]) */
  cfg.intensity[0] = cfg.intensity[1] = 140.0;
  cfg.fill_kind = MBT_FILL_EXPONENTIAL;      /* ExponentialFillFunction(fill_exponent=1.5) */
  cfg.fill_exponent = kappa;
  cfg.dynamics_kind = MBT_DYN_LIMIT;         /* LimitOrderModelDynamics */
  cfg.reward_kind = MBT_REW_PNL;
  cfg.impact_kind = MBT_IMPACT_NONE;
  cfg.inventory_exponent = 2.0;
  cfg.max_inventory = 200.0;
  cfg.max_cash = STEPS * 108.0;              /* n_steps * max_stock_price (TE:229-230) */
  cfg.reward_scale = 1.0;
  cfg.seed = 50;

  mbt_env* env = NULL;
  CHECK(mbt_env_create(&cfg, &env));
  /* host buffers of the step: pinned memory from the library, which its DMA copies then read and write directly (any host
     pointer works - pageable ones go through a bounce buffer) */
  float* obs = mbt_host_alloc(sizeof(float) * N * 4);
  float* act = mbt_host_alloc(sizeof(float) * N * 2);
  float* rew = mbt_host_alloc(sizeof(float) * N);
  double* total = calloc(N, sizeof(double));
  if (!obs || !act || !rew || !total) return 1;

  /* the reference's loop (generate_trajectory.py:21-34): agent.get_action(obs); env.step(action) */
  CHECK(mbt_env_reset_host(env, 0.0, NULL, obs));
  int32_t done = 0;
  int steps = 0;
  while (!done) {
    for (int i = 0; i < N; ++i) { /* AvellanedaStoikovAgent.get_action (agents/BaselineAgents.py:70-83) */
      const double q = obs[4 * i + 1], tau = T - obs[4 * i + 2];
      const double shift = q * gamma * sigma * sigma * tau;
      const double spread = gamma * sigma * sigma * tau + 2.0 / gamma * log(1.0 + gamma / kappa);
      act[2 * i] = (float)(shift + spread / 2);
      act[2 * i + 1] = (float)(-shift + spread / 2);
    }
    CHECK(mbt_env_step_host(env, act, obs, rew, &done));
    for (int i = 0; i < N; ++i) total[i] += rew[i];
    ++steps;
  }
  double mean = 0.0, sums[3];
  for (int i = 0; i < N; ++i) mean += total[i] / N;
  CHECK(mbt_env_return_sums(env, sums));
  printf("step loop : %d steps, mean episode return %.4f (device reduction %.4f)\n", steps, mean, sums[0] / sums[2]);
  if (steps != STEPS || fabs(mean - sums[0] / sums[2]) > 1e-3) return 2;

  /* the same episode in one launch: reseeding restarts the Philox stream, so the draws are identical */
  mbt_policy policy;
  memset(&policy, 0, sizeof policy);
  policy.kind = MBT_POLICY_AVELLANEDA_STOIKOV;
  policy.params[0] = gamma;
  uint32_t steps_done = 0;
  CHECK(mbt_env_seed(env, cfg.seed));
  CHECK(mbt_env_reset(env, 0.0, NULL));
  CHECK(mbt_env_rollout_host(env, &policy, STEPS, NULL, NULL, NULL, &steps_done, &done));
  CHECK(mbt_env_return_sums(env, sums));
  printf("rollout   : %u steps, mean episode return %.4f\n", steps_done, sums[0] / sums[2]);
  if (steps_done != STEPS || !done || fabs(mean - sums[0] / sums[2]) > 0.02) return 3; /* the device policy rounds its quotes in float32: a fill may flip */

  mbt_host_free(obs); mbt_host_free(act); mbt_host_free(rew); free(total);
  mbt_env_destroy(env);
  return 0;
}
