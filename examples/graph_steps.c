/* The graph-capturable step from plain C: capture k launches of mbt_env_step_device_captured in a HIP graph ONCE, replay it, and arrive where
 * the ordinary loop (mbt_env_step_many_device) arrives - state, clock and episode log, to the bit.
 *
 *   gcc -std=gnu99 -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/graph_steps.c -Lmbt_gym_amd -lmbtenv -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,'$ORIGIN/../mbt_gym_amd' -Wl,-rpath,/opt/rocm/lib -o examples/graph_steps
 *
 * Why a consumer wants this: a policy that lives on the device (here: none - a fixed quote, so that the example needs no second library) pays the host
 * 4-5 us per enqueued launch; a graph pays them once.  mbt_env_step_device hands the step kernel its clock as kernel ARGUMENTS computed on the host, so
 * a captured graph of it would replay one and the same step; between mbt_env_device_clock_begin and _end the clock lives on the device, the launch
 * arguments are the same for every step, and the launch that ends an episode resets the lanes and logs the return sums itself (include/mbt_env.h,
 * "graph-capturable stepping"; the reference's counterpart: the auto-reset of gym/StableBaselinesTradingEnvironment.py:28-37). */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mbt_env.h"

#define CHECK(call)                                                    \
  do {                                                                 \
    int rc_ = (call);                                                  \
    if (rc_ < 0) {                                                     \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mbt_last_error()); \
      return 1;                                                        \
    }                                                                  \
  } while (0)
#define HIP(call)                                                              \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));             \
      return 1;                                                                \
    }                                                                          \
  } while (0)

enum { N = 4096, STEPS = 200, GRAPH_STEPS = 7, REPLAYS = 60 }; /* 420 steps: two episode ends, neither at a graph boundary */

static int make(mbt_env** env, float* action) {
  mbt_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = MBT_ABI_VERSION;
  cfg.num_trajectories = N;
  cfg.n_steps = STEPS;
  cfg.terminal_time = 1.0;
  cfg.midprice_kind = MBT_MID_BROWNIAN;
  cfg.volatility = 2.0;
  cfg.initial_price = 100.0;
  cfg.arrival_kind = MBT_ARR_POISSON;
  cfg.intensity[0] = cfg.intensity[1] = 140.0;
  cfg.fill_kind = MBT_FILL_EXPONENTIAL;
  cfg.fill_exponent = 1.5;
  cfg.dynamics_kind = MBT_DYN_LIMIT;
  cfg.reward_kind = MBT_REW_PNL;
  cfg.impact_kind = MBT_IMPACT_NONE;
  cfg.inventory_exponent = 2.0;
  cfg.max_inventory = 200.0;
  cfg.max_cash = STEPS * 108.0;
  cfg.reward_scale = 1.0;
  cfg.seed = 50;
  CHECK(mbt_env_create(&cfg, env));
  CHECK(mbt_env_reset(*env, 0.0, NULL));
  CHECK(mbt_env_set_action_host(*env, action));
  return 0;
}

int main(void) {
  float* action = malloc(sizeof(float) * N * 2);
  float* state_a = malloc(sizeof(float) * N * 4);
  float* state_b = malloc(sizeof(float) * N * 4);
  if (!action || !state_a || !state_b) return 1;
  for (int i = 0; i < 2 * N; ++i) action[i] = 0.7f;
  const unsigned total = GRAPH_STEPS * REPLAYS;

  /* the ordinary loop: k launches in one call, the host's clock, reset + episode log enqueued by the library */
  mbt_env* loop = NULL;
  if (make(&loop, action)) return 1;
  uint32_t steps_done = 0, episodes = 0;
  CHECK(mbt_env_step_many_device(loop, total, NULL, 1, &steps_done, &episodes));
  CHECK(mbt_env_get_state_host(loop, state_a));

  /* the same steps from a graph */
  mbt_env* env = NULL;
  if (make(&env, action)) return 1;
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));
  CHECK(mbt_env_set_stream(env, stream));
  CHECK(mbt_env_device_clock_begin(env, MBT_CLOCK_AUTO_RESET));
  hipGraph_t graph;
  hipGraphExec_t executable;
  HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
  for (int k = 0; k < GRAPH_STEPS; ++k) CHECK(mbt_env_step_device_captured(env, NULL)); /* launches only: recorded, not run */
  HIP(hipStreamEndCapture(stream, &graph));
  HIP(hipGraphInstantiate(&executable, graph, NULL, NULL, 0));
  for (int r = 0; r < REPLAYS; ++r) HIP(hipGraphLaunch(executable, stream));
  mbt_device_clock now;
  CHECK(mbt_env_device_clock_read(env, &now)); /* waits for the stream */
  CHECK(mbt_env_device_clock_end(env));        /* the host's clock takes over; the episodes that ended are in the log */
  CHECK(mbt_env_get_state_host(env, state_b));

  double time_a, time_b;
  uint32_t episode_step_a, episode_step_b, philox_a, philox_b;
  CHECK(mbt_env_get_clock(loop, &time_a, &episode_step_a, &philox_a));
  CHECK(mbt_env_get_clock(env, &time_b, &episode_step_b, &philox_b));
  int same = memcmp(state_a, state_b, sizeof(float) * N * 4) == 0 && time_a == time_b && episode_step_a == episode_step_b && philox_a == philox_b &&
             now.steps == total && now.episodes == episodes;
  double log_a[3], log_b[3];
  unsigned logged = 0;
  while (mbt_env_episode_log_pop(loop, log_a, 1) == 1) {
    if (mbt_env_episode_log_pop(env, log_b, 1) != 1 || memcmp(log_a, log_b, sizeof log_a) != 0) same = 0;
    printf("episode %u: mean return %.6f over %.0f lanes (loop) | %.6f (graph)\n", logged, log_a[0] / log_a[2], log_a[2], log_b[0] / log_b[2]);
    ++logged;
  }
  printf("loop  : %u steps, %u episodes ended, clock t = %.4f (episode step %u, Philox step %u)\n", steps_done, episodes, time_a, episode_step_a, philox_a);
  printf("graph : %u steps = %d replays of a %d-step graph, %u episodes ended, clock t = %.4f (episode step %u, Philox step %u)\n", now.steps, REPLAYS, GRAPH_STEPS,
         now.episodes, time_b, episode_step_b, philox_b);
  printf("state, clock and episode log %s\n", same && logged == episodes ? "identical" : "DIFFER");

  HIP(hipGraphExecDestroy(executable));
  HIP(hipGraphDestroy(graph));
  mbt_env_destroy(env);
  mbt_env_destroy(loop);
  HIP(hipStreamDestroy(stream));
  free(action);
  free(state_a);
  free(state_b);
  return same && logged == episodes ? 0 : 2;
}
