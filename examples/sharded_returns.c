/* Multi-GPU from plain C: the trajectory axis sharded over ranks (one process per GPU), no data-path collective, and the
 * ONE exchange of the path - [sum R, sum R^2, lanes] of every finished episode - all-reduced over RCCL through the C ABI
 * (what replaces the concatenation of MultiprocessTradingEnv workers, gym/MultiprocessTradingEnv.py:74-80,112-116).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/sharded_returns.c -Lmbt_gym_amd -lmbtenv -lm -Wl,-rpath,'$ORIGIN/../mbt_gym_amd' -o examples/sharded_returns
 *   examples/sharded_returns                      one rank (a world of 1: the same code path, RCCL included)
 *   examples/sharded_returns R W /tmp/id.bin      rank R of W, GPU R; rank 0 writes the 128-byte RCCL id to the file, the others
 *                                                 wait for it (any side channel of the launcher would do)
 * Every rank prints the GLOBAL mean episode return; it does not depend on W (Philox is keyed on global lane ids). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "mbt_env.h"

#define CHECK(call)                                                    \
  do {                                                                 \
    int rc_ = (call);                                                  \
    if (rc_ < 0) {                                                     \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mbt_last_error()); \
      return 1;                                                        \
    }                                                                  \
  } while (0)

int main(int argc, char** argv) {
  enum { TOTAL_LANES = 1 << 16, STEPS = 100, EPISODES = 3 };
  const int rank = argc > 2 ? atoi(argv[1]) : 0, world = argc > 2 ? atoi(argv[2]) : 1;
  const char* id_path = argc > 3 ? argv[3] : NULL;
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && id_path == NULL) || TOTAL_LANES % (1024 * world) != 0) {
    fprintf(stderr, "usage: sharded_returns [rank world id-file]   (world must divide %d lanes into multiples of 1024)\n", TOTAL_LANES);
    return 2;
  }
  const int device = mbt_device_count() > rank ? rank : 0;

  /* the RCCL communicator: id from rank 0, by file */
  unsigned char id[MBT_COMM_ID_BYTES];
  if (rank == 0) {
    CHECK(mbt_comm_unique_id(id));
    if (id_path != NULL) {
      char tmp[1024];
      snprintf(tmp, sizeof tmp, "%s.tmp", id_path);
      FILE* f = fopen(tmp, "wb");
      if (f == NULL || fwrite(id, 1, sizeof id, f) != sizeof id) return 3;
      fclose(f);
      if (rename(tmp, id_path) != 0) return 3; /* atomic: readers never see a partial id */
    }
  } else {
    FILE* f = NULL;
    for (int tries = 0; tries < 600 && (f = fopen(id_path, "rb")) == NULL; ++tries) usleep(100000);
    if (f == NULL || fread(id, 1, sizeof id, f) != sizeof id) return 3;
    fclose(f);
  }
  void* comm = NULL;
  CHECK(mbt_comm_init_rank(device, world, id, rank, &comm));

  /* this rank's shard of an Avellaneda-Stoikov market (BASELINE.json configs[1] at a small size) */
  mbt_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = MBT_ABI_VERSION;
  cfg.device = device;
  cfg.num_trajectories = TOTAL_LANES / world;
  cfg.trajectory_offset = (uint64_t)rank * cfg.num_trajectories; /* global id of this shard's lane 0 */
  cfg.n_steps = STEPS;
  cfg.terminal_time = 0.1; /* dt = 1e-3, as in BASELINE.json configs[1] */
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
  cfg.max_inventory = STEPS;
  cfg.max_cash = STEPS * 108.0;
  cfg.reward_scale = 1.0;
  cfg.seed = 50;
  mbt_env* env = NULL;
  CHECK(mbt_env_create(&cfg, &env));
  CHECK(mbt_env_set_communicator(env, comm));
  CHECK(mbt_env_track_lane_returns(env, 1));

  float* quote = malloc(sizeof(float) * 2 * cfg.num_trajectories);
  if (quote == NULL) return 1;
  for (uint64_t i = 0; i < 2 * cfg.num_trajectories; ++i) quote[i] = 0.7f;
  CHECK(mbt_env_set_action_host(env, quote));
  CHECK(mbt_env_reset(env, 0.0, NULL));

  /* EPISODES episodes in ONE call: launches, per-episode reductions + all-reduces and resets are all enqueued */
  uint32_t steps = 0, episodes = 0;
  CHECK(mbt_env_step_many_device(env, EPISODES * STEPS, NULL, 1, &steps, &episodes));
  for (uint32_t k = 0; k < episodes; ++k) {
    double sums[3];
    if (mbt_env_episode_log_pop(env, sums, 1) != 1) return 4;
    printf("rank %d/%d episode %u: %.0f lanes in total, mean return %.6f, second moment %.4f\n", rank, world, k, sums[2], sums[0] / sums[2], sums[1] / sums[2]);
    if (sums[2] != TOTAL_LANES) return 5;
  }
  /* the blocking form, on sums the caller holds (here: a partial episode) */
  CHECK(mbt_env_step_many_device(env, 10, NULL, 1, NULL, NULL));
  double partial[3];
  CHECK(mbt_env_return_sums(env, partial));
  CHECK(mbt_env_allreduce_returns(env, comm, partial));
  printf("rank %d/%d after 10 more steps: %.0f lanes, mean return so far %.6f\n", rank, world, partial[2], partial[0] / partial[2]);

  free(quote);
  mbt_env_destroy(env);
  CHECK(mbt_comm_destroy(comm));
  return steps == EPISODES * STEPS && episodes == EPISODES ? 0 : 6;
}
