#!/bin/bash
# speed-dynamics precise_state kernel: lanes per group x min waves per SIMD (MBT_SPEED_PRECISE_GROUPS / _WAVES), each a library of its own
set -u
OUT=gpurun_out/r05l; mkdir -p "$OUT"
for round in 1 2; do
for v in default g2_w1 g2_w5 g4_w6 g1_w1; do
  cp ab/variants/lib_$v.so mbt_gym_amd/libmbtenv.so
  echo "== $v" | tee -a "$OUT/speed_variants.txt"
  MBT_BENCH_STEPS=1500 MBT_BENCH_ONLY="speed temp+perm impact, CjOe 2^20" python tests/perf/bench_configs.py 2>/dev/null | grep "us_per_step" | tee -a "$OUT/speed_variants.txt"
done
done
cp ab/variants/lib_default.so mbt_gym_amd/libmbtenv.so
