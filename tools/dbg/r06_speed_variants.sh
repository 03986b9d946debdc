#!/bin/bash
# speed-dynamics precise_state kernels after the out-of-line pow()/exp() and the body refactor of round 6: lanes per group (MBT_SPEED_PRECISE_GROUPS)
set -u
OUT=gpurun_out/r06d; mkdir -p "$OUT"
CASES=("speed temp+perm impact, CjOe 2^20, precise_state" "speed temp+perm impact, CjOe 2^20 (D=5" "speed power impact ^1.5")
for round in 1 2; do
  for v in default g1 g4; do
    for c in "${CASES[@]}"; do
      if [ "$v" = default ]; then
        r=$(MBT_BENCH_STEPS=1500 MBT_BENCH_ONLY="$c" python tests/perf/bench_configs.py 2>/dev/null | grep -o '"us_per_step": [0-9.]*')
      else
        r=$(MBT_LIBRARY_VARIANT=$v MBT_EXTRA_HIPCC_FLAGS="-DMBT_SPEED_PRECISE_GROUPS=${v#g}" MBT_BENCH_STEPS=1500 MBT_BENCH_ONLY="$c" python tests/perf/bench_configs.py 2>/dev/null | grep -o '"us_per_step": [0-9.]*')
      fi
      echo "$v | $c | $r" | tee -a "$OUT/speed_variants.txt"
    done
  done
done
