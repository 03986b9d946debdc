#!/bin/bash
set -u
OUT=gpurun_out/r05m; mkdir -p "$OUT"; export TMPDIR=/tmp
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/soak.txt"; }
MBT_HAWKES_SOAK=1 timeout 900 python -m pytest tests/test_gpu_round5.py -q -k "soak" 2>&1 | tail -3 | tee -a "$OUT/soak.txt"; stamp "Hawkes soak: 2^17 lanes x 800 steps = 1.05e8 lane-steps of BASELINE configs[3], default tier vs the float64 oracle"
MBT_FUZZ_SCALE=8 MBT_FUZZ_SEED=500000 timeout 1500 python -m pytest tests/test_gpu_random_configs.py -q -x -n 4 2>&1 | tail -4 | tee -a "$OUT/soak.txt"; stamp "fuzz soak: MBT_FUZZ_SCALE=8 MBT_FUZZ_SEED=500000 (1200 default-tier + 240 float32-intensity + 720 precise + 480 speed x2 + 480 rollout configurations)"
for v in 0 24576 32768; do echo "== MBT_STEP_DYNAMIC_LDS=$v" | tee -a "$OUT/precise_cap.txt"; MBT_STEP_DYNAMIC_LDS=$v MBT_BENCH_STEPS=1500 MBT_BENCH_ONLY="precise_state" python tests/perf/bench_configs.py 2>/dev/null | grep "us_per_step" | tee -a "$OUT/precise_cap.txt"; done
stamp "precise cap"
