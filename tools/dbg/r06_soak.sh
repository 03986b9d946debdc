#!/bin/bash
# soak of the round-6 library: random configurations against the oracle (every tier and family), the Hawkes default tier, and the graph-capturable step
set -u
OUT=gpurun_out/r06_soak; mkdir -p "$OUT"; t0=$(date +%s)
MBT_FUZZ_SCALE=${SCALE:-100} MBT_FUZZ_SEED=${SEED:-6000000} timeout 3000 python -m pytest tests/test_gpu_random_configs.py -q -n 12 -p no:cacheprovider 2>&1 | tail -6 | tee "$OUT/soak.txt"
echo "[$(( $(date +%s) - t0 )) s] MBT_FUZZ_SCALE=${SCALE:-100} MBT_FUZZ_SEED=${SEED:-6000000} (final round-6 library): 450 x scale random configurations against the oracle" | tee -a "$OUT/soak.txt"
MBT_HAWKES_SOAK=2 timeout 900 python -m pytest tests/test_gpu_round5.py -q -k soak -p no:cacheprovider 2>&1 | tail -3 | tee -a "$OUT/soak.txt"
echo "[$(( $(date +%s) - t0 )) s] MBT_HAWKES_SOAK=2: 2 seeds x 2^17 lanes x 800 steps of BASELINE configs[3], default tier vs the float64 oracle" | tee -a "$OUT/soak.txt"
MBT_RESIDENT_STEP=1 MBT_FUZZ_SCALE=20 MBT_FUZZ_SEED=6500000 timeout 1500 python -m pytest tests/test_gpu_random_configs.py -q -n 8 -p no:cacheprovider 2>&1 | tail -3 | tee -a "$OUT/soak.txt"
echo "[$(( $(date +%s) - t0 )) s] MBT_RESIDENT_STEP=1 MBT_FUZZ_SCALE=20, eight processes in resident mode" | tee -a "$OUT/soak.txt"
MBT_FUZZ_SCALE=${GRAPH_SCALE:-50} MBT_FUZZ_SEED=6700000 timeout 1500 python -m pytest tests/test_gpu_graph_step.py -q -n 8 -k random_configurations -p no:cacheprovider 2>&1 | tail -3 | tee -a "$OUT/soak.txt"
echo "[$(( $(date +%s) - t0 )) s] MBT_FUZZ_SCALE=${GRAPH_SCALE:-50}: 40 x scale random configurations, the graph-capturable step against the ordinary loop over two episode ends (bit for bit)" | tee -a "$OUT/soak.txt"
