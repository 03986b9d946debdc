#!/bin/bash
set -u
OUT=gpurun_out/r05s; mkdir -p "$OUT"; export TMPDIR=/tmp
t0=$(date +%s)
MBT_HAWKES_SOAK=10 timeout 1200 python -m pytest tests/test_gpu_round5.py -q -k soak -p no:cacheprovider 2>&1 | tail -4 | tee "$OUT/hawkes_soak10.txt"
echo "[$(( $(date +%s) - t0 )) s] MBT_HAWKES_SOAK=10: seeds 50..59 x 2^17 lanes x 800 steps = 1.05e9 lane-steps of BASELINE configs[3], default tier vs the float64 oracle" | tee -a "$OUT/hawkes_soak10.txt"
