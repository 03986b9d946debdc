#!/bin/bash
set -u
OUT=gpurun_out/r05n; mkdir -p "$OUT"; export TMPDIR=/tmp
t0=$(date +%s)
MBT_FUZZ_SCALE=100 MBT_FUZZ_SEED=900000 timeout 2400 python -m pytest tests/test_gpu_random_configs.py -q -n 12 -p no:cacheprovider 2>&1 | tail -6 | tee "$OUT/bigsoak.txt"
echo "[$(( $(date +%s) - t0 )) s] MBT_FUZZ_SCALE=100 MBT_FUZZ_SEED=900000: 15000 default-tier + 3000 float32-intensity + 9000 precise + 6000 + 6000 speed + 6000 rollout configurations" | tee -a "$OUT/bigsoak.txt"
