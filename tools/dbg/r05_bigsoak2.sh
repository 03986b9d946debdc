#!/bin/bash
set -u
OUT=gpurun_out/r05s; mkdir -p "$OUT"; export TMPDIR=/tmp
t0=$(date +%s)
MBT_FUZZ_SCALE=250 MBT_FUZZ_SEED=2000000 timeout 3000 python -m pytest tests/test_gpu_random_configs.py -q -n 12 -p no:cacheprovider 2>&1 | tail -8 | tee "$OUT/bigsoak2.txt"
echo "[$(( $(date +%s) - t0 )) s] MBT_FUZZ_SCALE=250 MBT_FUZZ_SEED=2000000 (final round-5 kernels)" | tee -a "$OUT/bigsoak2.txt"
