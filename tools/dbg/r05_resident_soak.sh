#!/bin/bash
set -u
OUT=gpurun_out/r05s; mkdir -p "$OUT"; export TMPDIR=/tmp
t0=$(date +%s)
MBT_RESIDENT_STEP=1 MBT_FUZZ_SCALE=${SCALE:-60} MBT_FUZZ_SEED=${SEED:-3000000} timeout 1500 python -m pytest tests/test_gpu_random_configs.py -q -n ${PROCS:-8} -p no:cacheprovider ${PYTEST_ARGS:-} > "$OUT/resident_soak_full.log" 2>&1
tail -4 "$OUT/resident_soak_full.log" | tee "$OUT/resident_soak.txt"
echo "[$(( $(date +%s) - t0 )) s] MBT_RESIDENT_STEP=1 MBT_FUZZ_SCALE=${SCALE:-60} MBT_FUZZ_SEED=${SEED:-3000000}, pytest -n ${PROCS:-8} ${PYTEST_ARGS:-}" | tee -a "$OUT/resident_soak.txt"
