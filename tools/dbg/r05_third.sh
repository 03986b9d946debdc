#!/bin/bash
set -u
OUT=gpurun_out/r05c; mkdir -p "$OUT"; export TMPDIR=/tmp
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.txt"; }
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_rollout.py -x -q > "$OUT/pytest_new.log" 2>&1; stamp "new tests rc=$?"
tail -8 "$OUT/pytest_new.log"
timeout 300 tools/microbench/mb_floor record > "$OUT/mb_floor_record.txt" 2>&1; stamp "floor record rc=$?"
cat "$OUT/mb_floor_record.txt"
for p in 0 1 2; do timeout 120 tools/microbench/mb_rollout_p$p >> "$OUT/mb_rollout.txt" 2>&1; done; stamp "mb_rollout"
cat "$OUT/mb_rollout.txt"
