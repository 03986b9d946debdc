#!/bin/bash
set -u
OUT=gpurun_out/r05d; mkdir -p "$OUT"; export TMPDIR=/tmp
for round in 1 2; do
timeout 300 tools/microbench/mb_floor record >> "$OUT/mb_floor_record.txt" 2>&1
for p in 0 1 2 3 4; do timeout 120 tools/microbench/mb_rollout_p$p >> "$OUT/mb_rollout.txt" 2>&1; done
done
cat "$OUT/mb_floor_record.txt" "$OUT/mb_rollout.txt"
