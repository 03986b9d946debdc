#!/bin/bash
set -u
OUT=gpurun_out/r05e; mkdir -p "$OUT"; export TMPDIR=/tmp
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.txt"; }
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > "$OUT/pytest_r5.log" 2>&1; stamp "round5 tests rc=$?"
tail -5 "$OUT/pytest_r5.log"
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_round5.py > "$OUT/pytest_all.log" 2>&1; rc=$?; stamp "suite rc=$rc"
tail -5 "$OUT/pytest_all.log"
bash tools/refresh_profiles.sh r05 > "$OUT/refresh.log" 2>&1; stamp "refresh rc=$?"
tail -30 "$OUT/refresh.log"
