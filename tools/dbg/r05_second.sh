#!/bin/bash
set -u
OUT=gpurun_out/r05b; mkdir -p "$OUT"; export TMPDIR=/tmp
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.txt"; }
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_round4.py -x -q > "$OUT/pytest_new.log" 2>&1; stamp "new tests rc=$?"
tail -8 "$OUT/pytest_new.log"
ab() { echo "== $*"; env "$@" MBT_BENCH_STEPS=1500 MBT_BENCH_ONLY="exact intensities" python tests/perf/bench_configs.py 2>/dev/null | grep us_per_step; }
ab A=1 | tee -a "$OUT/ab.txt"
ab MBT_STREAM_LOADS=1 | tee -a "$OUT/ab.txt"
ab MBT_STEP_DYNAMIC_LDS=8192 | tee -a "$OUT/ab.txt"
ab MBT_STEP_DYNAMIC_LDS=16384 | tee -a "$OUT/ab.txt"
ab MBT_STEP_DYNAMIC_LDS=24576 | tee -a "$OUT/ab.txt"
ab MBT_STREAM_LOADS=1 MBT_STEP_DYNAMIC_LDS=16384 | tee -a "$OUT/ab.txt"
ab A=2 | tee -a "$OUT/ab.txt"
stamp "ab"
