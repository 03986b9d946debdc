#!/bin/bash
# HBM-side traffic (FETCH_SIZE x2 + WRITE_SIZE, one counter per rocprofv3 pass) of the step kernel of EVERY config in
# tests/perf/bench_configs.py:   gpurun -- 'bash tools/pmc_all_configs.sh r01'  ->  gpurun_out/profiles_r01/r01_pmc_all_configs.json
set -u
TAG=${1:-r01}; OUT=gpurun_out/profiles_$TAG; mkdir -p "$OUT"; ROOT=$(pwd); export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pa_$C && (cd /tmp && MBT_BENCH_STEPS=40 MBT_BENCH_WARMUP=10 rocprofv3 --pmc $C --output-format csv -d /tmp/pa_$C -- python "$ROOT/tests/perf/bench_configs.py" > /dev/null 2> "$ROOT/$OUT/pmc_all_$C.stderr")
done
F=$(find /tmp/pa_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find /tmp/pa_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py "$F" "$W" "$OUT/${TAG}_pmc_all_configs.json" step_kernel
