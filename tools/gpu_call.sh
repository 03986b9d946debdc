#!/bin/bash
# One GPU-box call of the round: tests, the driver's bench line, the default bench line, floors and regimes.
#   gpurun --timeout 1500 -- 'bash tools/gpu_call.sh r02a'
set -u
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.txt"; }
python -c "import torch" 2>/dev/null; stamp "torch imported"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; stamp "pytest rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_args.json" 2> "$OUT/bench_driver_args.err"; stamp "bench driver args rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-resident > "$OUT/bench_driver_args_2.json" 2>> "$OUT/bench_driver_args.err"; stamp "bench driver args (2) rc=$?"
timeout 400 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; stamp "bench default rc=$?"
make -C tools/microbench mb_floor > /dev/null 2>&1
timeout 300 tools/microbench/mb_floor > "$OUT/mb_floor.txt" 2>&1; stamp "mb_floor rc=$?"
timeout 600 python tests/perf/bench_regimes.py > "$OUT/regimes.json" 2> "$OUT/regimes.err"; stamp "regimes rc=$?"
MBT_BENCH_STEPS=1000 timeout 300 python tests/perf/bench_configs.py > "$OUT/step_kernel_all_configs.json" 2> /dev/null; stamp "all configs rc=$?"
tail -5 "$OUT/pytest.log"
cat "$OUT/bench_driver_args.json" | cut -c1-600
