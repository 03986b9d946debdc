#!/bin/bash
set -u
TAG=${1:-r03c}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.txt"; }
python -c "import torch" 2>/dev/null; stamp "torch imported"
timeout 1200 python -m pytest tests/test_gpu_speed.py tests/test_gpu_host_buffers.py tests/test_gpu_round2.py tests/test_gpu_random_configs.py tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_multi_device.py -m gpu -q -x > "$OUT/pytest.log" 2>&1; stamp "pytest subset rc=$?"
for i in 1 2; do
MBT_BENCH_STEPS=1000 timeout 600 python tests/perf/bench_configs.py > "$OUT/step_kernel_all_configs_$i.json" 2> "$OUT/all_configs.err"; stamp "all configs $i rc=$?"
done
for pin in 1 0 1 0; do
MBT_BENCH_PIN=$pin timeout 300 python bench.py --gpus 1 --force-distributed --steps 2000 --warmup 5 --no-cpu-baseline --no-hbm-resident > "$OUT/bench_forced_pin${pin}_$RANDOM.json" 2>> "$OUT/bench_forced.err"; stamp "forced distributed pin=$pin rc=$?"
done
timeout 300 python bench.py --steps 2000 --warmup 5 --no-cpu-baseline --no-hbm-resident > "$OUT/bench_plain_2000.json" 2>> "$OUT/bench_forced.err"; stamp "plain rc=$?"
timeout 300 python tests/dbg/gym_loop_breakdown.py > "$OUT/gym_loop_breakdown.json" 2> "$OUT/gym_loop_breakdown.err"; stamp "gym loop breakdown rc=$?"
tail -8 "$OUT/pytest.log"
