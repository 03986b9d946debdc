#!/bin/bash
set -u
TAG=${1:-r03d}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.txt"; }
python -c "import torch" 2>/dev/null; stamp "torch imported"
timeout 1200 python -m pytest tests/test_gpu_policy.py tests/test_gpu_policy_gradient.py tests/test_gpu_multi_device.py tests/test_user_plugins.py tests/test_gpu_round2.py tests/test_gpu_host_buffers.py tests/test_gpu_speed.py -m gpu -q > "$OUT/pytest.log" 2>&1; stamp "pytest subset rc=$?"
for i in 1 2 3; do
MBT_BENCH_STEPS=2000 timeout 600 python tests/perf/bench_configs.py > "$OUT/step_kernel_all_configs_$i.json" 2> "$OUT/all_configs.err"; stamp "all configs $i rc=$?"
done
tail -8 "$OUT/pytest.log"
