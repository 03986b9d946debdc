#!/bin/bash
# round 5, first GPU call: the split library, the exact-intensity tier, the whole suite, per-config timings
set -u
OUT=gpurun_out/r05a; mkdir -p "$OUT"; export TMPDIR=/tmp
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.txt"; }
python -c "import torch" 2>/dev/null; stamp "torch imported"
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q > "$OUT/pytest_new.log" 2>&1; stamp "new tests rc=$?"
tail -15 "$OUT/pytest_new.log"
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_round5.py --deselect tests/test_gpu_parity.py > "$OUT/pytest_all.log" 2>&1; stamp "suite rc=$?"
tail -15 "$OUT/pytest_all.log"
MBT_BENCH_STEPS=1000 timeout 300 python tests/perf/bench_configs.py > "$OUT/step_kernel_all_configs.json" 2> "$OUT/configs.err"; stamp "all configs rc=$?"
cat "$OUT/step_kernel_all_configs.json"
