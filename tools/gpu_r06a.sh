#!/bin/bash
# round 6, first GPU call: the graph-capturable step (tests, the device-loop bench), then the whole GPU suite
set -u
OUT=gpurun_out/r06a
mkdir -p "$OUT"
export TMPDIR=/tmp
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.txt"; }
python -c "import torch" 2>/dev/null; stamp "torch imported"
timeout 900 python -m pytest tests/test_gpu_graph_step.py tests/test_gpu_zero_copy.py -x -q > "$OUT/pytest_graph.log" 2>&1; stamp "graph tests rc=$?"
tail -15 "$OUT/pytest_graph.log"
timeout 600 python tools/bench_device_loop.py > "$OUT/device_loop.json" 2> "$OUT/device_loop.err"; stamp "device loop rc=$?"
cat "$OUT/device_loop.json"
tail -5 "$OUT/device_loop.err"
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_graph_step.py > "$OUT/pytest_all.log" 2>&1; stamp "all gpu tests rc=$?"
tail -8 "$OUT/pytest_all.log"
