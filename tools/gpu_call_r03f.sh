#!/bin/bash
# round 3, call f: learned-policy rollout with the network streamed from LDS
set -u
OUT=gpurun_out/r03f; mkdir -p "$OUT"
python tools/bench_policy.py > "$OUT/policy_rollout.json" 2> "$OUT/policy_rollout.err"; head -40 "$OUT/policy_rollout.json"
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_policy_gradient.py -x -q 2>&1 | tail -5 | tee "$OUT/pytest_policy.txt"
