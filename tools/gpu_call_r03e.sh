#!/bin/bash
# round 3, call e: learned-policy rollout after the packed tanh: throughput, bit identity, shader counters
set -u
OUT=gpurun_out/r03e; mkdir -p "$OUT"
python tools/bench_policy.py > "$OUT/policy_rollout.json" 2> "$OUT/policy_rollout.err"; cat "$OUT/policy_rollout.json" | head -40
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_policy_gradient.py -x -q 2>&1 | tail -5 | tee "$OUT/pytest_policy.txt"
timeout 1500 bash tools/dbg/pmc_learned.sh > "$OUT/pmc.txt" 2>&1; tail -50 "$OUT/pmc.txt"
