#!/bin/bash
set -u
OUT=gpurun_out/r05k; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_host_callbacks.py tests/test_user_plugins.py tests/test_gpu_precise.py -x -q > "$OUT/pytest.log" 2>&1; echo "host-callback tests rc=$?"
grep -v "Warning\|warn\|return TradingEnvironment\|^$" "$OUT/pytest.log" | tail -8
python - <<'PY' | tee "$OUT/callbacks.json"
import json, sys, os
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests", "perf"))
import bench_host_path
print(json.dumps(bench_host_path.host_callback_rows(), indent=1))
PY
