#!/bin/bash
set -u
OUT=gpurun_out/r06b
mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_graph_step.py tests/test_gpu_zero_copy.py -x -q > "$OUT/pytest_graph.log" 2>&1; echo "graph tests rc=$?"
tail -5 "$OUT/pytest_graph.log"
timeout 600 python tools/dbg/r06_captured_ab.py > "$OUT/captured_ab.txt" 2>&1; echo "ab rc=$?"
cat "$OUT/captured_ab.txt"
timeout 600 python tools/bench_device_loop.py > "$OUT/device_loop.json" 2> "$OUT/device_loop.err"; echo "device loop rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06b/device_loop.json"))
for r in d["rows"]:
    print(r["lanes"], {k:{kk:round(vv,2) for kk,vv in v.items() if kk.endswith("us_per_step")} for k,v in r.items() if isinstance(v,dict)})
PY
