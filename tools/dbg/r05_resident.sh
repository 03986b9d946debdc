#!/bin/bash
set -u
OUT=gpurun_out/r05j; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round5.py -x -q -k "resident" > "$OUT/pytest.log" 2>&1; echo "resident tests rc=$?"
tail -15 "$OUT/pytest.log"
timeout 200 python tests/perf/bench_host_path.py > "$OUT/host_path_plain.json" 2>/dev/null; echo "host path plain rc=$?"
MBT_RESIDENT_STEP=1 timeout 200 python tests/perf/bench_host_path.py > "$OUT/host_path_resident.json" 2>/dev/null; echo "host path resident rc=$?"
MBT_RESIDENT_STEP=1 MBT_RESIDENT_VRAM=0 timeout 200 python tests/perf/bench_host_path.py > "$OUT/host_path_resident_hostmem.json" 2>/dev/null; echo "host path resident (host memory mailbox) rc=$?"
python - <<'PY'
import json
for name in ("plain", "resident", "resident_hostmem"):
    try:
        d = json.load(open(f"gpurun_out/r05j/host_path_{name}.json"))
    except Exception as exc:
        print(name, "unreadable", exc); continue
    print(name, json.dumps({k: v for k, v in d.items() if "1000" in k or "small" in k.lower()}, indent=0)[:1500])
PY
