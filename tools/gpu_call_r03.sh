#!/bin/bash
# One GPU-box call of round 3: the precise_state tier first (fast feedback), then the whole GPU suite, then whatever
# measurement scripts are named on the command line.
#   gpurun --timeout 1500 -- 'bash tools/gpu_call_r03.sh r03a [script ...]'
set -u
TAG=${1:-r03a}
shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.txt"; }
python -c "import torch" 2>/dev/null; stamp "torch imported"
timeout 600 python -m pytest tests/test_gpu_precise.py tests/test_user_plugins.py "tests/test_gpu_random_configs.py" -m gpu -q -k "precise or user" > "$OUT/pytest_precise.log" 2>&1; stamp "pytest precise rc=$?"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; stamp "pytest all rc=$?"
for script in "$@"; do
  name=$(basename "$script" .py)
  timeout 600 python "$script" > "$OUT/$name.json" 2> "$OUT/$name.err"; stamp "$name rc=$?"
done
tail -30 "$OUT/pytest_precise.log"
tail -15 "$OUT/pytest.log"
