#!/bin/bash
set -u
OUT=gpurun_out/r05g; mkdir -p "$OUT"; export TMPDIR=/tmp
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.txt"; }
timeout 1500 python -m pytest tests -m gpu -q -x > "$OUT/pytest_all.log" 2>&1; rc=$?; stamp "suite rc=$rc"
grep -v "Warning\|warn\|^$\|return TradingEnvironment\|amdgpu.ids" "$OUT/pytest_all.log" | tail -25
MBT_BENCH_STEPS=1500 python tests/perf/bench_configs.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print('   %-90s %7.2f us  moved %5.0f GB/s  credited frac %.3f' % (k[:90], v['us_per_step'], v['moved_GBps'], v['credited_frac_of_8TBps']))" | tee "$OUT/configs.txt"
stamp "configs"
