#!/bin/bash
# state updated in place vs the two-buffer scheme (mbt_env.hip: mbt_env::state; profiles/r05_in_place_state.txt)
set -u
OUT=gpurun_out/r05f; mkdir -p "$OUT"; export TMPDIR=/tmp
ab() { echo "== $*"; env "$@" MBT_BENCH_STEPS=1500 python tests/perf/bench_configs.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print('   %-90s %7.2f us  moved %5.0f GB/s' % (k[:90], v['us_per_step'], v['moved_GBps']))"; }
for i in 1 2 3; do
ab MBT_PING_PONG_STATE=1 MBT_BENCH_ONLY="cfg3" | tee -a "$OUT/ab.txt"
ab MBT_PING_PONG_STATE=0 MBT_BENCH_ONLY="cfg3" | tee -a "$OUT/ab.txt"
done
ab MBT_PING_PONG_STATE=1 MBT_BENCH_ONLY="cfg" | tee -a "$OUT/ab.txt"
ab MBT_PING_PONG_STATE=0 MBT_BENCH_ONLY="cfg" | tee -a "$OUT/ab.txt"
python tests/perf/bench_regimes.py > "$OUT/regimes_pingpong.json" 2>/dev/null
MBT_PING_PONG_STATE=0 python tests/perf/bench_regimes.py > "$OUT/regimes_inplace.json" 2>/dev/null
