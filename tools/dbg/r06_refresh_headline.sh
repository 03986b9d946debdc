#!/bin/bash
# the three steps of tools/refresh_profiles.sh that produce the headline's committed summary and lines (after a change to bench.py's blocks)
set -u
TAG=r06; OUT=gpurun_out/profiles_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; ROOT=$(pwd)
rm -rf /tmp/prof_main && (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_main -- python "$ROOT/bench.py" --no-cpu-baseline > "$ROOT/$OUT/${TAG}_bench_under_rocprof.json" 2> "$ROOT/$OUT/rocprof_main.stderr")
find /tmp/prof_main -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/${TAG}_bench_kernel_stats.csv"
cp "$OUT/${TAG}_bench_kernel_stats.csv" profiles/
python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.stderr"; echo "bench default rc=$?"
python bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_args.json" 2>> "$OUT/bench.stderr"; echo "bench driver args rc=$?"
grep "step_kernel<mbt::Variant<mbt::shape::brownian, mbt::shape::pnl>, false, false>" "$OUT/${TAG}_bench_kernel_stats.csv"
