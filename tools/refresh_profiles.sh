#!/bin/bash
# Regenerates every artefact under profiles/ in ONE GPU-box call:
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r01'   then   cp gpurun_out/profiles_r01/* profiles/
# rocprofv3 passes: kernel trace + stats alone; FETCH_SIZE and WRITE_SIZE each in its own --pmc pass (the MI355X guide's
# HBM recipe); never combined with hip/hsa/sys tracing.
set -u
TAG=${1:-r01}
OUT=gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)

python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.stderr"

# the same command under rocprofv3 (kernel trace + stats)
rm -rf /tmp/prof_stats && (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python "$ROOT/bench.py" --no-cpu-baseline > "$ROOT/$OUT/${TAG}_bench_under_rocprof.json" 2> "$ROOT/$OUT/rocprof_stats.stderr")
find /tmp/prof_stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/${TAG}_bench_kernel_stats.csv"

# PMC: one counter per pass, short run (every launch is serialised by the profiler)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C && (cd /tmp && rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$C -- python "$ROOT/bench.py" --no-cpu-baseline --steps 200 --warmup 20 > /dev/null 2> "$ROOT/$OUT/rocprof_$C.stderr")
done
F=$(find /tmp/prof_FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find /tmp/prof_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py "$F" "$W" "$OUT/${TAG}_pmc_summary.json" > /dev/null

python tests/perf/parity_report.py > "$OUT/${TAG}_parity_report.txt" 2> /dev/null
MBT_BENCH_STEPS=1000 python tests/perf/bench_configs.py > "$OUT/${TAG}_step_kernel_all_configs.json" 2> /dev/null
python tools/bench_rollout.py > "$OUT/${TAG}_rollout_kernel.json" 2> /dev/null
python tests/perf/bench_host_path.py > "$OUT/${TAG}_host_path.json" 2> /dev/null

make -C tools/microbench > /dev/null 2>&1
{
  for n in 20 22 24; do echo "== mb_copy $n (buffers of zeros)"; tools/microbench/mb_copy $n; done
  for n in 20 22 24; do echo "== mb_copy $n, random data"; MB_RANDOM_DATA=1 tools/microbench/mb_copy $n; done
  for n in 18 20 22 24; do echo "== mb_step $n"; tools/microbench/mb_step $n; done
  for n in 20 22 24; do echo "== mb_rows6 $n"; tools/microbench/mb_rows6 $n; done
} > "$OUT/${TAG}_microbench_raw.txt" 2>&1
ls -la "$OUT"
