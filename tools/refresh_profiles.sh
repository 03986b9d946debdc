#!/bin/bash
# Regenerates every artefact under profiles/ in ONE GPU-box call:
#   gpurun --timeout 1800 -- 'bash tools/refresh_profiles.sh r05'   then   cp gpurun_out/profiles_r05/r05_* profiles/
# rocprofv3 passes: kernel trace + stats alone; FETCH_SIZE and WRITE_SIZE each in its own --pmc pass (the MI355X guide's
# HBM recipe); never combined with hip/hsa/sys tracing.
set -u
TAG=${1:-r05}
OUT=gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.txt"; }

# What bench.py QUOTES from committed files comes first and is put where bench.py looks for it (profiles/ of this scratch copy),
# so that the lines below carry this round's figures: the write-only floors of the rollout's recording (mb_floor), the shader
# counters of the returns-only rollout (one small --pmc group per pass), and - further down - the kernel-stats summary.
make -C tools/microbench > /dev/null 2>&1
tools/microbench/mb_floor > "$OUT/${TAG}_floors.txt" 2>&1; stamp "floors"
cp "$OUT/${TAG}_floors.txt" profiles/
python tools/pmc_rollout_summary.py "$OUT/${TAG}_pmc_rollout.json" > /dev/null 2> "$OUT/pmc_rollout.stderr"; stamp "pmc rollout rc=$?"
cp "$OUT/${TAG}_pmc_rollout.json" profiles/ 2>/dev/null
for p in 0 1 2 3 4; do tools/microbench/mb_rollout_p$p; done > "$OUT/${TAG}_mb_rollout.txt" 2>&1; stamp "rollout store policies"

# THE summary of the contract: `rocprofv3 --kernel-trace --stats` of bench.py at its default arguments (minus the CPU baseline,
# which launches nothing) - the headline kernel, the same kernel at 2^24 lanes (its STREAM instantiation) and the eight kernels
# of the per-configuration block, each with its own row.  bench.py sees the tracer (ROCP_TOOL_LIBRARIES) and enqueues its launches
# in bursts behind a gate kernel, so the traced kernels run back to back like the untraced ones (include/mbt_env.h:
# mbt_env_set_launch_gate); MBT_BENCH_GATE=0 gives the ungated trace for comparison.
rm -rf /tmp/prof_main && (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_main -- python "$ROOT/bench.py" --no-cpu-baseline > "$ROOT/$OUT/${TAG}_bench_under_rocprof.json" 2> "$ROOT/$OUT/rocprof_main.stderr")
find /tmp/prof_main -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/${TAG}_bench_kernel_stats.csv"; stamp "kernel stats of bench.py (gated launches)"
cp "$OUT/${TAG}_bench_kernel_stats.csv" profiles/
# the bench lines: default arguments, and the driver's (--steps 20 --warmup 5) - after the summary above, which they quote (frac_rocprof)
python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.stderr"; stamp "bench default rc=$?"
python bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_args.json" 2>> "$OUT/bench.stderr"; stamp "bench driver args rc=$?"
rm -rf /tmp/prof_ungated && (cd /tmp && MBT_BENCH_GATE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ungated -- python "$ROOT/bench.py" --no-cpu-baseline --no-hbm-resident --no-configs --no-rollout --no-device-loop > "$ROOT/$OUT/${TAG}_bench_under_rocprof_ungated.json" 2>> "$ROOT/$OUT/rocprof_main.stderr")
find /tmp/prof_ungated -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/${TAG}_bench_kernel_stats_ungated.csv"; stamp "kernel stats of bench.py (ungated, for comparison)"
# which launches belong to which phase of bench.py: its roctx ranges (bench.py: class phase) beside the kernel trace, at the driver's arguments
rm -rf /tmp/prof_marker && (cd /tmp && rocprofv3 --marker-trace --kernel-trace -d /tmp/prof_marker -o bench -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>> "$ROOT/$OUT/rocprof_main.stderr")
find /tmp/prof_marker -name 'bench_results.db' | head -1 | xargs -I{} python tools/marker_trace_summary.py {} | cut -c1-220 > "$OUT/${TAG}_bench_marker_trace.txt"; stamp "marker trace of bench.py"
cp "$OUT/${TAG}_bench_marker_trace.txt" profiles/

profile() {  # profile <name> <bench args...>: kernel trace + stats, then the two PMC passes
  local name=$1; shift
  if [ "$name" != bench ]; then
  rm -rf /tmp/prof_stats && (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python "$ROOT/bench.py" --no-cpu-baseline --no-hbm-resident --no-configs --no-rollout --no-device-loop "$@" > "$ROOT/$OUT/${TAG}_${name}_under_rocprof.json" 2> "$ROOT/$OUT/rocprof_${name}.stderr")
  find /tmp/prof_stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/${TAG}_${name}_kernel_stats.csv"
  fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_$C && (cd /tmp && rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$C -- python "$ROOT/bench.py" --no-cpu-baseline --no-hbm-resident --no-configs --no-rollout --no-device-loop --prewarm-steps 0 "${@:1:2}" --steps 200 --warmup 20 > /dev/null 2> "$ROOT/$OUT/rocprof_${name}_$C.stderr")
  done
  F=$(find /tmp/prof_FETCH_SIZE -name '*counter_collection.csv' | head -1)
  W=$(find /tmp/prof_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  python tools/pmc_summary.py "$F" "$W" "$OUT/${TAG}_${name}_pmc_summary.json" > /dev/null
  stamp "profile $name"
}
# the headline workload (2^20 lanes, Infinity-Cache resident) and the same kernel where every byte crosses HBM (2^24 lanes)
profile bench --lanes 1048576
profile hbm_resident --lanes 16777216 --steps 600 --warmup 100
cp "$OUT/${TAG}_bench_pmc_summary.json" "$OUT/${TAG}_pmc_summary.json"   # the name bench.py's roofline.traffic reads

# per-dispatch durations and gaps under the tracer: gated launches (what the summary above is made of) and ungated ones (the host becomes the bottleneck)
MBT_KT_STATS=--stats MBT_KT_ARGS=--no-configs bash tools/dbg/kernel_trace_hist.sh > /dev/null 2>&1; cp gpurun_out/dbg/kernel_trace_hist.txt "$OUT/${TAG}_bench_kernel_trace_hist_gated.txt"
MBT_BENCH_GATE=0 MBT_KT_STATS=--stats MBT_KT_ARGS=--no-configs bash tools/dbg/kernel_trace_hist.sh > /dev/null 2>&1; cp gpurun_out/dbg/kernel_trace_hist.txt "$OUT/${TAG}_bench_kernel_trace_hist_ungated.txt"; stamp "kernel trace histograms"
python tests/perf/parity_report.py > "$OUT/${TAG}_parity_report.txt" 2> /dev/null; stamp "parity report"
MBT_BENCH_STEPS=1000 python tests/perf/bench_configs.py > "$OUT/${TAG}_step_kernel_all_configs.json" 2> /dev/null; stamp "all configs"
python tests/perf/bench_regimes.py > "$OUT/${TAG}_regimes.json" 2> /dev/null; stamp "regimes"
python tools/bench_rollout.py > "$OUT/${TAG}_rollout_kernel.json" 2> /dev/null; stamp "rollout"
python tools/bench_policy.py > "$OUT/${TAG}_policy_rollout.json" 2> /dev/null; stamp "policy rollout"
python tests/perf/bench_host_path.py > "$OUT/${TAG}_host_path.json" 2> /dev/null; stamp "host path"
python tests/dbg/gym_loop_breakdown.py > "$OUT/${TAG}_gym_loop_breakdown.json" 2> /dev/null; stamp "gym loop breakdown"
# the multi-rank code path of bench.py with a world of one (RCCL communicator through the C ABI, collective check): what an 8-GPU run adds
python bench.py --gpus 1 --force-distributed --no-cpu-baseline --no-hbm-resident --no-configs --no-rollout --no-device-loop > "$OUT/${TAG}_bench_forced_distributed.json" 2>> "$OUT/bench.stderr"; stamp "bench forced distributed rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --force-distributed --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-resident --no-configs --no-rollout --no-device-loop > "$OUT/${TAG}_bench_torchrun_world1.json" 2>> "$OUT/bench.stderr"; stamp "bench under torchrun rc=$?"
python tests/perf/bench_timed_region.py > "$OUT/${TAG}_timed_region.json" 2> /dev/null; stamp "timed region"

# per-kernel statistics of the other kernel families (every BASELINE config's step kernel, the fused rollouts, the learned
# policies) and the HBM-side traffic of every config's step kernel
stats() {  # stats <name> <script>
  rm -rf /tmp/prof_$1 && (cd /tmp && MBT_BENCH_STEPS=2000 MBT_BENCH_WARMUP=500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -- python "$ROOT/$2" > /dev/null 2> "$ROOT/$OUT/rocprof_$1.stderr")
  find /tmp/prof_$1 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/${TAG}_$1_kernel_stats.csv"
  stamp "kernel stats $1"
}
stats all_configs tests/perf/bench_configs.py
stats rollout tools/bench_rollout.py
stats policy tools/bench_policy.py
bash tools/pmc_all_configs.sh "$TAG" > /dev/null 2>&1; stamp "pmc all configs"

tools/microbench/mb_sync > "$OUT/${TAG}_mb_sync.txt" 2>&1; stamp "sync latency"
ls -la "$OUT"
