#!/bin/bash
# GPU call r03b: full suite, host-path split, bench lines (driver args, default, 2 gloo ranks, forced-distributed), all configs + rocprof stats, floors.
set -u
TAG=${1:-r03b}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$OUT/timeline.txt"; }
python -c "import torch" 2>/dev/null; stamp "torch imported"
timeout 1200 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; stamp "pytest all rc=$?"
timeout 600 python tests/perf/bench_host_path.py > "$OUT/host_path.json" 2> "$OUT/host_path.err"; stamp "host path rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-resident > "$OUT/bench_driver_args_nobase.json" 2> "$OUT/bench_driver_args.err"; stamp "bench driver args (no baselines) rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_args.json" 2>> "$OUT/bench_driver_args.err"; stamp "bench driver args rc=$?"
timeout 600 python bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; stamp "bench default rc=$?"
timeout 300 python bench.py --gpus 1 --force-distributed --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-resident > "$OUT/bench_forced_distributed.json" 2> "$OUT/bench_forced_distributed.err"; stamp "bench forced distributed rc=$?"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-distributed --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-resident > "$OUT/bench_torchrun_1.json" 2> "$OUT/bench_torchrun_1.err"; stamp "bench under torchrun (1 rank, forced distributed) rc=$?"
MBT_BENCH_STEPS=1000 timeout 600 python tests/perf/bench_configs.py > "$OUT/step_kernel_all_configs.json" 2> "$OUT/all_configs.err"; stamp "all configs rc=$?"
rm -rf /tmp/prof_cfg && (cd /tmp && MBT_BENCH_STEPS=2000 MBT_BENCH_WARMUP=200 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg -- python "$ROOT/tests/perf/bench_configs.py" > /dev/null 2> "$ROOT/$OUT/rocprof_all_configs.err")
find /tmp/prof_cfg -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/all_configs_kernel_stats.csv"; stamp "rocprof all configs"
make -C tools/microbench mb_floor > /dev/null 2>&1
timeout 300 tools/microbench/mb_floor > "$OUT/mb_floor.txt" 2>&1; stamp "mb_floor rc=$?"
tail -12 "$OUT/pytest.log"
cut -c1-900 "$OUT/bench_driver_args_nobase.json"
