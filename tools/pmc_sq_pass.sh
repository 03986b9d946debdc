#!/bin/bash
# Shader-side counters of the benchmark's step kernel (one small group per rocprofv3 pass; no tracing domains):
#   gpurun -- 'bash tools/pmc_sq_pass.sh r01'  ->  gpurun_out/profiles_r01/r01_pmc_sq.txt
set -u
TAG=${1:-r01}
OUT=gpurun_out/profiles_$TAG
mkdir -p "$OUT"
ROOT=$(pwd)
export TMPDIR=/tmp
: > "$OUT/${TAG}_pmc_sq.txt"
for GROUP in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  D=/tmp/prof_sq_$$_$(echo $GROUP | tr ' ' '_' | cut -c1-40)
  rm -rf "$D"
  (cd /tmp && rocprofv3 --pmc $GROUP --output-format csv -d "$D" -- python "$ROOT/bench.py" --no-cpu-baseline --steps 100 --warmup 10 > /dev/null 2> "$D.err") || { echo "# group '$GROUP' failed: $(tail -1 $D.err)" >> "$OUT/${TAG}_pmc_sq.txt"; continue; }
  F=$(find "$D" -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && python - "$F" >> "$OUT/${TAG}_pmc_sq.txt" <<'PY'
import collections, csv, sys
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "step_kernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, v in acc.items():
    print(f"{name:24s} mean per launch {sum(v) / len(v):16.1f}   ({len(v)} launches of the AS step kernel, 2^20 lanes)")
PY
done
cat "$OUT/${TAG}_pmc_sq.txt"
