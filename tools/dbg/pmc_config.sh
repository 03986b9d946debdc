#!/bin/bash
# Shader-side counters of the step kernel of ONE case of tests/perf/bench_configs.py (substring of its name):
#   gpurun -- 'bash tools/dbg/pmc_config.sh "cfg1 AS 2^20, precise" as_precise'  ->  gpurun_out/dbg/pmc_config_as_precise.txt
set -u
ONLY=$1; TAG=$2; OUT=gpurun_out/dbg; mkdir -p "$OUT"; ROOT=$(pwd); export TMPDIR=/tmp
RES="$OUT/pmc_config_$TAG.txt"; echo "# $ONLY" > "$RES"
# (one small group per pass; MBT_PMC_EXTRA="A B|C D" appends groups, e.g. the LDS counters)
PMC_GROUPS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_TRANS SQ_INSTS_VMEM" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE")
if [ -n "${MBT_PMC_EXTRA:-}" ]; then IFS='|' read -ra MORE <<< "$MBT_PMC_EXTRA"; PMC_GROUPS+=("${MORE[@]}"); fi
for GROUP in "${PMC_GROUPS[@]}"; do
  D=/tmp/prof_cfg_$$_$(echo $GROUP | tr ' ' '_' | cut -c1-40); rm -rf "$D"
  (cd /tmp && MBT_BENCH_ONLY="$ONLY" MBT_BENCH_STEPS=40 MBT_BENCH_WARMUP=10 timeout 300 rocprofv3 --pmc $GROUP --output-format csv -d "$D" -- python "$ROOT/tests/perf/bench_configs.py" > /dev/null 2> "$D.err") || { echo "# group '$GROUP' failed: $(tail -1 $D.err)" >> "$RES"; continue; }
  F=$(find "$D" -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && python - "$F" >> "$RES" <<'PY'
import collections, csv, sys
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "step" in row["Kernel_Name"] and "kernel" in row["Kernel_Name"] and "reset" not in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, v in acc.items():
    print(f"{name:28s} mean per launch {sum(v) / len(v):18.1f}   ({len(v)} launches)")
PY
done
cat "$RES"
