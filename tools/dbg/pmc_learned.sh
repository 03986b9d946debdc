#!/bin/bash
# Shader-side counters of the fused rollout under a learned MLP policy (one small group per rocprofv3 pass; no tracing domains).
set -u
OUT=gpurun_out/dbg; mkdir -p "$OUT"; ROOT=$(pwd); export TMPDIR=/tmp
for ACT in tanh relu; do
  export MBT_ACT=$ACT
  RES="$OUT/pmc_learned_$ACT.txt"; : > "$RES"
  for GROUP in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_TRANS SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
    D=/tmp/prof_lr_$$_${ACT}_$(echo $GROUP | tr ' ' '_' | cut -c1-40); rm -rf "$D"
    (cd /tmp && timeout 300 rocprofv3 --pmc $GROUP --output-format csv -d "$D" -- python "$ROOT/tools/dbg/learned_once.py" > /dev/null 2> "$D.err") || { echo "# group '$GROUP' failed: $(tail -1 $D.err)" >> "$RES"; continue; }
    F=$(find "$D" -name '*counter_collection.csv' | head -1)
    [ -n "$F" ] && python - "$F" >> "$RES" <<'PY'
import collections, csv, sys
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "learned_rollout_kernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, v in acc.items():
    print(f"{name:28s} mean per launch {sum(v) / len(v):18.1f}   ({len(v)} launches)")
PY
  done
  echo "== $ACT"; cat "$RES"
done
