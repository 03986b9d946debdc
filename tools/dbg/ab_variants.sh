#!/bin/bash
# Library variants (mbt_gym_amd/libmbtenv.<name>.so, tools/dbg/build_variant.py; "main" = the shipped library) timed ALTERNATELY, one process per
# measurement, medians over ROUNDS: a single pair of runs cannot tell 2 % apart (the same binary moves by +-0.15 us between processes at 2^20 lanes).
#   ROUNDS=8 ONLY="2^20" bash tools/dbg/ab_variants.sh oldlayout main b2      ->  a table of medians (us per step) per case of tests/perf/bench_configs.py
set -u
OUT=gpurun_out/dbg; mkdir -p "$OUT"; LOG="$OUT/ab_variants.jsonl"; : > "$LOG"
for r in $(seq 1 ${ROUNDS:-6}); do
  for v in "$@"; do
    lib=$v; [ "$v" = main ] && lib=""
    MBT_LIBRARY_VARIANT=$lib MBT_BENCH_STEPS=${STEPS:-2000} MBT_BENCH_ONLY="${ONLY:-2^20}" python tests/perf/bench_configs.py 2>/dev/null | python -c "
import json, sys
d = json.load(sys.stdin)
print(json.dumps({'variant': '$v', 'us': {k: v['us_per_step'] for k, v in d.items()}}))" >> "$LOG"
  done
done
python - "$LOG" "$@" <<'PY'
import json, statistics, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
variants = sys.argv[2:]
cases = list(rows[0]["us"])
print(f"{'case':60s} " + " ".join(f"{v:>22s}" for v in variants))
for c in cases:
    cells = []
    for v in variants:
        xs = [r["us"][c] for r in rows if r["variant"] == v and c in r["us"]]
        cells.append(f"{statistics.median(xs):7.2f} [{min(xs):.2f}-{max(xs):.2f}]" if xs else "-")
    print(f"{c[:60]:60s} " + " ".join(f"{x:>22s}" for x in cells))
PY
