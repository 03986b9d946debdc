#!/bin/bash
# Distribution of the step kernel's dispatch-to-completion time and of the gaps between consecutive dispatches under
# rocprofv3 --kernel-trace (the tracer serialises dispatches): gpurun_out/dbg/kernel_trace_hist.txt
set -u
OUT=gpurun_out/dbg; mkdir -p "$OUT"; ROOT=$(pwd); export TMPDIR=/tmp
D=/tmp/prof_kt_$$; rm -rf "$D"
(cd /tmp && rocprofv3 --kernel-trace ${MBT_KT_STATS:-} --output-format csv -d "$D" -- python "$ROOT/bench.py" --no-cpu-baseline --no-hbm-resident --no-rollout --steps 4000 --warmup 500 ${MBT_KT_ARGS:-} > "$ROOT/$OUT/kernel_trace_bench.json" 2> "$ROOT/$OUT/kernel_trace.err")
F=$(find "$D" -name '*kernel_trace.csv' | head -1)
python - "$F" > "$OUT/kernel_trace_hist.txt" <<'PY'
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "step_kernel" in r["Kernel_Name"]]
s = np.array([int(r["Start_Timestamp"]) for r in rows]); e = np.array([int(r["End_Timestamp"]) for r in rows])
o = np.argsort(s); s, e = s[o], e[o]
d = (e - s) / 1e3; gap = (s[1:] - e[:-1]) / 1e3
print(f"{len(d)} launches; duration us: mean {d.mean():.2f} median {np.median(d):.2f} p10 {np.percentile(d,10):.2f} p90 {np.percentile(d,90):.2f} min {d.min():.2f} max {d.max():.2f}")
print(f"gap between consecutive launches us: mean {gap.mean():.2f} median {np.median(gap):.2f} p10 {np.percentile(gap,10):.2f} p90 {np.percentile(gap,90):.2f}")
h, edges = np.histogram(d, bins=np.arange(4.5, 15.1, 0.5))
for c, lo in zip(h, edges): print(f"  {lo:5.1f}-{lo+0.5:4.1f} us {c:6d} {'#' * int(60 * c / max(1, h.max()))}")
tail = d[-4000:]
print(f"last 4000 (the timed region): mean {tail.mean():.2f} median {np.median(tail):.2f}")
# does the duration depend on the gap before it?
g = gap[-4000:]; dd = d[-4000:]
for lo, hi in ((0, 3), (3, 4), (4, 5), (5, 8), (8, 1e9)):
    m = (g >= lo) & (g < hi)
    if m.sum(): print(f"  gap in [{lo}, {hi}) us: {m.sum():5d} launches, mean duration {dd[m].mean():.2f}")
PY
cat "$OUT/kernel_trace_hist.txt"
