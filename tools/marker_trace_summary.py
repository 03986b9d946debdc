#!/usr/bin/env python3
"""Which launches belong to which phase of bench.py?  Reads the rocpd database of

    rocprofv3 --marker-trace --kernel-trace -d DIR -o bench -- python bench.py ...          (DIR/bench_results.db)

and prints, per roctx range bench.py pushed (bench.py: class phase), the kernels dispatched between the range's push and pop: calls, total and
average duration.  A dispatch is attributed by its START time on the device against the range's host interval - exact for the ranges bench.py
uses, each of which ends behind a synchronisation.

    python tools/marker_trace_summary.py gpurun_out/.../bench_results.db > profiles/rNN_bench_marker_trace.txt"""
import json
import sqlite3
import sys
from collections import defaultdict


def main(path):
    cur = sqlite3.connect(path).cursor()
    ranges = [(start, end, json.loads(ext or "{}").get("message", name)) for name, start, end, ext in
              cur.execute("select name, start, end, extdata from regions where category like 'MARKER%' order by start")]
    kernels = list(cur.execute("select name, start, end from kernels order by start"))
    print(f"{len(ranges)} roctx ranges, {len(kernels)} kernel dispatches ({path.split('/')[-1]})")
    inside = 0
    for start, end, message in ranges:
        by_name = defaultdict(lambda: [0, 0])
        for name, k_start, k_end in kernels:
            if start <= k_start < end:
                by_name[name][0] += 1
                by_name[name][1] += k_end - k_start
        calls = sum(v[0] for v in by_name.values())
        inside += calls
        print(f"\n== {message}: {(end - start) * 1e-6:.1f} ms on the host, {calls} dispatches, {sum(v[1] for v in by_name.values()) * 1e-6:.1f} ms of kernel time")
        for name, (n, total) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"   {n:7d} x {total / n * 1e-3:9.2f} us  {name[:150]}")
        if len(by_name) > 8:
            print(f"   ... and {len(by_name) - 8} more kernels")
    print(f"\n{len(kernels) - inside} dispatches outside every range (resets, construction, the blocks bench.py skips under a tracer)")


if __name__ == "__main__":
    main(sys.argv[1])
