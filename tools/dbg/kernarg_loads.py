#!/usr/bin/env python3
"""Which bytes of its kernel-argument segment does a kernel read?  Lists the scalar loads from the kernarg pointer (s[0:1] on entry) of the kernels
whose demangled name contains the argument, and the 64-byte lines they touch - the cost of a launch grows with the number of distinct lines
(tools/microbench/mb_lanes_per_thread.hip: 5.9 us with one line, 6.4-6.5 with eight or more, the same kernel).  No GPU needed.

    python tools/dbg/kernarg_loads.py 'mbt::step_kernel<mbt::Variant<mbt::shape::brownian, mbt::shape::pnl>, false, false>'"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import isa_stats  # noqa: E402

WIDTH = {"dword": 4, "dwordx2": 8, "dwordx4": 16, "dwordx8": 32, "dwordx16": 64}


def kernarg_reads(pattern, keep=None):
    """{demangled kernel name: (sorted [(offset, bytes)], sorted [64-byte line])} for the kernels whose name contains `pattern`."""
    keep = keep or os.path.join(ROOT, "build", "isa")
    os.makedirs(keep, exist_ok=True)
    out = {}
    for co in isa_stats.code_objects(keep):
        meta = isa_stats.kernel_metadata(co)
        names = isa_stats.demangle(list(meta))
        for symbol, pretty in names.items():
            if pattern not in pretty:
                continue
            text = subprocess.run([isa_stats.LLVM + "/llvm-objdump", "-d", f"--disassemble-symbols={symbol}", co], check=True, capture_output=True, text=True).stdout
            loads = []
            for line in text.splitlines():
                m = re.match(r"\s+s_load_(dword(?:x\d+)?)\s+\S+,\s*s\[0:1\],\s*(0x[0-9a-f]+|\d+)", line)
                if m:
                    loads.append((int(m.group(2), 0), WIDTH[m.group(1)]))
            out[pretty] = (sorted(loads), sorted({b // 64 for off, w in loads for b in range(off, off + w)}))
    return out


def main(pattern):
    for pretty, (loads, lines) in kernarg_reads(pattern).items():
        print(f"{pretty}\n   {len(loads)} loads from the kernarg segment, {sum(w for _, w in loads)} bytes, {len(lines)} distinct 64-byte lines: {lines}")
        print("   offsets: " + ", ".join(f"{off}+{w}" for off, w in loads))


if __name__ == "__main__":
    main(sys.argv[1])
