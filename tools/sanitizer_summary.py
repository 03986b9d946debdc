#!/usr/bin/env python3
"""Sorts the reports of a sanitizer run (tools/run_sanitizers.sh) into those whose stack passes through libmbtenv - findings - and
the rest: the python interpreter, torch and the HIP / HSA runtimes are not instrumented and not this repository's to fix (e.g. ROCm's
ASan runtime asks the HSA runtime about every freed pointer and faults when the HIP runtime's exit handler has torn HSA down first).
    python tools/sanitizer_summary.py <reports.txt>"""
import re
import sys


def main(path):
    try:
        text = open(path, errors="replace").read()
    except FileNotFoundError:
        print("   reports: none (no report file)")
        return 0
    # a report starts at a line that names the sanitizer's verdict and runs to the next such line
    starts = [m.start() for m in re.finditer(r"^.*(ERROR: AddressSanitizer|WARNING: ThreadSanitizer|runtime error:|ERROR: LeakSanitizer)", text, flags=re.M)]
    reports = [text[a:b] for a, b in zip(starts, starts[1:] + [len(text)])]
    # (the canary of tools/run_sanitizers.sh - mbt_exact_split handed a 2-byte heap block on purpose - proves the instrumentation is live;
    # it is reported as such, not as a finding)
    canary = [r for r in reports if "in mbt_exact_split" in r and "2-byte region" in r]
    reports = [r for r in reports if r not in canary]
    ours = [r for r in reports if "libmbtenv" in r or "mbt_env.hip" in r]
    kinds = {}
    for r in reports:
        head = r.splitlines()[0]
        kind = re.sub(r"==\d+==|\(pc .*|0x[0-9a-f]+", "", head).strip()[:90]
        where = "libmbtenv" if r in ours else "outside libmbtenv"
        kinds[(where, kind)] = kinds.get((where, kind), 0) + 1
    print(f"   reports: {len(reports)} in all, {len(ours)} with a frame in libmbtenv" + (f"  (+ the canary: {len(canary)} heap-buffer-overflow in mbt_exact_split, caught)" if canary else ""))
    for (where, kind), count in sorted(kinds.items()):
        print(f"     {count:4d} x [{where}] {kind}")
    for r in ours[:5]:
        print("   ---- first lines of a libmbtenv report ----")
        print("\n".join("   " + line[:200] for line in r.splitlines()[:25]))
    return 1 if ours else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
