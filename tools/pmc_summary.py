#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide prescribes)
into profiles/<tag>_pmc_summary.json.  Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM):
counter values are KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced
streaming read, so it is doubled; WRITE_SIZE is used as reported.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [kernel substring]
"""
import collections
import csv
import json
import sys


def mean_by_kernel(path, counter):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    needle = sys.argv[4] if len(sys.argv) > 4 else "step_kernel"
    fetch = mean_by_kernel(fetch_csv, "FETCH_SIZE")
    write = mean_by_kernel(write_csv, "WRITE_SIZE")
    result = {}
    for name, (f_kib, calls) in fetch.items():
        if needle not in name:
            continue
        w_kib = write.get(name, (0.0, 0))[0]
        result[name] = {
            "launches_profiled": calls,
            "FETCH_SIZE_KiB_raw": f_kib,
            "FETCH_SIZE_bytes_corrected_x2": f_kib * 1024 * 2,
            "WRITE_SIZE_KiB_raw": w_kib,
            "WRITE_SIZE_bytes": w_kib * 1024,
            "hbm_bytes_per_launch": f_kib * 1024 * 2 + w_kib * 1024,
        }
    json.dump(result, open(out, "w"), indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
