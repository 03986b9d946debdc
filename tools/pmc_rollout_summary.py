#!/usr/bin/env python3
"""Shader-side counters of the returns-only fused rollout -> profiles/<tag>_pmc_rollout.json (what bench.py's `rollout` block quotes).

    gpurun -- 'python tools/pmc_rollout_summary.py gpurun_out/profiles_r05/r05_pmc_rollout.json'

Runs rocprofv3 --pmc once per small counter group (no tracing domains) over tools/dbg/rollout_once.py with each policy, three
launches of rollout_kernel at 2^20 lanes x 1000 steps each.  SURVEY section 8d asks for instruction throughput in this mode (the
kernel touches HBM once per episode):
    valu_issue_fraction = SQ_INSTS_VALU x 4 cycles (a wave64 VALU instruction occupies its SIMD's issue port for 4 cycles)
                          / (1024 SIMDs x launch cycles),   launch cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs)
Philox's 32-bit multiplies and the transcendentals issue at a lower rate than that, so the fraction UNDERSTATES how full the port is.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = ["SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY SQ_WAIT_ANY", "SQ_WAVE_CYCLES SQ_BUSY_CYCLES",
          "SQ_INSTS_VALU_TRANS", "GRBM_GUI_ACTIVE"]


def collect(policy):
    acc = {}
    for group in GROUPS:
        out = f"/tmp/pmc_ro_{policy}_{group.split()[0]}"
        subprocess.run(["rm", "-rf", out])
        env = dict(os.environ, TMPDIR="/tmp", MBT_ROLLOUT_POLICY=policy)
        rc = subprocess.run(["rocprofv3", "--pmc", *group.split(), "--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "tools/dbg/rollout_once.py")],
                            cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
        if rc.returncode != 0 or not files:
            acc["failed: " + group] = rc.stderr[-300:]
            continue
        values = collections.defaultdict(list)
        for row in csv.DictReader(open(files[0])):
            if "rollout_kernel" in row["Kernel_Name"]:
                values[row["Counter_Name"]].append(float(row["Counter_Value"]))
                acc["kernel"] = row["Kernel_Name"]
        for name, v in values.items():
            acc[name] = sum(v) / len(v)
            acc["launches_profiled"] = len(v)
    if "SQ_INSTS_VALU" in acc and "GRBM_GUI_ACTIVE" in acc:
        cycles = acc["GRBM_GUI_ACTIVE"] / 8.0
        acc["launch_cycles"] = cycles
        acc["valu_issue_fraction"] = acc["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cycles)
        acc["valu_instructions_per_wave_and_step"] = acc["SQ_INSTS_VALU"] / acc["SQ_WAVES"] / 1000.0
    acc["lanes"], acc["steps_per_launch"] = 1 << 20, 1000
    return acc


def main():
    out_path = sys.argv[1]
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    result = {"avellaneda_stoikov_policy": collect("as"), "fixed_policy": collect("fixed"),
              "formula": "valu_issue_fraction = SQ_INSTS_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); one counter group per rocprofv3 pass"}
    json.dump(result, open(out_path, "w"), indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
