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
# the recorded rollout: what the waves wait for, and whether the memory side pushes back on its writes
RECORD_GROUPS = GROUPS + ["SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM", "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum",
                          "TCC_EA0_WRREQ_64B_sum", "TA_BUSY_avr"]


def collect(policy, record=0):
    acc = {}
    for group in (RECORD_GROUPS if record else GROUPS):
        out = f"/tmp/pmc_ro_{policy}_{record}_{group.split()[0]}"
        subprocess.run(["rm", "-rf", out])
        env = dict(os.environ, TMPDIR="/tmp", MBT_ROLLOUT_POLICY=policy, MBT_ROLLOUT_RECORD=str(record))
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
    acc["lanes"], acc["steps_per_launch"] = 1 << (record or 20), 1000
    if record and "SQ_WAVE_CYCLES" in acc:  # where a wave's resident cycles go (the three are disjoint and add up to SQ_WAVE_CYCLES)
        total = acc["SQ_WAVE_CYCLES"]
        acc["wave_cycles_issuing"] = acc.get("SQ_ACTIVE_INST_ANY", 0.0) / total
        acc["wave_cycles_stalled_at_issue"] = acc.get("SQ_WAIT_INST_ANY", 0.0) / total
        acc["wave_cycles_parked_on_waitcnt"] = acc.get("SQ_WAIT_ANY", 0.0) / total
    if record and acc.get("TCC_EA0_WRREQ_sum"):
        acc["l2_write_requests_stalled_cycles_per_request"] = acc.get("TCC_EA0_WRREQ_STALL_sum", 0.0) / acc["TCC_EA0_WRREQ_sum"]
        acc["l2_write_request_bytes_per_launch_if_64B"] = acc.get("TCC_EA0_WRREQ_64B_sum", 0.0) * 64.0
    return acc


def main():
    out_path = sys.argv[1]
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    result = {"avellaneda_stoikov_policy": collect("as"), "fixed_policy": collect("fixed"),
              "recorded_avellaneda_stoikov_2^18": collect("as", 18), "recorded_avellaneda_stoikov_2^20": collect("as", 20),
              "formula": "valu_issue_fraction = SQ_INSTS_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); one counter group per rocprofv3 pass"}
    json.dump(result, open(out_path, "w"), indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
