#!/usr/bin/env python3
"""tools/dbg/pmc_config.sh outputs (gpurun_out/dbg/pmc_config_<tag>.txt) -> one JSON with the shares that say what bounds a kernel:
vector-issue, s_waitcnt, instruction-issue wait, LDS, and the memory traffic the L2 saw (TCC_EA0 requests).
    python tools/pmc_config_summary.py tag=lanes=us_per_step ... > profiles/rNN_<name>_pmc.json"""
import json
import re
import sys


def main(specs):
    out = {"source": "rocprofv3 --pmc, one small counter group per pass (tools/dbg/pmc_config.sh), 50 launches of the step kernel each; per-launch means",
           "how_to_read": "shares are of SQ_WAVE_CYCLES (the sum over waves of the cycles a wave is resident): valu_issue = SQ_ACTIVE_INST_VALU, waitcnt = SQ_WAIT_ANY "
                          "(parked at s_waitcnt: memory), issue_wait = SQ_WAIT_INST_ANY (an instruction ready, waiting for its unit), lds = SQ_ACTIVE_INST_LDS; "
                          "simd_valu_busy = valu_issue x waves per SIMD (all of the launch's waves are resident at once when waves <= 1024 SIMDs x the occupancy); "
                          "l2_to_fabric bytes = TCC_EA0_RDREQ x 128 B + TCC_EA0_WRREQ x 64 B"}
    for spec in specs:
        tag, lanes, us = spec.split("=")
        lanes, us = int(lanes), float(us)
        text = open(f"gpurun_out/dbg/pmc_config_{tag}.txt").read()
        c = {m.group(1): float(m.group(2)) for m in re.finditer(r"^(\w+)\s+mean per launch\s+([0-9.]+)", text, flags=re.M)}
        wave_cycles, waves = c["SQ_WAVE_CYCLES"], c["SQ_WAVES"]
        per_simd = waves / 1024.0
        moved = c.get("TCC_EA0_RDREQ_sum", 0.0) * 128 + c.get("TCC_EA0_WRREQ_sum", 0.0) * 64
        out[tag] = {
            "case": text.splitlines()[0].lstrip("# "), "lanes": lanes, "us_per_step": us, "waves": waves, "waves_per_simd_in_the_whole_launch": per_simd,
            "valu_instructions_per_wave": c["SQ_INSTS_VALU"] / waves, "resident_cycles_per_wave": wave_cycles / waves,
            "share_valu_issue": c["SQ_ACTIVE_INST_VALU"] / wave_cycles, "share_waitcnt": c["SQ_WAIT_ANY"] / wave_cycles,
            "share_issue_wait": c["SQ_WAIT_INST_ANY"] / wave_cycles, "share_lds": c.get("SQ_ACTIVE_INST_LDS", 0.0) / wave_cycles,
            "share_wait_for_lds": c.get("SQ_WAIT_INST_LDS", 0.0) / wave_cycles, "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT", 0.0),
            "simd_valu_busy": c["SQ_ACTIVE_INST_VALU"] / wave_cycles * min(per_simd, 8.0),
            "l2_to_fabric_bytes_per_lane": moved / lanes, "l2_to_fabric_TBps": moved / us / 1e6, "counters": c,
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
