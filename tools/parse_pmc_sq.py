#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc SQ CSVs of tools/gpu_pmc_sq.sh into profiles/valu_issue_per_launch.json.

bench.py replays `valu_instructions_per_launch` as the numerator of roofline.achieved (wave64 VALU instructions per
second against 1024 SIMDs x 2.4 GHz / 2 cycles), but only while the kernel source hash recorded here matches the
source it runs.  usage: parse_pmc_sq.py gpurun_out/pmc_sq_<tag> <tag>
"""
import collections
import csv
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]


def load(sub):
    agg = collections.defaultdict(list)
    path = os.path.join(src, sub, "p_counter_collection.csv")
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        if "lnsfaid_decode" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg["_kernel"] = r["Kernel_Name"]
            agg["_waves"] = int(r["Grid_Size"]) // 64
    return agg


p1, p2 = load("p1"), load("p2")
mean = lambda v: sum(v) / len(v)
out = {
    "tag": tag,
    "kernel": p1["_kernel"],
    "kernel_source_hash": open(os.path.join(src, "kernel_source_hash.txt")).read().strip(),
    "source": "profiles/%s/pmc_sq_pass1.csv (tools/gpu_pmc_sq.sh)" % tag,
    "valu_instructions_per_launch": mean(p1["SQ_INSTS_VALU"]),
    "lds_instructions_per_launch": mean(p1["SQ_INSTS_LDS"]),
    "salu_instructions_per_launch": mean(p1["SQ_INSTS_SALU"]),
    "waves_per_launch": mean(p1["SQ_WAVES"]) if p1.get("SQ_WAVES") else p1["_waves"],
    "simds": 1024,
    "clock_GHz": 2.4,
    "cycles_per_wave64_valu_instruction_at_full_rate": 2,
    "note": "MI355X SIMDs are 32 lanes wide: a wave64 VALU instruction of the full-rate class (two-source 32-bit operations, "
            "v_bitop3_b32) issues in 2 cycles; packed 16-bit, most three-source VOP3 and SGPR-operand forms take 4 "
            "(tools/ubench/valu_rate.hip).  peak = simds x clock / 2.",
}
for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
          "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
    for p in (p1, p2):
        if p.get(k):
            out[k] = mean(p[k])
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
json.dump(out, open(os.path.join(root, "valu_issue_per_launch.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
