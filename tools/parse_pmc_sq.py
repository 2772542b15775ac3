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


def half_rate_fraction(kernel_name):
    """Share of the half-rate issue class (tools/isa_histogram.py) among the VALU instructions of the layer step, from the kernel
    compiled here: the two per-degree instances of the layer loop, weighted 11 : 1 (11 layers of check degree 23, one of degree
    22).  The layer step is 93 % of the launch (profiles/r02_kernel4/launch_breakdown.txt); the counted total is split by it."""
    import re
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import isa_histogram
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    csrc = os.path.join(root, "mod-interleaveavx_multithreads-faid_amd", "csrc")
    m = re.search(r"lnsfaid_decode4_kernel<(\d), (true|false), (true|false)>", kernel_name)
    if not m:
        return None, None
    mangled = "lnsfaid_decode4_kernelILi%sELb%dELb%dE" % (m.group(1), m.group(2) == "true", m.group(3) == "true")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k4.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"), "-I" + csrc,
                        "-S", "--cuda-device-only", "-o", out, os.path.join(csrc, "lnsfaid_kernel4.hip")], check=True, capture_output=True)
        text = open(out).read()
    body = [b for n, b in zip(*[iter(re.split(r"^(_Z\w+):", text, flags=re.M)[1:])] * 2) if mangled in n][0].split(".end_amdhsa_kernel")[0]
    blocks = []
    for chunk in re.split(r"^\.LBB\d+_\d+:", body, flags=re.M):
        ins = [l.split(";")[0].strip() for l in chunk.split("\n")]
        valu = [isa_histogram.classify(i)[1] for i in ins if i.startswith("v_")]
        blocks.append((len(valu), sum(1 for c in valu if c != "full")))
    blocks.sort(reverse=True)
    (v23, h23), (v22, h22) = blocks[0], blocks[1]  # the degree-23 and the degree-22 instance of the layer step
    return (11.0 * h23 + h22) / (11.0 * v23 + v22), {"deg23": {"valu": v23, "half_rate": h23}, "deg22": {"valu": v22, "half_rate": h22}}


half_frac, half_detail = half_rate_fraction(p1["_kernel"])
out = {
    "tag": tag,
    "kernel": p1["_kernel"],
    "kernel_instance": p1["_kernel"],   # bench.py replays the file only for this template instance ...
    "library": open(os.path.join(src, "library_version.txt")).read().strip() if os.path.exists(os.path.join(src, "library_version.txt")) else None,  # ... of this build
    "valu_half_rate_fraction": half_frac,
    "valu_half_rate_fraction_source": "hipcc -S of the layer step, tools/isa_histogram.py classes, instances weighted 11 : 1",
    "layer_step_instances": half_detail,
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
