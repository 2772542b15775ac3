#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc CSVs of tools/gpu_pmc.sh into profiles/hbm_traffic_per_launch.json.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
(16 B per lane) coalesced read stream and is uncalibrated for other widths
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section); the 4 B per lane pattern is calibrated here on
lnsfaid_count_errors_kernel, whose read volume is known exactly (n_frames x K bytes).
usage: parse_pmc.py gpurun_out/pmc_<tag> <tag>
"""
import collections
import csv
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
N, K = 17664, 14592


kernel_name = [None]


def load(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    grid = {}
    for r in csv.DictReader(open(os.path.join(src, sub, "p_counter_collection.csv"))):
        name = "decode" if "lnsfaid_decode" in r["Kernel_Name"] else ("count" if "lnsfaid_count" in r["Kernel_Name"] else None)
        if name == "decode":
            kernel_name[0] = r["Kernel_Name"]
        if name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            grid[name] = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
    return agg, grid


fetch, grid = load("fetch")
write, _ = load("write")
mean = lambda v: sum(v) / len(v)
n_cw = grid["decode"]  # one workgroup per codeword in both decode kernels
dec_fetch_kib = mean(fetch["decode"]["FETCH_SIZE"])
dec_write_kib = mean(write["decode"]["WRITE_SIZE"])
cnt_fetch_kib = mean(fetch["count"]["FETCH_SIZE"])
cal4 = (n_cw * K) / (cnt_fetch_kib * 1024.0)  # true bytes / reported bytes for 4 B-per-lane reads
# decode-kernel reads: 16 B per lane compressed-message rows (x2 rule) and 4 B per lane LLR / state words
corrected_read = dec_fetch_kib * 1024.0 * 2.0
out = {
    "tag": tag,
    "kernel_source_hash": open(os.path.join(src, "kernel_source_hash.txt")).read().strip(),
    "kernel": kernel_name[0],
    "kernel_instance": kernel_name[0],   # bench.py replays the file only for this template instance ...
    "library": open(os.path.join(src, "library_version.txt")).read().strip() if os.path.exists(os.path.join(src, "library_version.txt")) else None,  # ... of this build
    "codewords_per_launch": n_cw,
    "FETCH_SIZE_KiB_raw": dec_fetch_kib,
    "WRITE_SIZE_KiB_raw": dec_write_kib,
    "fetch_calibration_4B_per_lane": cal4,
    "read_bytes_corrected_x2": corrected_read,
    "write_bytes": dec_write_kib * 1024.0,
    "bytes_per_launch": corrected_read + dec_write_kib * 1024.0,
    "bytes_per_codeword": (corrected_read + dec_write_kib * 1024.0) / n_cw,
    "note": "reads doubled per the gfx950 FETCH_SIZE rule for wide coalesced streams (upper bound for the 4 B-per-lane part, "
            "whose own calibration factor on the counter kernel is fetch_calibration_4B_per_lane); writes taken as reported",
}
try:
    l2, _ = load("l2")
    h, m = mean(l2["decode"]["TCC_HIT_sum"]), mean(l2["decode"]["TCC_MISS_sum"])
    out["l2_hit_rate"] = h / (h + m)
except Exception:
    pass
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
json.dump(out, open(os.path.join(root, "hbm_traffic_per_launch.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
