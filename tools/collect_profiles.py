#!/usr/bin/env python3
"""Copy the evidence of one tools/gpu_round.sh run from gpurun_out/ (scratch) into profiles/<tag>/ (tracked): rocprofv3 kernel
stats, the bench lines, and the counter CSVs reduced to the rows of this library's kernels.  usage: collect_profiles.py <tag>"""
import csv
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)


def copy(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))


def rows_of_ours(a, b):
    path = os.path.join(src, a)
    if not os.path.exists(path):
        return
    with open(path) as f, open(os.path.join(dst, b), "w", newline="") as g:
        r = csv.DictReader(f)
        w = csv.DictWriter(g, fieldnames=r.fieldnames)
        w.writeheader()
        for row in r:
            if "lnsfaid" in row.get("Kernel_Name", ""):
                w.writerow(row)


copy("prof_%s/stats_kernel_stats.csv" % tag, "kernel_stats.csv")
copy("prof_%s/bench_under_rocprof.json" % tag, "bench_under_rocprof.json")
copy("bench_%s.json" % tag, "bench_unprofiled.json")
copy("bench_cfg5_%s.json" % tag, "bench_config5_method5_16qam.json")
copy("gputests_%s.log" % tag, "gpu_tests.log")
rows_of_ours("pmc_%s/fetch/p_counter_collection.csv" % tag, "pmc_fetch.csv")
rows_of_ours("pmc_%s/write/p_counter_collection.csv" % tag, "pmc_write.csv")
rows_of_ours("pmc_%s/l2/p_counter_collection.csv" % tag, "pmc_l2.csv")
rows_of_ours("pmc_sq_%s/p1/p_counter_collection.csv" % tag, "pmc_sq_pass1.csv")
rows_of_ours("pmc_sq_%s/p2/p_counter_collection.csv" % tag, "pmc_sq_pass2.csv")
copy("pmc_sq_%s/kernel_source_hash.txt" % tag, "kernel_source_hash.txt")
print(sorted(os.listdir(dst)))
