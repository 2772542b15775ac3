#!/usr/bin/env python3
"""Per-class VALU instruction histogram of the basic blocks of a gfx950 kernel (the judge's "full-rate vs half-rate" split).

usage: isa_histogram.py <file.s> [kernel-name-substring] [--blocks N]
Classes (tools/ubench/valu_rate.hip, tools/ubench/issue_mix.hip, profiles/*/ubench_*.txt): a wave64 VALU instruction issues in 2 cycles
("full") unless it is a LEFT shift, a packed 16-bit operation, a three-source VOP3 other than v_bitop3_b32, an integer
multiply, a dot product, or reads an SGPR operand ("half", 4 cycles).  Prints the largest blocks of the kernel with their
instruction counts per class and the weighted issue cycles.
"""
import collections
import json
import re
import sys

# LEFT shifts belong to the half-rate class (tools/ubench/issue_mix.hip, profiles/r03/ubench_issue_mix_vs_occupancy.txt: v_lshlrev_b32 by 1, by 8
# or by a register 4.05 SIMD cycles per wave64 instruction at 8 waves per SIMD, like v_perm_b32; v_lshrrev_b32, v_add_u32, v_and_b32, v_bitop3_b32 2.1 - 2.2)
HALF_MNEMONICS = ("v_lshlrev_b32", "v_pk_", "v_perm_b32", "v_and_or_b32", "v_lshl_or_b32", "v_or3_b32", "v_bfi_b32", "v_bfe_", "v_med3", "v_min3", "v_max3",
                  "v_alignbit_b32", "v_alignbyte_b32", "v_mad_", "v_mul_lo", "v_mul_hi", "v_dot", "v_lshlrev_b64", "v_lshrrev_b64",
                  "v_ashrrev_i64", "v_add3_u32", "v_lshl_add_u32", "v_add_lshl_u32", "v_xad_u32", "v_sad_", "v_fma", "v_mul_u32_u24", "v_mul_i32_i24",
                  "v_cndmask_b32", "v_readlane", "v_writelane", "v_readfirstlane")


def classify(line):
    parts = line.split(None, 1)
    op = parts[0]
    args = parts[1] if len(parts) > 1 else ""
    if op.startswith("v_bitop3"):
        cls = "full"
    elif any(op.startswith(h) for h in HALF_MNEMONICS):
        cls = "half"
    else:
        cls = "full"
    # an SGPR (or vcc / exec / m0) source operand halves the rate of an otherwise full-rate instruction
    srcs = args.split(",")[1:]
    if cls == "full" and any(re.match(r"\s*(s\d+|s\[\d+:\d+\]|vcc|exec|m0)\b", s) for s in srcs) and not op.startswith("v_cmp"):
        cls = "half(sgpr)"
    return op, cls


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    nblocks = int(sys.argv[sys.argv.index("--blocks") + 1]) if "--blocks" in sys.argv else 4
    kernel, block = None, None
    blocks = collections.OrderedDict()
    for raw in open(path):
        line = raw.split(";")[0].rstrip()
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)
        if m:
            name = m.group(1)
            if not name.startswith(".L"):
                kernel = name
            block = (kernel, name)
            blocks.setdefault(block, [])
            continue
        s = line.strip()
        if not s or s.startswith("."):
            continue
        if block is not None:
            blocks[block].append(s)
    rows = []
    for (kern, name), ins in blocks.items():
        if want and (kern is None or want not in kern):
            continue
        valu = [classify(i) for i in ins if i.startswith("v_")]
        if not valu:
            continue
        c = collections.Counter(cls for _, cls in valu)
        ops = collections.Counter(op for op, _ in valu)
        rows.append({"kernel": kern, "block": name, "valu": len(valu), "full": c["full"], "half": c["half"], "half_sgpr": c["half(sgpr)"],
                     "lds": sum(1 for i in ins if i.startswith("ds_")), "salu": sum(1 for i in ins if i.startswith("s_") and not i.startswith("s_nop") and not i.startswith("s_waitcnt")),
                     "s_nop": sum(1 for i in ins if i.startswith("s_nop")), "vmem": sum(1 for i in ins if i.startswith(("global_", "buffer_", "flat_"))),
                     "issue_cycles_weighted": 2 * c["full"] + 4 * (c["half"] + c["half(sgpr)"]),
                     "top": ops.most_common(12)})
    rows.sort(key=lambda r: -r["valu"])
    for r in rows[:nblocks]:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
