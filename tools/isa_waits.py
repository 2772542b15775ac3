#!/usr/bin/env python3
"""List the vector-memory instructions and the s_waitcnt vmcnt(...) of the depth-2 loops (the layer loops) of a kernel in a gfx950
assembly file: a wait in front of a layer for anything younger than the previous layer's prefetch is an exposed memory round trip.
usage: isa_waits.py <file.s> <kernel-name-substring>"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
inside = False
blk, info, n = None, "", 0
for l in lines:
    if re.match(r"^_Z\S+:", l):
        inside = want in l
        continue
    if not inside:
        continue
    m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", l)
    if m:
        blk, info, n = m.group(1), m.group(2), 0
        continue
    if re.match(r"^\s+[a-z]", l) and not l.strip().startswith(";"):
        n += 1
        if "Depth=2" in info and ("vmcnt" in l or "global_" in l):
            print(blk, n, l.strip()[:70])
