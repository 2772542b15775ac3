#!/usr/bin/env python3
"""Extract the reference's one known-answer fixture: the 17664-bit codeword kept (commented out) in
reference Codeword.h:7-460, into tests/golden/codeword_50gpon.bin (MSB-first packbits, 2208 bytes).

Data only: the numbers of the initialiser are parsed and packed; no reference source text is stored.
Run in the build container (needs /root/reference):  python tools/extract_codeword.py
"""
import hashlib
import os
import re
import sys

import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/Codeword.h"
text = open(ref).read()
start = text.index("CodeWord_sym[_NoVar] = { 1")
end = text.index("};", start)
bits = np.array([int(x) for x in re.findall(r"\b[01]\b", text[start:end].split("{", 1)[1])], dtype=np.uint8)
assert bits.size == 17664, bits.size
packed = np.packbits(bits)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "codeword_50gpon.bin")
packed.tofile(out)
print("bits", bits.size, "weight", int(bits.sum()), "sha256", hashlib.sha256(packed.tobytes()).hexdigest())
