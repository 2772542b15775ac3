#!/usr/bin/env python3
"""Write a code-definition header in the reference's Constants_SSE.h format
(reference Constants/50GPON-dc-original/Constants_SSE.h: macros :4-25, PosNoeudsVariable :29-3102)
from the built-in 50G-PON base matrix (csrc/lnsfaid_tables.c).  The output is a build artefact
(host/Constants/, git-ignored): the table is regenerated from the 12 x 69 base matrix, not stored.

usage: gen_constants.py <out_dir>
"""
import importlib.util
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
# the package directory name is not a Python identifier: its ctypes module is imported by path (nothing from tests/ or oracle/)
_spec = importlib.util.spec_from_file_location("lnsfaid_pyabi", os.path.join(ROOT, "mod-interleaveavx_multithreads-faid_amd", "pyabi.py"))
pyabi = importlib.util.module_from_spec(_spec)
sys.modules["lnsfaid_pyabi"] = pyabi
_spec.loader.exec_module(pyabi)
code = pyabi.Code50GPON(pyabi.load())
out_dir = sys.argv[1]
sub = os.path.join(out_dir, "50GPON-dc-original")
os.makedirs(sub, exist_ok=True)
with open(os.path.join(out_dir, "Constants_SSE.h"), "w") as f:
    f.write('#define MATRIX_FILE "./50GPON-dc-original/Constants_SSE.h"\n#include MATRIX_FILE\n')
deg, rows = list(code.deg), list(code.deg_rows)
with open(os.path.join(sub, "Constants_SSE.h"), "w") as f:
    f.write("#ifndef CONSTANTES\n#define CONSTANTES\n")
    f.write("#define NB_DEGRES\t%d\n#define _NoVar\t%d\n#define _NoCheck\t%d\n#define _NoOnes\t%d\n" % (
        len(deg), code.N, code.M, code.code.n_edges))
    f.write("#define NOEUD   _NoVar\n#define MESSAGE _NoOnes\n#define  _PunctureBits\t0\n#define NmoinsK     (_NoVar-_NoCheck)\n")
    f.write("#define  _ShortenBits\t0\n#define  BitsOverChannel\t%d\n" % code.N)
    for k, (d, r) in enumerate(zip(deg, rows), 1):
        f.write("#define DEG_%d\t%d\n#define DEG_%d_COMPUTATIONS\t%d\n" % (k, d, k, r))
    f.write("#define NB_BITS_VARIABLES    6\n#define NB_BITS_MESSAGES     4\n")
    f.write("#define SAT_POS_VAR  ( (0x0001<<(NB_BITS_VARIABLES-1))-1)\n#define SAT_NEG_VAR  (-(0x0001<<(NB_BITS_VARIABLES-1))+1)\n")
    f.write("#define SAT_POS_MSG  ( (0x0001<<(NB_BITS_MESSAGES -1))-1)\n#define SAT_NEG_MSG  (-(0x0001<<(NB_BITS_MESSAGES -1))+1)\n#endif\n")
    f.write("#ifndef _PosNoeudsVariable_\n#define _PosNoeudsVariable_\nconst static unsigned short PosNoeudsVariable[_NoOnes]={\n")
    e = 0
    r = 0
    for d, n in zip(deg, rows):
        for _ in range(n):
            f.write("/*msg=%6d,deg=%3d*/\t" % (r, d) + ",\t".join(str(code.pos_vn[e + j]) for j in range(d)) + ",\n")
            e += d
            r += 1
    f.write("};\n#endif\n")
print("wrote", os.path.join(sub, "Constants_SSE.h"))
