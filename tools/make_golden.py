#!/usr/bin/env python3
"""Generate tests/golden/*.npz and tests/golden/anchors.json with the CPU oracle (oracle/).

The reference itself cannot be built here (DESIGN.md "Oracle"), so these vectors are outputs of the
restatement, whose link to the reference is the anchor table: the error counters the reference produced
in this container during the survey (SURVEY.md §6 / BASELINE.md §2), reproduced exactly by the oracle when
it is driven by the restated reference channel with the same seed.

Input LLRs are stored 4-bit packed ((v + 8) in a nibble, two per byte), hard decisions bit-packed.
Run:  python tools/make_golden.py   (a few minutes on 8 cores)
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_abi as oa  # noqa: E402

pyabi = oa.pyabi
lib = pyabi.load()
code = pyabi.Code50GPON(lib)
N = code.N
GOLD = os.path.join(ROOT, "tests", "golden")

codeword = np.unpackbits(np.fromfile(os.path.join(GOLD, "codeword_50gpon.bin"), dtype=np.uint8))[:N].astype(np.int8)


def pack_llr(fix):
    u = (fix.astype(np.int16) + 8).astype(np.uint8)
    return (u[0::2] | (u[1::2] << 4)).astype(np.uint8)


def group_case(name, method, max_iter, eb_n0, seed, skip, cw):
    """group number `skip` (0-based) of the stream the reference's thread with `seed` would see at eb_n0"""
    cfg = pyabi.default_cfg(method, max_iter, lib)
    ch = oa.ReferenceChannel(code, seed, 13.0)
    fix = ch.groups(eb_n0, skip + 1, cw)[skip * 32 * N:]
    dec, st = oa.Oracle(code, cfg).decode(fix, 1)
    ref_bits = np.tile(cw, 32) if cw is not None else None
    inp = None if cw is None else np.ascontiguousarray(np.tile(cw[:code.K], 32))
    cnt = oa.Oracle(code, cfg).count_errors(dec, inp, 1)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), fix_packed=pack_llr(fix), decoded_packed=np.packbits(dec.astype(np.uint8)),
                        stats=st, counters=np.array(cnt, dtype=np.uint64), method=method, max_iter=max_iter, eb_n0=eb_n0,
                        seed=seed, group_index=skip, known_codeword=int(cw is not None))
    print(name, "I/J", st.tolist(), "counters", cnt)


if __name__ == "__main__":
    # (a) full vectors for a handful of groups
    group_case("m2_3p5dB_g0", 2, 10, 3.5, 101, 0, None)   # mixed: converged + BF-repaired + failed lanes
    group_case("m2_4p2dB_g0", 2, 10, 4.2, 101, 0, None)   # all converge at different iterations (group coupling)
    group_case("m1_3p5dB_g1", 1, 10, 3.5, 101, 1, None)
    group_case("m5_3p5dB_g2", 5, 10, 3.5, 101, 2, None)
    group_case("m2_3p55dB_cw_g0", 2, 10, 3.55, 103, 0, codeword)  # non-zero codeword: sign asymmetries
    # (b) anchors: survey-recorded reference counters + oracle digests over 30 calls of seed 101
    survey = [  # DecodeMethod, Eb/N0, MaxIteration, frame errors, bit errors  (SURVEY.md §6 probe table)
        (2, 3.4, 10, 472, 198464), (2, 3.5, 10, 119, 39446), (2, 3.6, 10, 9, 2263), (2, 3.7, 10, 0, 0),
        (2, 4.2, 10, 0, 0), (1, 3.6, 10, 3, 83), (5, 3.6, 10, 4, 961), (2, 3.6, 6, 164, 12873),
    ]
    anchors = []
    for method, eb, mi, fe, be in survey:
        cfg = pyabi.default_cfg(method, mi, lib)
        fix = oa.ReferenceChannel(code, 101, 13.0).groups(eb, 30)
        dec, st = oa.decode_mt(code, cfg, fix, 30)
        cnt = oa.Oracle(code, cfg).count_errors(dec, None, 30)
        anchors.append(dict(method=method, eb_n0=eb, max_iter=mi, seed=101, groups=30, survey_frame_errors=fe,
                            survey_bit_errors=be, oracle_counters=cnt, oracle_sha256=hashlib.sha256(dec.tobytes()).hexdigest(),
                            oracle_sum_I=int(st[:, 0].sum()), oracle_sum_J=int(st[:, 1].sum())))
        print(anchors[-1])
        assert cnt[1] == fe and cnt[2] == be, "oracle does not reproduce the survey's reference counters"
    json.dump(anchors, open(os.path.join(GOLD, "anchors.json"), "w"), indent=1)
