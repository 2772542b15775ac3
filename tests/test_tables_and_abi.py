"""CPU tests: the code definition, the shipped constants and the C-ABI surface (no GPU calls)."""
import ctypes as C
import hashlib
import os
import re

import numpy as np

import oracle_abi as oa

ROOT = oa.ROOT
GOLD = os.path.join(ROOT, "tests", "golden")

# SHA-256 of the reference's PosNoeudsVariable as 70400 little-endian uint16
POS_SHA256 = "4e4d6d5c493c87bcdf8c20238f1d62c03ee28c8563d1df01a3e07db2241fe15d"
# sha256 of the MSB-first packbits of the codeword in reference Codeword.h:7-460
CODEWORD_SHA256 = "8c8cb551186a68461b13f3cee7991a170aef7a2a35b7478e0f3c359b49596163"


def test_code_table_matches_reference_digest(code50):
    pos = np.ctypeslib.as_array(code50.pos_vn)
    assert (code50.N, code50.M, code50.code.n_edges, code50.code.z, code50.code.puncture_tail) == (17664, 3072, 70400, 256, 384)
    assert list(code50.deg) == [23, 22, 23] and list(code50.deg_rows) == [256, 256, 2560]
    assert hashlib.sha256(pos.astype("<u2").tobytes()).hexdigest() == POS_SHA256


def test_code_table_matches_reference_header_text(code50):
    """Only in the build container: compare with the header the reference ships (read as text)."""
    path = "/root/reference/Constants/50GPON-dc-original/Constants_SSE.h"
    if not os.path.exists(path):
        import pytest
        pytest.skip("reference not mounted")
    txt = open(path).read()
    i = txt.index("{", txt.index("_PosNoeudsVariable_"))
    body = re.sub(r"/\*.*?\*/", "", txt[i + 1:txt.index("};", i)])
    ref = np.array([int(x) for x in re.findall(r"\d+", body)], dtype=np.uint16)
    assert np.array_equal(ref, np.ctypeslib.as_array(code50.pos_vn))


def test_known_codeword_satisfies_every_check(code50):
    """The reference's one known-answer vector (Codeword.h:7-460) has zero syndrome against the table."""
    packed = np.fromfile(os.path.join(GOLD, "codeword_50gpon.bin"), dtype=np.uint8)
    assert hashlib.sha256(packed.tobytes()).hexdigest() == CODEWORD_SHA256
    bits = np.unpackbits(packed)[:code50.N]
    assert int(bits.sum()) == 8759
    pos = np.ctypeslib.as_array(code50.pos_vn).astype(np.int64)
    row_deg = np.repeat(np.array(list(code50.deg)), np.array(list(code50.deg_rows)))
    starts = np.concatenate([[0], np.cumsum(row_deg)[:-1]])
    synd = np.add.reduceat(bits[pos].astype(np.int64), starts) & 1
    assert synd.shape == (3072,) and not synd.any()


def test_default_configurations_match_the_reference_constants(abi, lib):
    faid3 = [[0, 1, 1, 2, 3, 3, 3, 3], [0, 1, 1, 2, 3, 3, 3, 3], [0, 1, 1, 2, 4, 4, 4, 4],
             [0, 1, 1, 3, 3, 4, 4, 4], [0, 1, 1, 3, 3, 3, 6, 6], [0, 1, 1, 3, 3, 3, 7, 7]]  # CDecoder_FAID.cpp:13-48
    b2c1 = [[0, 0, 1, 2, 3, 3, 3, 3], [0, 1, 1, 2, 3, 3, 3, 3], [0, 1, 1, 2, 3, 3, 3, 3],
            [0, 1, 1, 3, 3, 4, 4, 4], [0, 1, 1, 3, 3, 3, 6, 6], [0, 1, 1, 3, 3, 3, 7, 7]]  # CDecoder_FAID_2B1C.cpp:12-47
    ef = [2, 3, 3, 4, 5, 6, 6, 7]
    for method, table, fec, fit, efe, bf, l0 in [(2, faid3, 0, -1, 0, 10, 50), (5, b2c1, 50, 6, 1, 10, 100)]:
        c = abi.default_cfg(method, 10, lib)
        assert (c.floor_err_count, c.floor_iter_thresh, c.ef_elimination, c.max_bf_iter, c.bf_L0) == (fec, fit, efe, bf, l0)
        assert (c.bf_L1, c.bf_alpha, c.bf_delta, c.regular_col_weight, c.hard2_threshold, c.factor_1, c.factor_2) == (0, 1, 1, 3, 13, 1, 6)
        for it in range(6):
            for w in range(4):
                assert list(c.v2c_map[it][w]) == table[it]
                assert list(c.v2c_map_ef[it][w]) == ef
    c = abi.default_cfg(1, 10, lib)
    assert (c.floor_err_count, c.floor_iter_thresh, c.max_bf_iter) == (100, 4, 0)
    c = abi.default_cfg(3, 10, lib)  # CDecoder_OMSBF.cpp:28-30, :3332
    assert (c.floor_err_count, c.floor_iter_thresh, c.max_bf_iter, c.bf_vote_cap) == (100, 4, 50, 5)
    c = abi.default_cfg(4, 10, lib)  # CDecoder_OMS_DTBF.cpp:6-9, :33-35
    assert (c.floor_err_count, c.floor_iter_thresh, c.max_bf_iter, c.bf_L0, c.bf_L1, c.bf_alpha, c.bf_delta) == (100, 4, 50, 0, 50, 1, 1)
    c = abi.default_cfg(0, 7, lib)  # CLDPC::Decode: no early stop, no bit flipping
    assert (c.decode_method, c.max_iteration, c.max_bf_iter, c.factor_1, c.factor_2) == (0, 7, 0, 1, 6)
    bad = abi.Cfg()
    assert lib.lnsfaid_cfg_default(C.byref(bad), 6, 10) != 0 and lib.lnsfaid_cfg_default(C.byref(bad), -1, 10) != 0


def test_library_exports_every_symbol_of_the_header(abi, lib):
    header = open(os.path.join(ROOT, "include", "lnsfaid.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(lnsfaid_[a-z0-9_]+)\s*\(", header))
    assert declared == set(abi.SYMBOLS), declared ^ set(abi.SYMBOLS)
    raw = C.CDLL(abi.LIB_PATH)
    for name in declared:
        getattr(raw, name)  # AttributeError if not exported
    assert lib.lnsfaid_version().decode().startswith("lnsfaid-amd")
    assert lib.lnsfaid_strerror(-2).decode().startswith("code table")


def test_create_rejects_bad_arguments_without_touching_a_gpu(abi, lib, code50):
    ctx = C.c_void_p()
    cfg = abi.default_cfg(2, 10, lib)
    assert lib.lnsfaid_create(C.byref(ctx), C.byref(code50.code), C.byref(cfg), 0, 0) == -1  # max_groups 0
    # a table that is not quasi-cyclic is refused before any device work (LNSFAID_E_CODE)
    broken = abi.Code50GPON(lib)
    broken.pos_vn[5], broken.pos_vn[6] = broken.pos_vn[6], broken.pos_vn[5]
    assert lib.lnsfaid_create(C.byref(ctx), C.byref(broken.code), C.byref(cfg), 0, 1) == -2
    cfg.v2c_map[0][0][0] = 9  # outside the 3-bit alphabet
    assert lib.lnsfaid_create(C.byref(ctx), C.byref(code50.code), C.byref(cfg), 0, 1) == -1
    assert not ctx.value


def test_table_presets_match_the_reference_text(abi, lib):
    """FAID3 / FAID32 / FAID2 of CDecoder_FAID.cpp:12-127: the 'weight = 3' rows as printed there (all four rows of a
    table are equal in the reference)."""
    want = {
        0: [[0, 1, 1, 2, 3, 3, 3, 3], [0, 1, 1, 2, 3, 3, 3, 3], [0, 1, 1, 2, 4, 4, 4, 4],
            [0, 1, 1, 3, 3, 4, 4, 4], [0, 1, 1, 3, 3, 3, 6, 6], [0, 1, 1, 3, 3, 3, 7, 7]],
        1: [[0, 1, 1, 2, 3, 3, 3, 3], [0, 1, 1, 2, 3, 3, 3, 3], [0, 1, 1, 2, 4, 4, 4, 4],
            [1, 1, 1, 1, 4, 4, 4, 4], [1, 1, 1, 1, 5, 5, 5, 5], [1, 1, 1, 1, 6, 6, 6, 6]],
        2: [[0, 0, 2, 2, 2, 2, 2, 2], [0, 0, 2, 2, 2, 2, 2, 2], [1, 1, 1, 3, 3, 3, 3, 3],
            [1, 1, 1, 4, 4, 4, 4, 4], [1, 1, 1, 5, 5, 5, 5, 5], [1, 1, 1, 6, 6, 6, 6, 6]],
    }
    for preset, rows in want.items():
        c = abi.default_cfg(2, 10, lib)
        assert lib.lnsfaid_cfg_table_preset(C.byref(c), preset) == 0
        for it in range(6):
            for w in range(4):
                assert list(c.v2c_map[it][w]) == rows[it]
    assert lib.lnsfaid_cfg_table_preset(C.byref(c), 3) != 0
