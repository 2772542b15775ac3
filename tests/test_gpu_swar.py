"""GPU checks of the byte-parallel layer step's building blocks (csrc/lnsfaid_swar.h) in isolation: the device instructions
against the header's host restatements of their ISA semantics, and the layer step on the device against the same statements run
on the CPU (oracle/swar_emul.cpp), which tests/test_swar_emul.py holds bit-exact against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_abi as oa

pytestmark = pytest.mark.gpu

DEV = os.path.join(oa.ORACLE_DIR, "libswar_devtest.so")
EMU = os.path.join(oa.ORACLE_DIR, "libswar_emul.so")


class SwParams(C.Structure):
    _fields_ = [("lut_lo", C.c_uint32), ("lut_hi", C.c_uint32), ("ef_lo", C.c_uint32), ("ef_hi", C.c_uint32),
                ("f1", C.c_int32), ("f2", C.c_int32), ("window", C.c_int32), ("ef_tables", C.c_int32),
                ("nms_t", C.c_uint32 * 4),
                ("oms_lo", C.c_uint32 * 2), ("oms_hi", C.c_uint32 * 2)]  # = struct SwParams (the tables: min-sum decoders only)


def _check_layout(dev):
    """this mirror of struct SwParams against the header the device library was built from (a field added there and forgotten
    here once made the layer test fail with wrong En everywhere: profiles/r02a/gpu_tests.log)"""
    dev.swar_devtest_sizeof_params.restype = C.c_int
    assert dev.swar_devtest_sizeof_params() == C.sizeof(SwParams), "tests/test_gpu_swar.py: SwParams differs from lnsfaid_swar.h"


def test_instruction_semantics_match_the_host_restatements():
    dev, emu = C.CDLL(DEV), C.CDLL(EMU)
    rng = np.random.default_rng(5)
    n = 1 << 16
    a = rng.integers(0, 2**32, n, dtype=np.uint32)
    b = rng.integers(0, 2**32, n, dtype=np.uint32)
    c = rng.integers(0, 2**32, n, dtype=np.uint32)
    c[: n // 2] &= 0x0f0f0f0f  # selectors 0..15 often: the table / sign-replicate / constant cases of v_perm_b32
    got = np.zeros((8, n), dtype=np.uint32)
    want = np.zeros((8, n), dtype=np.uint32)
    assert dev.swar_devtest_ops(n, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p)) == 0
    emu.swar_emul_ops(n, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p))
    names = ["v_perm_b32", "v_alignbyte_b32", "bitop3 select", "bitop3 (~a&b)|c", "mask7", "bitop3 a&(b|c)", "bitop3 c?b:a", "bitop3 (a^b)&c"]
    for k, name in enumerate(names):
        bad = np.nonzero(got[k] != want[k])[0]
        assert bad.size == 0, "%s differs, e.g. a=%08x b=%08x c=%08x dev=%08x host=%08x" % (
            name, a[bad[0]], b[bad[0]], c[bad[0]], got[k][bad[0]], want[k][bad[0]])


@pytest.mark.parametrize("n_iter", [1, 2, 10])
def test_layer_step_on_the_device_equals_the_cpu_run(abi, code50, n_iter):
    dev, emu = C.CDLL(DEV), C.CDLL(EMU)
    _check_layout(dev)
    cfg = abi.default_cfg(2, 10)
    N, M = code50.N, code50.M
    K = N - M
    fix = oa.ReferenceChannel(code50, 101, 13.0).groups(3.4, 1)
    want = np.empty(32 * N, dtype=np.int8)
    emu.swar_emul_layered.argtypes = [C.POINTER(abi.Code), C.POINTER(abi.Cfg), C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    assert emu.swar_emul_layered(C.byref(code50.code), C.byref(cfg), fix.ctypes.data, n_iter, 1, want.ctypes.data) == 0
    # interleaved, biased En images of the 32 codewords (what the decode kernel stages)
    v = np.arange(N)
    pos = (v & ~255) | ((v & 63) << 2) | ((v >> 6) & 3)
    img = np.zeros((32, N), dtype=np.uint8)
    for l in range(32):
        llr = np.concatenate([fix[l * K:(l + 1) * K], fix[32 * K + l * M:32 * K + (l + 1) * M]]).astype(np.int16)
        llr[N - code50.code.puncture_tail:] = 0
        img[l, pos] = (llr + 120).astype(np.uint8)
    nbr = M // 256
    pv = np.ctypeslib.as_array(code50.pos_vn)
    deg = np.array([23, 22] + [23] * 10, dtype=np.int32)
    sb = np.zeros((nbr, 24), dtype=np.uint32)
    e = 0
    for br in range(nbr):
        sb[br, :deg[br]] = pv[e:e + deg[br]]
        e += deg[br] * 256
    p6 = (SwParams * 6)()
    for it in range(6):
        lo = hi = 0
        for a in range(8):
            val = cfg.v2c_map[it][0][a] & 0xff
            if a < 4:
                lo |= val << (8 * a)
            else:
                hi |= val << (8 * (a - 4))
        p6[it] = SwParams(lo, hi, 0, 0, 0, 0, 0, 0, (C.c_uint32 * 4)(0, 0, 0, 0), (C.c_uint32 * 2)(0, 0), (C.c_uint32 * 2)(0, 0))
    assert dev.swar_devtest_layers(32, N, nbr, deg.ctypes.data_as(C.c_void_p), sb.ctypes.data_as(C.c_void_p), p6, n_iter,
                                   img.ctypes.data_as(C.c_void_p)) == 0
    got = (img[:, pos].astype(np.int16) - 120).astype(np.int8).reshape(-1)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "En differs at %s: device %s, cpu %s" % (bad[:8].tolist(), got[bad[:8]].tolist(), want[bad[:8]].tolist())
