"""CPU check of the four-rows-per-lane layer step: csrc/lnsfaid_swar.h compiled for the host (oracle/swar_emul.cpp; the ISA
semantics of v_perm_b32 / v_alignbyte_b32 / v_bitop3_b32 are restated in that header and checked against the device by
tests/test_gpu_swar.py) must leave exactly the oracle's a-posteriori LLRs after every layered iteration."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_abi as oa

EMU = os.path.join(oa.ORACLE_DIR, "libswar_emul.so")


def _both(abi, code50, method, eb_n0, n_iter, spec, seed=101, max_iter=10):
    lib = oa.load()
    lib.lnsfaid_oracle_layered_en.restype = C.c_int
    lib.lnsfaid_oracle_layered_en.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    em = C.CDLL(EMU)
    em.swar_emul_layered.restype = C.c_int
    em.swar_emul_layered.argtypes = [C.POINTER(abi.Code), C.POINTER(abi.Cfg), C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    cfg = abi.default_cfg(method, max_iter)
    fix = oa.ReferenceChannel(code50, seed, 13.0).groups(eb_n0, 1)
    o = oa.Oracle(code50, cfg)
    ref = np.empty(32 * code50.N, dtype=np.int8)
    assert lib.lnsfaid_oracle_layered_en(o.h, fix.ctypes.data, n_iter, ref.ctypes.data) == 0
    got = np.empty(32 * code50.N, dtype=np.int8)
    assert em.swar_emul_layered(C.byref(code50.code), C.byref(cfg), fix.ctypes.data, n_iter, spec, got.ctypes.data) == 0
    return ref, got


@pytest.mark.parametrize("method,eb_n0,n_iter,spec", [(2, 3.4, 1, 1), (2, 3.4, 3, 1), (2, 3.0, 10, 0), (2, 4.2, 10, 1),
                                                       (1, 3.4, 10, 1), (4, 3.6, 7, 0), (5, 3.4, 10, 1), (5, 3.6, 10, 0)])
def test_swar_layer_step_equals_the_oracle(abi, code50, method, eb_n0, n_iter, spec):
    """En of all 32 lanes after n_iter iterations (no early stop): saturations at +-31, zero messages, ties between the minima,
    the back-tracked sign and the error-floor tables (DecodeMethod 5) / selective offsets (OMS loop) all pass through here."""
    ref, got = _both(abi, code50, method, eb_n0, n_iter, spec)
    bad = np.nonzero(ref != got)[0]
    assert bad.size == 0, "En differs at %s: oracle %s, layer step %s" % (bad[:8].tolist(), ref[bad[:8]].tolist(), got[bad[:8]].tolist())
    if n_iter == 10 and eb_n0 >= 3.4:
        assert np.abs(ref).max() == 31  # these batches reach the saturation limit: the merged clamp is exercised


@pytest.mark.parametrize("factor,eb_n0,n_iter,spec", [(24, 3.4, 10, 1), (24, 4.2, 6, 0), (32, 3.6, 10, 1), (15, 3.4, 8, 1), (19, 3.0, 10, 0),
                                                       (45, 3.6, 10, 1), (200, 3.6, 4, 1), (2000, 3.6, 4, 0)])
def test_swar_layer_step_runs_normalised_min_sum_with_one_factor(abi, code50, factor, eb_n0, n_iter, spec):
    """DecodeMethod 0 (CLDPC::Decode, CLDPC.cpp:287-375) on the byte-parallel layer step: the search keeps 16 levels of |t| and maps
    them through cste(m) = min((m * Factor) >> 5, 7); Factor_1 == Factor_2 >= 15 (lnsfaid_swar.h sw_nms_fits)."""
    lib = oa.load()
    lib.lnsfaid_oracle_layered_en.restype = C.c_int
    lib.lnsfaid_oracle_layered_en.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    em = C.CDLL(EMU)
    em.swar_emul_layered.restype = C.c_int
    em.swar_emul_layered.argtypes = [C.POINTER(abi.Code), C.POINTER(abi.Cfg), C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    cfg = abi.default_cfg(0, n_iter)
    cfg.factor_1 = cfg.factor_2 = factor
    fix = oa.ReferenceChannel(code50, 163, 13.0).groups(eb_n0, 1)
    ref = np.empty(32 * code50.N, dtype=np.int8)
    got = np.empty(32 * code50.N, dtype=np.int8)
    assert lib.lnsfaid_oracle_layered_en(oa.Oracle(code50, cfg).h, fix.ctypes.data, n_iter, ref.ctypes.data) == 0
    assert em.swar_emul_layered(C.byref(code50.code), C.byref(cfg), fix.ctypes.data, n_iter, spec, got.ctypes.data) == 0
    bad = np.nonzero(ref != got)[0]
    assert bad.size == 0, "En differs at %s: oracle %s, layer step %s" % (bad[:8].tolist(), ref[bad[:8]].tolist(), got[bad[:8]].tolist())


def test_normalised_min_sum_factors_the_layer_step_takes(abi, code50):
    """Two factors, or a factor whose cste() needs more than 16 levels of |t| (< 15) or wraps in 16 bits, stay on the other kernel:
    the CPU run of the layer step refuses them like lnsfaid_create routes them."""
    em = C.CDLL(EMU)
    em.swar_emul_layered.restype = C.c_int
    em.swar_emul_layered.argtypes = [C.POINTER(abi.Code), C.POINTER(abi.Cfg), C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    fix = oa.ReferenceChannel(code50, 163, 13.0).groups(3.6, 1)
    got = np.empty(32 * code50.N, dtype=np.int8)
    for f1, f2 in [(24, 28), (14, 14), (1, 1), (2185, 2185)]:  # 30 * 2185 wraps to 14 in 16 bits: cste(30) = 0
        cfg = abi.default_cfg(0, 2)
        cfg.factor_1, cfg.factor_2 = f1, f2
        assert em.swar_emul_layered(C.byref(code50.code), C.byref(cfg), fix.ctypes.data, 2, 1, got.ctypes.data) == -1, (f1, f2)


def _few_confident_errors(code50, rng, nflip, amp, noise):
    """All-zero codeword received almost cleanly, plus a few weight-3 variable nodes with the wrong sign at high confidence:
    few unsatisfied checks, and nodes whose three checks all fail - what EF_ELIMINATION 2 reacts to."""
    N, K = code50.N, code50.K
    llr = np.full((32, N), -amp, dtype=np.int16) + rng.integers(-noise, noise + 1, size=(32, N))
    for l in range(32):
        llr[l, rng.integers(17 * 256, 67 * 256, size=nflip)] = rng.integers(3, 8, size=nflip)
    llr = np.clip(llr, -7, 7).astype(np.int8)
    return np.concatenate([llr[:, :K].reshape(-1), llr[:, K:].reshape(-1)])


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("nflip,amp,noise,max_iter,n_iter", [(3, 4, 2, 6, 3), (6, 3, 2, 6, 4), (12, 3, 3, 7, 7), (20, 2, 2, 6, 6)])
def test_swar_layer_step_with_ef_elimination(abi, lib, code50, mode, nflip, amp, noise, max_iter, n_iter):
    """EF_ELIMINATION 1 (error-floor tables) and 2 (tables + erasure, CDecoder_FAID.cpp:673-680) of Decode_FAID: the layer
    step's erasing variant against the oracle's statement-by-statement restatement."""
    olib = oa.load()
    olib.lnsfaid_oracle_layered_en.restype = C.c_int
    olib.lnsfaid_oracle_layered_en.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    em = C.CDLL(EMU)
    em.swar_emul_layered.restype = C.c_int
    em.swar_emul_layered.argtypes = [C.POINTER(abi.Code), C.POINTER(abi.Cfg), C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    fix = _few_confident_errors(code50, np.random.default_rng(10 * nflip + mode), nflip, amp, noise)
    cfg = abi.default_cfg(2, max_iter)
    assert lib.lnsfaid_cfg_ef_elimination(cfg, mode) == 0
    o = oa.Oracle(code50, cfg)
    ref = np.empty(32 * code50.N, dtype=np.int8)
    got = np.empty(32 * code50.N, dtype=np.int8)
    assert olib.lnsfaid_oracle_layered_en(o.h, fix.ctypes.data, n_iter, ref.ctypes.data) == 0
    assert em.swar_emul_layered(C.byref(code50.code), C.byref(cfg), fix.ctypes.data, n_iter, 0, got.ctypes.data) == 0
    assert np.array_equal(ref, got)
    if mode == 2 and nflip <= 6:  # (more wrong nodes than that mean 20 or more unsatisfied checks: outside the rule) the erasure really happened: without it (tables only, same window) En comes out differently
        cfg.ef_elimination = 1
        o1 = oa.Oracle(code50, cfg)
        other = np.empty_like(ref)
        assert olib.lnsfaid_oracle_layered_en(o1.h, fix.ctypes.data, n_iter, other.ctypes.data) == 0
        assert not np.array_equal(other, ref)
