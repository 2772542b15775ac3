"""CPU tests of the oracle itself: pinned to the reference, to the committed golden vectors, and checked on
size-independent properties.  (The oracle is test infrastructure; these tests are what make it trustworthy.)"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_abi as oa

GOLD = os.path.join(oa.ROOT, "tests", "golden")
ANCHORS = json.load(open(os.path.join(GOLD, "anchors.json")))
GOLDEN = ["m2_3p5dB_g0", "m2_4p2dB_g0", "m1_3p5dB_g1", "m5_3p5dB_g2", "m2_3p55dB_cw_g0"]


def unpack_llr(packed):
    out = np.empty(packed.size * 2, dtype=np.int8)
    out[0::2] = (packed & 15).astype(np.int8) - 8
    out[1::2] = (packed >> 4).astype(np.int8) - 8
    return out


def load_golden(name, n_var):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    fix = unpack_llr(z["fix_packed"])
    dec = np.unpackbits(z["decoded_packed"])[:32 * n_var].astype(np.int8)
    return z, fix, dec


@pytest.mark.parametrize("a", ANCHORS[2:3] + ANCHORS[5:7] + ANCHORS[7:8] + ANCHORS[0:1],
                         ids=lambda a: "m%d_%.1fdB_it%d" % (a["method"], a["eb_n0"], a["max_iter"]))
def test_oracle_reproduces_the_reference_counters_recorded_by_the_survey(abi, code50, a):
    """the reference's own run in this container (seed 101, 30 calls) gave these
    frame-error / bit-error counts; the oracle driven by the restated channel must give the same."""
    cfg = abi.default_cfg(a["method"], a["max_iter"])
    fix = oa.ReferenceChannel(code50, a["seed"], 13.0).groups(a["eb_n0"], a["groups"])
    dec, st = oa.decode_mt(code50, cfg, fix, a["groups"])
    cnt = oa.Oracle(code50, cfg).count_errors(dec, None, a["groups"])
    assert cnt[0] == 960 and cnt[1] == a["survey_frame_errors"] and cnt[2] == a["survey_bit_errors"]
    assert hashlib.sha256(dec.tobytes()).hexdigest() == a["oracle_sha256"]
    assert (int(st[:, 0].sum()), int(st[:, 1].sum())) == (a["oracle_sum_I"], a["oracle_sum_J"])


ITER_ANCHORS = json.load(open(os.path.join(GOLD, "anchors_iterations.json")))["rows"]


@pytest.mark.parametrize("a", ITER_ANCHORS, ids=lambda a: "m%d_%.1fdB" % (a["method"], a["eb_n0"]))
def test_oracle_reproduces_the_iteration_means_recorded_by_the_survey(abi, code50, a):
    """BASELINE.md section 2 also records what the reference EXECUTED: layered (I) and bit-flipping (J) iterations per group,
    averaged over 25 groups of the seed-101 stream, plus three more error-counter rows (all zero) over 30 calls.  The group
    early stop, the bit-flipping break and the iteration tables all enter these numbers; the oracle (through its bit-exact AVX2
    port, tests below) reproduces every one of them."""
    cfg = abi.default_cfg(a["method"], 10)
    fix = oa.ReferenceChannel(code50, 101, 13.0).groups(a["eb_n0"], 30)
    dec, st = oa.decode_mt(code50, cfg, fix, 30, kind="avx2")
    if "mean_I_25" in a:
        assert round(float(st[:25, 0].mean()), 2) == a["mean_I_25"]
        assert round(float(st[:25, 1].mean()), 2) == a["mean_J_25"]
    if "survey_frame_errors" in a:
        cnt = oa.Oracle(code50, cfg).count_errors(dec, None, 30)
        assert cnt[0] == 960 and cnt[1] == a["survey_frame_errors"] and cnt[2] == a["survey_bit_errors"]


@pytest.mark.parametrize("name", GOLDEN)
def test_oracle_against_golden_vectors(abi, code50, name):
    z, fix, dec = load_golden(name, code50.N)
    cfg = abi.default_cfg(int(z["method"]), int(z["max_iter"]))
    out, st = oa.Oracle(code50, cfg).decode(fix, 1)
    assert np.array_equal(out, dec)
    assert np.array_equal(st, z["stats"])


def test_golden_inputs_come_from_the_restated_channel(code50):
    z, fix, _ = load_golden("m2_3p5dB_g0", code50.N)
    regen = oa.ReferenceChannel(code50, int(z["seed"]), 13.0).groups(float(z["eb_n0"]), 1)
    assert np.array_equal(regen, fix)
    assert fix.min() >= -7 and fix.max() <= 7


@pytest.mark.parametrize("method", [1, 2, 5])
def test_noiseless_codewords_are_fixed_points(abi, code50, method):
    """A valid codeword at full confidence decodes to itself (the 384 erased tail VNs are recovered by the first
    iteration, the group then stops at the next syndrome check); lane order is kept."""
    cw = np.unpackbits(np.fromfile(os.path.join(GOLD, "codeword_50gpon.bin"), dtype=np.uint8))[:code50.N].astype(np.int8)
    N, K, M = code50.N, code50.K, code50.M
    frames = np.zeros((32, N), dtype=np.int8)
    frames[1::2] = cw  # odd lanes carry the known codeword, even lanes the all-zero word
    llr = np.where(frames > 0, 7, -7).astype(np.int8)
    fix = np.concatenate([llr[:, :K].reshape(-1), llr[:, K:].reshape(-1)])
    out, st = oa.Oracle(code50, abi.default_cfg(method, 10)).decode(fix, 1)
    out = out.reshape(32, N)
    assert st.tolist() == [[1, 0]]
    assert np.array_equal(out, frames)


def test_empty_batch_and_bad_arguments(abi, code50):
    o = oa.Oracle(code50, abi.default_cfg(2, 10))
    assert o.lib.lnsfaid_oracle_decode(o.h, None, 0, None, None) == 0
    assert o.lib.lnsfaid_oracle_decode(o.h, None, 1, None, None) == -1


def test_count_errors_rule(abi, code50):
    """CalculateErrors counts information bits only; 1-2 wrong bits also count as LT3ErrBitFrame."""
    N, K = code50.N, code50.K
    dec = np.zeros((2, 32, N), dtype=np.int8)
    dec[0, 3, 5] = 1
    dec[0, 4, [1, 2, 3]] = 1
    dec[1, 0, K + 7] = 1  # parity-bit error: not counted
    dec[1, 31, [0, K - 1]] = 1
    cnt = oa.Oracle(code50, abi.default_cfg(2, 10)).count_errors(dec.reshape(-1), None, 2)
    assert cnt == [64, 3, 6, 2]


@pytest.mark.parametrize("method,max_iter,eb_n0", [(2, 10, 3.5), (2, 10, 4.2), (1, 10, 3.6), (5, 10, 3.55), (2, 6, 3.6), (5, 3, 3.0),
                                                   (4, 10, 3.6), (4, 4, 3.6), (3, 10, 3.4), (3, 3, 3.6)])
def test_avx2_port_equals_oracle(abi, code50, method, max_iter, eb_n0):
    """The vectorised CPU port used as bench.py's cpu_baseline must agree bit for bit with the pinned oracle."""
    cfg = abi.default_cfg(method, max_iter)
    fix = oa.ReferenceChannel(code50, 127, 13.0).groups(eb_n0, 6)
    ref, rst = oa.decode_mt(code50, cfg, fix, 6)
    out, st = oa.Oracle(code50, cfg, "avx2").decode(fix, 6)
    assert np.array_equal(out, ref) and np.array_equal(st, rst)


@pytest.mark.parametrize("f1,f2,eb_n0,max_iter", [(24, 24, 3.6, 10), (24, 28, 3.8, 6), (1, 6, 3.6, 3), (20, 30, 4.2, 10), (200, 2000, 3.6, 4)])
def test_nms_avx2_port_equals_oracle(abi, code50, f1, f2, eb_n0, max_iter):
    """DecodeMethod 0 (CLDPC::Decode): Factor_1 / Factor_2 are numerators over 32 in 16-bit lanes; every group runs all
    iterations.  The last case drives the 16-bit product into its wrap-around."""
    cfg = abi.default_cfg(0, max_iter)
    cfg.factor_1, cfg.factor_2 = f1, f2
    fix = oa.ReferenceChannel(code50, 137, 13.0).groups(eb_n0, 3)
    ref, rst = oa.decode_mt(code50, cfg, fix, 3)
    out, st = oa.Oracle(code50, cfg, "avx2").decode(fix, 3)
    assert np.array_equal(out, ref) and np.array_equal(st, rst)
    assert rst.tolist() == [[max_iter, 0]] * 3


def test_nms_decodes_with_sensible_factors(abi, code50):
    """Sanity of the restated NMS: 0.75-normalised min-sum (24/32) repairs nearly every frame at 3.8 dB."""
    fix = oa.ReferenceChannel(code50, 139, 13.0).groups(3.8, 3)
    cfg = abi.default_cfg(0, 10)
    cfg.factor_1, cfg.factor_2 = 24, 24
    out, _ = oa.decode_mt(code50, cfg, fix, 3)
    assert oa.Oracle(code50, cfg).count_errors(out, None, 3)[1] <= 2


def test_avx2_port_with_nondefault_constants(abi, code50):
    cfg = abi.default_cfg(1, 10)
    cfg.factor_1, cfg.factor_2 = 2, 5
    fix = oa.ReferenceChannel(code50, 131, 13.0).groups(3.5, 3)
    ref, rst = oa.decode_mt(code50, cfg, fix, 3)
    out, st = oa.Oracle(code50, cfg, "avx2").decode(fix, 3)
    assert np.array_equal(out, ref) and np.array_equal(st, rst)
    cfg = abi.default_cfg(5, 5)
    cfg.bf_alpha, cfg.bf_L0, cfg.bf_delta = 2, 2, 2
    for w in range(4):
        cfg.v2c_map[3][w][2] = 2 + (w == 1)  # class-dependent table row
    ref, rst = oa.decode_mt(code50, cfg, fix, 3)
    out, st = oa.Oracle(code50, cfg, "avx2").decode(fix, 3)
    assert np.array_equal(out, ref) and np.array_equal(st, rst)


def test_random_codewords_oracle_and_avx2_port(abi, code50, encoder):
    """Per-frame different, non-zero codewords (tests/gf2_encoder.py): noiseless words are fixed points, noisy ones
    decode identically in the oracle and the AVX2 port, and the counters use the transmitted information bits."""
    import gf2_encoder as ge
    enc = encoder
    rng = np.random.default_rng(7)
    info = rng.integers(0, 2, size=(64, code50.K), dtype=np.uint8)
    cw = enc.encode(info)
    clean = ge.to_group_layout(np.where(cw > 0, 7, -7).astype(np.int8), code50.K)
    for method in (1, 2, 5):
        cfg = abi.default_cfg(method, 10)
        out, st = oa.Oracle(code50, cfg).decode(clean, 2)
        assert np.array_equal(out.reshape(64, code50.N), cw) and st.tolist() == [[1, 0], [1, 0]]
    cfg = abi.default_cfg(2, 10)
    fix = ge.qpsk_llr(cw, 3.55, seed=3)
    ref, rst = oa.decode_mt(code50, cfg, fix, 2)
    out, st = oa.Oracle(code50, cfg, "avx2").decode(fix, 2)
    assert np.array_equal(out, ref) and np.array_equal(st, rst)
    inp = np.ascontiguousarray(info.astype(np.int8).reshape(-1))
    cnt = oa.Oracle(code50, cfg).count_errors(ref, inp, 2)
    err = (ref.reshape(64, code50.N)[:, :code50.K] != info).sum(axis=1)
    assert cnt == [64, int((err > 0).sum()), int(err.sum()), int(((err > 0) & (err < 3)).sum())]


def test_method4_is_oms_followed_by_dtbf(abi, code50):
    """DecodeMethod 4 (reference CDecoder_OMS_DTBF.cpp) has no recorded reference counters; its layered loop is textually
    Decode_OMS's and its bit-flipping stage textually Decode_FAID's, both of which are pinned.  Consistency: with the
    bit-flipping stage switched off it must equal DecodeMethod 1 exactly."""
    fix = oa.ReferenceChannel(code50, 137, 13.0).groups(3.35, 4)
    cfg4 = abi.default_cfg(4, 10)
    cfg4.max_bf_iter = 0
    a, sa = oa.decode_mt(code50, cfg4, fix, 4)
    b, sb = oa.decode_mt(code50, abi.default_cfg(1, 10), fix, 4)
    assert np.array_equal(a, b) and np.array_equal(sa, sb)
    c, sc = oa.decode_mt(code50, abi.default_cfg(4, 10), fix, 4)
    assert sc[:, 1].max() > 0 and not np.array_equal(c, b)  # the flipping stage does change frames at this Eb/N0
    # DecodeMethod 3 (Decode_OMSBF): same relation to DecodeMethod 1
    cfg3 = abi.default_cfg(3, 10)
    cfg3.max_bf_iter = 0
    a3, s3 = oa.decode_mt(code50, cfg3, fix, 4)
    assert np.array_equal(a3, b) and np.array_equal(s3, sb)
    c3, sc3 = oa.decode_mt(code50, abi.default_cfg(3, 10), fix, 4)
    assert sc3[:, 1].max() > 0 and not np.array_equal(c3, b)


@pytest.mark.parametrize("mod_type,interleave", [(2, 1), (4, 1), (6, 1), (8, 1), (2, 4), (6, 3), (8, 8)])
def test_general_channel_restatement(code50, encoder, mod_type, interleave):
    """oracle/frontend_oracle.c::lnsfaid_frontend_group: equals the QPSK / 16-QAM special cases bit for bit, and for
    every order and interleaver depth a noiseless pass returns LLRs whose decode is the sent frames (mapping, demapper and
    (de)interleaver are mutually consistent)."""
    rng = np.random.default_rng(3)
    frames = encoder.encode(rng.integers(0, 2, (32, code50.K), dtype=np.uint8))
    lib = oa.load()
    import ctypes as C
    if interleave == 1 and mod_type <= 4:
        cw = frames[0]
        a = oa.ReferenceChannel(code50, 101, 13.0, mod_type=mod_type).groups(5.0, 1, codeword=cw)
        fe = oa.Frontend()
        lib.lnsfaid_frontend_seed(C.byref(fe), 101)
        sigma = lib.lnsfaid_frontend_sigma(5.0, mod_type, oa.ReferenceChannel.RATE)
        out = np.empty(32 * code50.N, dtype=np.int8)
        assert lib.lnsfaid_frontend_group(C.byref(fe), code50.N, code50.M, cw.ctypes.data, 0, mod_type, 1, sigma, 13.0, out.ctypes.data) == 0
        assert np.array_equal(out, a)
    fe = oa.Frontend()
    lib.lnsfaid_frontend_seed(C.byref(fe), 7)
    out = np.empty(32 * code50.N, dtype=np.int8)
    fr = np.ascontiguousarray(frames, dtype=np.int8)
    scale = 40.0 if mod_type == 8 else 13.0  # 256-QAM: the least significant level is 0.077, below 1 / 13
    assert lib.lnsfaid_frontend_group(C.byref(fe), code50.N, code50.M, fr.ctypes.data, code50.N, mod_type, interleave, 0.0, scale, out.ctypes.data) == 0
    lib_cfg = oa.pyabi.default_cfg(2, 10)
    dec, _ = oa.Oracle(code50, lib_cfg, "avx2").decode(out, 1)
    assert np.array_equal(dec.reshape(32, code50.N), frames)


@pytest.mark.parametrize("mode", [1, 2])
def test_ef_elimination_variants_of_decode_faid_port_equals_oracle(abi, lib, code50, mode):
    """EF_ELIMINATION 1 / 2 of Decode_FAID (CDecoder_FAID.cpp:6, :192-203, :673-680; dead in the shipped build, so no reference
    output exists for them: PARITY UNPINNED): the scalar oracle and the AVX2 port, written independently, agree on channel
    batches around the waterfall and on nearly clean frames with a few confident errors (where the erasure acts)."""
    rng = np.random.default_rng(40 + mode)
    N, K = code50.N, code50.K
    batches = [(10, oa.ReferenceChannel(code50, 131, 13.0).groups(3.55, 3))]
    llr = np.full((3 * 32, N), -3, dtype=np.int16) + rng.integers(-2, 3, size=(3 * 32, N))
    for l in range(3 * 32):
        llr[l, rng.integers(17 * 256, 67 * 256, size=8)] = rng.integers(3, 8, size=8)
    llr = np.clip(llr, -7, 7).astype(np.int8).reshape(3, 32, N)
    batches.append((6, np.concatenate([np.concatenate([g[:, :K].reshape(-1), g[:, K:].reshape(-1)]) for g in llr])))
    for max_iter, fix in batches:
        cfg = abi.default_cfg(2, max_iter)
        assert lib.lnsfaid_cfg_ef_elimination(cfg, mode) == 0
        assert (cfg.floor_err_count, cfg.floor_iter_thresh) == ((100, 6) if mode == 1 else (20, 6))
        a, sa = oa.Oracle(code50, cfg).decode(fix, 3)
        b, sb = oa.decode_mt(code50, cfg, fix, 3, kind="avx2")
        assert np.array_equal(a, b) and np.array_equal(sa, sb)
    cfg = abi.default_cfg(5, 10)
    assert lib.lnsfaid_cfg_ef_elimination(cfg, 2) != 0  # Decode_FAID only
