"""GPU parity: the HIP decode path (through the C ABI) against the CPU oracle, bit for bit.

Inputs come from the restated reference channel (oracle/frontend_oracle.c) so that groups contain the
mixes that matter: all-converged, partly converged (clean lanes that keep iterating while a neighbour is
dirty), bit-flipping repaired, and failed frames.
"""
import numpy as np
import pytest

import oracle_abi as oa

pytestmark = pytest.mark.gpu


def _parity(abi, code50, method, max_iter, eb_n0, n_groups, seed=101, codeword=None, scale=13.0):
    cfg = abi.default_cfg(method, max_iter)
    fix = oa.ReferenceChannel(code50, seed, scale).groups(eb_n0, n_groups, codeword)
    ref, ref_stats = oa.Oracle(code50, cfg).decode(fix, n_groups)
    dec = abi.Decoder(code50, cfg, device=0, max_groups=n_groups)
    out, stats = dec.decode(fix, n_groups)
    dec.close()
    N = code50.N
    bad = np.nonzero((out != ref).reshape(n_groups * 32, N).any(axis=1))[0]
    assert bad.size == 0, "frames differ from the oracle: %s (oracle I/J %s, gpu I/J %s)" % (
        bad[:16].tolist(), ref_stats.tolist(), stats.tolist())
    assert np.array_equal(stats, ref_stats), (stats.tolist(), ref_stats.tolist())
    return out, stats


@pytest.mark.parametrize("method", [2, 1, 5, 4, 3])
@pytest.mark.parametrize("eb_n0", [3.5, 4.2, 3.0])
def test_decode_matches_oracle(abi, code50, method, eb_n0):
    _parity(abi, code50, method, 10, eb_n0, 4)


@pytest.mark.parametrize("f1,f2,eb_n0,max_iter", [(24, 24, 3.6, 10), (24, 24, 3.0, 10), (16, 16, 3.4, 7), (32, 32, 3.8, 10), (1, 1, 3.6, 3), (45, 45, 3.6, 6),
                                                  (24, 28, 3.0, 6), (1, 6, 3.6, 3), (20, 30, 4.2, 10), (3, 40, 3.6, 5),
                                                  (200, 2000, 3.6, 4), (24, 24, 3.6, 1), (24, 24, 3.6, 0)])
def test_nms_matches_oracle(abi, code50, f1, f2, eb_n0, max_iter):
    """DecodeMethod 0 (CLDPC::Decode, reference CLDPC.cpp:214): numerators over 32.  One factor in 15 .. 2114 (the usual
    normalisations: 24, 32, 45 ...) runs on the four-rows-per-lane kernel (16-level minimum search mapped through cste());
    other equal factors take the two-rows kernel's patch path (one c1 edge per row), different factors its by-value
    first-minimum masks.  Where both kernels apply, both are run."""
    cfg = abi.default_cfg(0, max_iter)
    cfg.factor_1, cfg.factor_2 = f1, f2
    fix = oa.ReferenceChannel(code50, 149, 13.0).groups(eb_n0, 3)
    ref, ref_stats = oa.Oracle(code50, cfg).decode(fix, 3)
    dec = abi.Decoder(code50, cfg, device=0, max_groups=3)
    four = f1 == f2 and 15 <= f1 <= 2114
    assert dec.rows_per_lane() == (4 if four else 2)
    out, stats = dec.decode(fix, 3)
    assert np.array_equal(out, ref)
    assert np.array_equal(stats, ref_stats) and stats.tolist() == [[max_iter, 0]] * 3
    if four:
        dec.select_kernel(2)
        out2, stats2 = dec.decode(fix, 3)
        assert np.array_equal(out2, ref) and np.array_equal(stats2, ref_stats)
    dec.close()


@pytest.mark.parametrize("method,max_iter", [(2, 6), (2, 1), (2, 0), (1, 3), (5, 7), (1, 0), (4, 5), (4, 0), (3, 4), (3, 0)])
def test_iteration_caps(abi, code50, method, max_iter):
    _parity(abi, code50, method, max_iter, 3.6, 2, seed=103)


def _set_tables(cfg, rows, ef_rows=None):
    for it in range(6):
        for w in range(4):
            for a in range(8):
                cfg.v2c_map[it][w][a] = rows[it][a]
                if ef_rows is not None:
                    cfg.v2c_map_ef[it][w][a] = ef_rows[it][a]


@pytest.mark.parametrize("name", ["faid32", "faid2", "identity", "steep", "not_monotone", "zero_heavy"])
@pytest.mark.parametrize("method", [2, 5])
def test_uniform_table_variants(abi, code50, method, name):
    """Table sets other than the shipped one, identical for the four weight classes.  Non-decreasing sets take the kernel's
    table-after-minimum path; large c1 - c2 gaps and tables with LUT[0] != 0 / many zeros stress the argmin patch and the
    back-track sign (reference CDecoder_FAID.cpp:682); a set that is not monotone must fall back to the per-edge path."""
    tables = {
        "faid32": [[0, 1, 1, 2, 3, 3, 3, 3], [0, 1, 1, 2, 3, 3, 3, 3], [0, 1, 1, 2, 4, 4, 4, 4],
                   [1, 1, 1, 1, 4, 4, 4, 4], [1, 1, 1, 1, 5, 5, 5, 5], [1, 1, 1, 1, 6, 6, 6, 6]],  # CDecoder_FAID.cpp:51-88
        "faid2": [[0, 0, 2, 2, 2, 2, 2, 2], [0, 0, 2, 2, 2, 2, 2, 2], [1, 1, 1, 3, 3, 3, 3, 3],
                  [1, 1, 1, 4, 4, 4, 4, 4], [1, 1, 1, 5, 5, 5, 5, 5], [1, 1, 1, 6, 6, 6, 6, 6]],  # CDecoder_FAID.cpp:91-126
        "identity": [[0, 1, 2, 3, 4, 5, 6, 7]] * 6,
        "steep": [[0, 0, 0, 7, 7, 7, 7, 7], [0, 0, 5, 5, 5, 7, 7, 7], [0, 3, 3, 3, 6, 6, 7, 7]] * 2,
        "not_monotone": [[0, 2, 1, 3, 3, 3, 7, 7], [0, 1, 1, 2, 3, 3, 3, 3], [1, 0, 1, 2, 4, 4, 4, 4]] * 2,
        "zero_heavy": [[0, 0, 0, 0, 3, 3, 3, 3], [0, 0, 1, 1, 3, 3, 5, 5], [0, 0, 0, 2, 2, 2, 6, 6]] * 2,
    }[name]
    cfg = abi.default_cfg(method, 10)
    _set_tables(cfg, tables, [[1, 2, 2, 4, 5, 6, 6, 7]] * 6 if method == 5 else None)
    for eb_n0, seed in [(3.3, 151), (3.7, 157)]:
        fix = oa.ReferenceChannel(code50, seed, 13.0).groups(eb_n0, 3)
        ref, ref_stats = oa.Oracle(code50, cfg).decode(fix, 3)
        dec = abi.Decoder(code50, cfg, device=0, max_groups=3)
        out, stats = dec.decode(fix, 3)
        dec.close()
        assert np.array_equal(out, ref), (name, method, eb_n0)
        assert np.array_equal(stats, ref_stats)


@pytest.mark.parametrize("f1,f2", [(1, 2), (3, 3), (0, 1), (5, 2), (7, 7)])
@pytest.mark.parametrize("method", [1, 4])
def test_oms_offset_corner_factors(abi, code50, method, f1, f2):
    """Factor pairs that make the selective offset non-monotone in the minimum (f2 <= f1 + 1) or zero most messages.
    Factor_2 < 1 would turn the minimum 0 into -1 (outside the message alphabet): rejected at configuration time."""
    cfg = abi.default_cfg(method, 8)
    cfg.factor_1, cfg.factor_2 = 0, 0
    with pytest.raises(RuntimeError):
        abi.Decoder(code50, cfg, device=0, max_groups=2)
    cfg.factor_1, cfg.factor_2 = f1, f2
    fix = oa.ReferenceChannel(code50, 163, 13.0).groups(3.5, 2)
    ref, ref_stats = oa.Oracle(code50, cfg).decode(fix, 2)
    dec = abi.Decoder(code50, cfg, device=0, max_groups=2)
    out, stats = dec.decode(fix, 2)
    dec.close()
    assert np.array_equal(out, ref) and np.array_equal(stats, ref_stats)


@pytest.mark.parametrize("thr", [13, 1, 0, 7, 31, 32, 100])
def test_2b1c_confidence_threshold(abi, code50, thr):
    """hard2 = |En| >= threshold (reference CDecoder_FAID_2B1C.cpp:6130-6134, shipped value 13): the byte-parallel plane
    build must agree for every threshold, including "always" (<= 0) and "never" (> 31)."""
    cfg = abi.default_cfg(5, 10)
    cfg.hard2_threshold = thr
    fix = oa.ReferenceChannel(code50, 167, 13.0).groups(3.5, 3)
    ref, ref_stats = oa.Oracle(code50, cfg).decode(fix, 3)
    dec = abi.Decoder(code50, cfg, device=0, max_groups=3)
    out, stats = dec.decode(fix, 3)
    dec.close()
    assert np.array_equal(out, ref) and np.array_equal(stats, ref_stats)
    assert stats[:, 1].max() > 0  # the bit-flipping stage (the only user of hard2) ran


@pytest.mark.parametrize("method", [1, 2, 3, 4, 5])
def test_both_kernels_agree(abi, code50, method):
    """The four-rows-per-lane kernel (default for these configurations) and the two-rows-per-lane kernel on the same batch:
    identical frames and iteration counts, and both equal to the oracle."""
    cfg = abi.default_cfg(method, 10)
    n = 6
    fix = oa.ReferenceChannel(code50, 211, 13.0).groups(3.55, n)
    ref, ref_stats = oa.Oracle(code50, cfg).decode(fix, n)
    dec = abi.Decoder(code50, cfg, device=0, max_groups=n)
    assert dec.rows_per_lane() == 4
    out4, st4 = dec.decode(fix, n)
    dec.select_kernel(2)
    assert dec.rows_per_lane() == 2
    out2, st2 = dec.decode(fix, n)
    dec.select_kernel(0)
    dec.close()
    assert np.array_equal(out4, ref) and np.array_equal(st4, ref_stats)
    assert np.array_equal(out2, ref) and np.array_equal(st2, ref_stats)


@pytest.mark.parametrize("method", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("eb_n0", [3.55, 4.0])
def test_messages_in_registers_and_streamed_through_hbm_agree(abi, code50, method, eb_n0):
    """The four-rows kernel with the compressed messages in registers (default for the 12-layer code) and with the messages
    streamed through HBM (the instance codes of more layers run on): identical frames and iteration counts, both equal to the
    oracle.  The batch has groups whose codewords park and resume (clean lanes beside dirty ones), i.e. messages that travel
    from the registers to HBM and back."""
    cfg = abi.default_cfg(method, 10)
    n = 6
    fix = oa.ReferenceChannel(code50, 223, 13.0).groups(eb_n0, n)
    ref, ref_stats = oa.Oracle(code50, cfg).decode(fix, n)
    dec = abi.Decoder(code50, cfg, device=0, max_groups=n)
    assert dec.rows_per_lane() == 4 and dec.message_store() == abi.MSG_REGISTERS
    out_r, st_r = dec.decode(fix, n)
    dec.select_message_store(abi.MSG_HBM)
    assert dec.message_store() == abi.MSG_HBM
    out_h, st_h = dec.decode(fix, n)
    dec.select_message_store(0)
    assert dec.message_store() == abi.MSG_REGISTERS
    dec.close()
    assert np.array_equal(out_r, ref) and np.array_equal(st_r, ref_stats)
    assert np.array_equal(out_h, ref) and np.array_equal(st_h, ref_stats)


@pytest.mark.parametrize("method", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("eb_n0", [3.55, 4.0, 3.0])
def test_two_waves_per_codeword_agree(abi, code50, method, eb_n0):
    """The experimental two-waves-per-codeword build of the four-rows kernel (lnsfaid_kernel5.hip: the edges of a layer dealt to
    two wavefronts that exchange partial minima, sign products and the new magnitudes through LDS, four barriers per layer): the
    same frames and iteration counts as the one-wave kernel and the oracle.  The 3.55 / 4.0 dB batches hold groups whose
    codewords park and resume, 3.0 dB runs every layered and every bit-flipping iteration."""
    cfg = abi.default_cfg(method, 10)
    n = 6
    fix = oa.ReferenceChannel(code50, 229, 13.0).groups(eb_n0, n)
    ref, ref_stats = oa.Oracle(code50, cfg).decode(fix, n)
    dec = abi.Decoder(code50, cfg, device=0, max_groups=n)
    assert dec.kernel_waves() == 1
    out1, st1 = dec.decode(fix, n)
    dec.select_waves(2)
    assert dec.kernel_waves() == 2 and dec.rows_per_lane() == 4 and dec.message_store() == abi.MSG_HBM
    assert dec.kernel_residency() == (8, 8)  # 128 registers per wave: LDS still decides
    out2, st2 = dec.decode(fix, n)
    dec.select_waves(0)
    assert dec.kernel_waves() == 1 and dec.message_store() == abi.MSG_REGISTERS
    dec.close()
    assert np.array_equal(out1, ref) and np.array_equal(st1, ref_stats)
    assert np.array_equal(out2, ref) and np.array_equal(st2, ref_stats)


def test_two_waves_per_codeword_selection_rules(abi, lib, code50):
    """Refused where the kernel is not built: DecodeMethod 0, the erasing EF_ELIMINATION 2, configurations of the two-rows kernel."""
    dec = abi.Decoder(code50, abi.default_cfg(0, 4), device=0, max_groups=1)
    with pytest.raises(RuntimeError):
        dec.select_waves(2)
    dec.close()
    cfg = abi.default_cfg(2, 10)
    assert lib.lnsfaid_cfg_ef_elimination(cfg, 2) == 0
    dec = abi.Decoder(code50, cfg, device=0, max_groups=1)
    with pytest.raises(RuntimeError):
        dec.select_waves(2)
    dec.close()
    cfg = abi.default_cfg(2, 10)
    assert lib.lnsfaid_cfg_ef_elimination(cfg, 1) == 0
    dec = abi.Decoder(code50, cfg, device=0, max_groups=1)
    dec.select_waves(2)  # the _ef tables are a run-time switch of the same layer step
    assert dec.kernel_waves() == 2
    dec.close()
    dec = abi.Decoder(code50, abi.default_cfg(1, 10), device=0, max_groups=1)
    dec.select_kernel(2)
    with pytest.raises(RuntimeError):
        dec.select_waves(2)
    with pytest.raises(RuntimeError):
        dec.select_waves(3)
    dec.close()


def test_kernel_residency_is_what_the_lds_footprint_allows(abi, code50):
    """Eight codewords per CU for the 50G-PON code, for every kernel instance a shipped configuration selects: a build that loses
    residency to registers (or a static __shared__ in a decode kernel) fails here, not silently at 60 % of the throughput."""
    for method in (0, 1, 2, 3, 4, 5):
        dec = abi.Decoder(code50, abi.default_cfg(method, 10), device=0, max_groups=1)
        wg, lim = dec.kernel_residency()
        assert (wg, lim) == (8, 8), (method, dec.rows_per_lane(), wg, lim)
        if dec.rows_per_lane() == 4:
            for where in (abi.MSG_HBM, 0):
                dec.select_message_store(where)
                assert dec.kernel_residency() == (8, 8), (method, where)
            dec.select_kernel(2)
            assert dec.kernel_residency() == (8, 8), (method, "two rows per lane")
        dec.close()


def test_kernel_selection_rules(abi, code50):
    """NMS with two factors or a factor outside 15 .. 2114, and tables that differ between weight classes, stay on the
    two-rows-per-lane kernel; forcing the other is refused.  NMS with one usual factor runs four rows per lane."""
    for f1, f2, rows in [(24, 24, 4), (32, 32, 4), (45, 45, 4), (15, 15, 4), (24, 28, 2), (14, 14, 2), (1, 6, 2), (2185, 2185, 2)]:
        cfg = abi.default_cfg(0, 4)
        cfg.factor_1, cfg.factor_2 = f1, f2
        dec = abi.Decoder(code50, cfg, device=0, max_groups=1)
        assert dec.rows_per_lane() == rows, (f1, f2)
        if rows == 2:
            with pytest.raises(RuntimeError):
                dec.select_kernel(4)
        else:
            assert dec.message_store() == abi.MSG_HBM  # the 16-level search leaves no registers for the messages
            with pytest.raises(RuntimeError):
                dec.select_message_store(abi.MSG_REGISTERS)
        dec.close()
    cfg = abi.default_cfg(2, 4)
    cfg.v2c_map[0][1][3] = 3  # weight class 1 differs from class 0 in iteration 1
    dec = abi.Decoder(code50, cfg, device=0, max_groups=1)
    assert dec.rows_per_lane() == 2
    dec.close()
