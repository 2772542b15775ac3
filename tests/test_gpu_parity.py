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


@pytest.mark.parametrize("f1,f2,eb_n0,max_iter", [(24, 24, 3.6, 10), (24, 28, 3.0, 6), (1, 6, 3.6, 3), (20, 30, 4.2, 10), (3, 40, 3.6, 5),
                                                  (200, 2000, 3.6, 4), (24, 24, 3.6, 1), (24, 24, 3.6, 0)])
def test_nms_matches_oracle(abi, code50, f1, f2, eb_n0, max_iter):
    """DecodeMethod 0 (CLDPC::Decode, reference CLDPC.cpp:214): by-value first-minimum masks, numerators over 32."""
    cfg = abi.default_cfg(0, max_iter)
    cfg.factor_1, cfg.factor_2 = f1, f2
    fix = oa.ReferenceChannel(code50, 149, 13.0).groups(eb_n0, 3)
    ref, ref_stats = oa.Oracle(code50, cfg).decode(fix, 3)
    dec = abi.Decoder(code50, cfg, device=0, max_groups=3)
    out, stats = dec.decode(fix, 3)
    dec.close()
    assert np.array_equal(out, ref)
    assert np.array_equal(stats, ref_stats) and stats.tolist() == [[max_iter, 0]] * 3


@pytest.mark.parametrize("method,max_iter", [(2, 6), (2, 1), (2, 0), (1, 3), (5, 7), (1, 0), (4, 5), (4, 0), (3, 4), (3, 0)])
def test_iteration_caps(abi, code50, method, max_iter):
    _parity(abi, code50, method, max_iter, 3.6, 2, seed=103)
