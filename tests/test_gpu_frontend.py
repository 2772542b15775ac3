"""Device front-end against the restated reference channel (oracle/frontend_oracle.c).

The Wichmann-Hill jump-ahead, the single-precision stages and the quantiser are exact; Box-Muller runs the device's
double-precision log / cos, so an LLR may differ from glibc's result in rare rounding cases.  The tests bound that
rate and check that the decoder's output on device-generated input still equals the oracle's on the same input."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_abi as oa

pytestmark = pytest.mark.gpu


def _device_groups(abi, lib, dec, code50, seeds, draws, mod_type, eb_n0, scale, codeword=None):
    import torch
    n = len(seeds)
    d_fix = torch.empty(n * 32 * code50.N, dtype=torch.int8, device="cuda")
    torch.cuda.synchronize()
    sigma = oa.load().lnsfaid_frontend_sigma(eb_n0, mod_type, oa.ReferenceChannel.RATE)
    s = (C.c_uint32 * n)(*seeds)
    d = (C.c_uint64 * n)(*draws)
    cw = None if codeword is None else np.ascontiguousarray(codeword, dtype=np.int8).ctypes.data
    rc = lib.lnsfaid_frontend_device(dec.ctx, s, d, n, mod_type, sigma, scale, cw, d_fix.data_ptr())
    assert rc == 0, lib.lnsfaid_last_hip_error()
    return d_fix


@pytest.mark.parametrize("mod_type,scale,eb_n0", [(2, 13.0, 3.6), (4, 12.5, 8.1)], ids=["qpsk", "16qam"])
def test_device_frontend_matches_host_generator(abi, lib, code50, mod_type, scale, eb_n0):
    seeds = [101, 103, 1019]
    cw = np.unpackbits(np.fromfile(os.path.join(oa.ROOT, "tests", "golden", "codeword_50gpon.bin"), dtype=np.uint8))[:code50.N].astype(np.int8)
    dec = abi.Decoder(code50, abi.default_cfg(2, 10), 0, 4)
    per_group = lib.lnsfaid_frontend_draws_per_group(dec.ctx, mod_type)
    assert per_group == 32 * code50.N // mod_type * 4
    host = [oa.ReferenceChannel(code50, s, scale, mod_type=mod_type) for s in seeds]
    total = mismatch = 0
    for call in range(3):  # three consecutive calls of every stream: the draw counters advance
        codeword = cw if call == 1 else None
        want = np.concatenate([h.groups(eb_n0, 1, codeword) for h in host])
        draws = [call * per_group] * len(seeds)
        got = _device_groups(abi, lib, dec, code50, seeds, draws, mod_type, eb_n0, scale, codeword).cpu().numpy()
        diff = got != want
        mismatch += int(diff.sum())
        total += diff.size
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1  # a rounding case moves an LLR by one level at most
    print("device front-end: %d of %d LLRs differ from the host generator" % (mismatch, total))
    assert mismatch <= total * 1e-5
    dec.close()


def test_decode_of_device_generated_batch(abi, lib, code50):
    """End to end on the device: generate 64 streams, decode, count; the oracle decodes the very same LLRs."""
    n = 64
    cfg = abi.default_cfg(2, 10)
    dec = abi.Decoder(code50, cfg, 0, n)
    import torch
    seeds = [101 + 2 * i for i in range(n)]
    d_fix = _device_groups(abi, lib, dec, code50, seeds, [0] * n, 2, 3.55, 13.0)
    d_out = torch.empty(n * 32 * code50.N, dtype=torch.int8, device="cuda")
    d_st = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    dec.decode_device(d_fix.data_ptr(), n, d_out.data_ptr(), d_st.data_ptr())
    cnt = dec.count_errors_device(d_out.data_ptr(), None, n)
    fix = d_fix.cpu().numpy()
    ref, rst = oa.decode_mt(code50, cfg, fix, n, kind="avx2")
    assert np.array_equal(d_out.cpu().numpy(), ref) and np.array_equal(d_st.cpu().numpy(), rst)
    assert cnt == oa.Oracle(code50, cfg).count_errors(ref, None, n)
    dec.close()


@pytest.mark.parametrize("mod_type,interleave,scale,eb_n0", [(6, 1, 12.5, 14.0), (8, 1, 40.0, 19.0), (2, 2, 13.0, 3.8), (4, 4, 12.5, 8.6),
                                                             (6, 3, 12.5, 14.0), (8, 8, 40.0, 19.0)],
                         ids=["64qam", "256qam", "qpsk_il2", "16qam_il4", "64qam_il3", "256qam_il8"])
def test_device_frontend_higher_orders_and_interleaver(abi, lib, code50, encoder, mod_type, interleave, scale, eb_n0):
    """64- / 256-QAM (reference CModulate.cpp:6-7, :216-264, :327-356) and the block interleaver InterleaveModType > 1
    (CModulate.cpp:95-212): device generator against the restated channel, with 32 different frames per stream
    (lnsfaid_frontend_set_frames).  No reference output exists for these modes: the two restatements check each other."""
    seeds = [101, 107]
    K, N = code50.K, code50.N
    rng = np.random.default_rng(5)
    frames = [encoder.encode(rng.integers(0, 2, (32, K), dtype=np.uint8)) for _ in seeds]  # [32, N] each
    dec = abi.Decoder(code50, abi.default_cfg(2, 10), 0, 2)
    out_layout = np.concatenate([np.concatenate([f[:, :K].reshape(-1), f[:, K:].reshape(-1)]) for f in frames]).astype(np.int8)
    info = np.concatenate([f[:, :K].reshape(-1) for f in frames]).astype(np.int8)
    assert lib.lnsfaid_frontend_set_frames(dec.ctx, out_layout.ctypes.data, info.ctypes.data, 2) == 0
    assert lib.lnsfaid_frontend_set_interleave(dec.ctx, interleave) == 0
    per_group = lib.lnsfaid_frontend_draws_per_group(dec.ctx, mod_type)
    assert per_group == 32 * N // mod_type * 4
    want = np.concatenate([oa.ReferenceChannel(code50, s, scale, mod_type=mod_type, interleave=interleave).groups(eb_n0, 1, frames=f)
                           for s, f in zip(seeds, frames)])
    got = _device_groups(abi, lib, dec, code50, seeds, [0, 0], mod_type, eb_n0, scale).cpu().numpy()
    diff = got != want
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and diff.sum() <= diff.size * 1e-5
    # the mapping is consistent end to end: at this Eb/N0 the decoder returns the sent frames
    ref, _ = oa.decode_mt(code50, abi.default_cfg(2, 10), want, 2, kind="avx2")
    assert np.array_equal(ref.reshape(2, 32, N), np.stack(frames))
    assert lib.lnsfaid_frontend_set_interleave(dec.ctx, 7) != 0  # must divide the frame length
    dec.close()


def test_fast_path_error_bounds_hold_for_every_float(abi, lib, code50):
    """The single-precision Box-Muller of the front-end kernel decides a quantised LLR only when no threshold lies within its error
    bound.  The bound rests on two constants - the largest distance of the hardware sqrt(-2 ln(1 - u)) and cos(2 pi u) from
    double precision - which the library measures over EVERY float in [0, 1); the kernel assumes at least twice the measured."""
    dec = abi.Decoder(code50, abi.default_cfg(2, 10), 0, 1)
    measured, assumed = (C.c_double * 2)(), (C.c_double * 2)()
    assert lib.lnsfaid_frontend_fastpath_bounds(dec.ctx, measured, assumed) == 0, lib.lnsfaid_last_hip_error()
    dec.close()
    print("front-end fast path: radius error %.3e (assumed %.1e), cosine error %.3e (assumed %.1e)" % (measured[0], assumed[0], measured[1], assumed[1]))
    assert 0.0 < measured[0] * 2.0 <= assumed[0], (measured[0], assumed[0])
    assert 0.0 < measured[1] * 2.0 <= assumed[1], (measured[1], assumed[1])


@pytest.mark.parametrize("mod_type,scale,eb_n0,interleave", [(2, 13.0, 3.0, 1), (2, 13.0, 4.2, 1), (2, 7.3, 3.6, 1), (2, 13.0, 3.6, 3), (4, 12.5, 8.1, 1),
                                                             (6, 13.0, 12.0, 2), (8, 13.0, 16.0, 1)])
def test_fast_path_equals_the_double_precision_chain(abi, lib, code50, mod_type, scale, eb_n0, interleave):
    """Fast path against lnsfaid_frontend_set_exact(1) - the reference's chain with its integer generator, IEEE float divisions
    and double-precision log / cos / sqrt for every symbol - on 256 streams x 3 calls (434 M LLRs for QPSK): identical bytes."""
    import torch
    n = 256
    dec = abi.Decoder(code50, abi.default_cfg(2, 10), 0, n)
    assert lib.lnsfaid_frontend_set_interleave(dec.ctx, interleave) == 0
    per_group = lib.lnsfaid_frontend_draws_per_group(dec.ctx, mod_type)
    seeds = [101 + 2 * i for i in range(n)]
    cw = np.unpackbits(np.fromfile(os.path.join(oa.ROOT, "tests", "golden", "codeword_50gpon.bin"), dtype=np.uint8))[:code50.N].astype(np.int8)
    for call in range(3):
        draws = [call * per_group + 4 * i for i in range(n)]  # (also draw counts that are not multiples of a group)
        codeword = cw if call == 2 else None
        assert lib.lnsfaid_frontend_set_exact(dec.ctx, 0) == 0
        fast = _device_groups(abi, lib, dec, code50, seeds, draws, mod_type, eb_n0, scale, codeword)
        assert lib.lnsfaid_frontend_set_exact(dec.ctx, 1) == 0
        exact = _device_groups(abi, lib, dec, code50, seeds, draws, mod_type, eb_n0, scale, codeword)
        assert int((fast != exact).sum().item()) == 0, (mod_type, call)
        assert int((fast != 0).sum().item()) > fast.numel() // 2  # (not an all-zero buffer)
    assert lib.lnsfaid_frontend_set_exact(dec.ctx, 2) != 0
    dec.close()


@pytest.mark.parametrize("mod_type,scale,eb_n0", [(2, 13.0, 3.6), (4, 12.5, 8.1), (6, 13.0, 12.0)])
def test_unaligned_output_buffer_takes_the_byte_store_path(abi, lib, code50, mod_type, scale, eb_n0):
    """Without interleaver the kernel writes 16 bytes per store (QPSK) or one symbol per store (higher orders) when the output
    allows it; an output buffer that does not start on a multiple of 16 gets the same bytes through single-byte stores."""
    import torch
    n = 8
    dec = abi.Decoder(code50, abi.default_cfg(2, 10), 0, n)
    seeds = (C.c_uint32 * n)(*[211 + 2 * i for i in range(n)])
    draws = (C.c_uint64 * n)(*[0] * n)
    sigma = oa.load().lnsfaid_frontend_sigma(eb_n0, mod_type, oa.ReferenceChannel.RATE)
    size = n * 32 * code50.N
    aligned = torch.zeros(size, dtype=torch.int8, device="cuda")
    shifted = torch.zeros(size + 16, dtype=torch.int8, device="cuda")
    assert aligned.data_ptr() % 16 == 0
    assert lib.lnsfaid_frontend_device(dec.ctx, seeds, draws, n, mod_type, sigma, scale, None, aligned.data_ptr()) == 0
    for off in (1, 4):
        shifted.zero_()
        assert lib.lnsfaid_frontend_device(dec.ctx, seeds, draws, n, mod_type, sigma, scale, None, shifted.data_ptr() + off) == 0
        torch.cuda.synchronize()
        assert torch.equal(shifted[off:off + size], aligned)
        assert int(shifted[:off].abs().sum().item()) == 0 and int(shifted[off + size:].abs().sum().item()) == 0
    dec.close()
