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
