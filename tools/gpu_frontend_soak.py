#!/usr/bin/env python3
"""Device front-end: the single-precision fast path against the double-precision chain (lnsfaid_frontend_set_exact) over a long run,
and the time of either.  Both run on the GPU; the comparison is a device reduction.

usage (on the GPU box): python tools/gpu_frontend_soak.py [calls] > gpurun_out/frontend_soak.txt
2048 streams per call (one headline batch: 1 157 627 904 LLRs); the calls cycle through modulation orders, Eb/N0 points, quantiser
scales and interleavers, with the draw counters advancing as in a sweep."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    pyabi = bench.load_pkg_module("pyabi")
    lib = pyabi.load()
    code = pyabi.Code50GPON(lib)
    n = 2048
    dec = pyabi.Decoder(code, pyabi.default_cfg(2, 10, lib), device=0, max_groups=n, lib=lib)
    measured, assumed = (C.c_double * 2)(), (C.c_double * 2)()
    assert lib.lnsfaid_frontend_fastpath_bounds(dec.ctx, measured, assumed) == 0
    print("error bounds over every float in [0, 1): radius %.3e (kernel assumes %.1e), cosine %.3e (assumes %.1e)" % (measured[0], assumed[0], measured[1], assumed[1]))
    plans = [(2, 13.0, 3.0, 1), (2, 13.0, 3.6, 1), (2, 13.0, 4.2, 1), (4, 12.5, 8.1, 1), (2, 13.0, 3.3, 1), (6, 13.0, 12.0, 1), (2, 9.0, 5.0, 3), (8, 13.0, 17.0, 1)]
    a = torch.empty(n * 32 * code.N, dtype=torch.int8, device="cuda")
    b = torch.empty_like(a)
    seeds = (C.c_uint32 * n)(*[101 + 2 * i for i in range(n)])
    total = bad = 0
    t_fast = t_exact = 0.0
    per_plan = {}
    for call in range(calls):
        mod_type, scale, eb_n0, il = plans[call % len(plans)]
        assert lib.lnsfaid_frontend_set_interleave(dec.ctx, il) == 0
        per_group = lib.lnsfaid_frontend_draws_per_group(dec.ctx, mod_type)
        draws = (C.c_uint64 * n)(*[call * per_group] * n)
        r = 0.8444444
        sigma = 1.0 / (r * mod_type * 10.0 ** (eb_n0 / 10.0)) ** 0.5
        for exact, buf in ((0, a), (1, b)):
            assert lib.lnsfaid_frontend_set_exact(dec.ctx, exact) == 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = lib.lnsfaid_frontend_device(dec.ctx, seeds, draws, n, mod_type, sigma, scale, None, buf.data_ptr())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert rc == 0, lib.lnsfaid_last_hip_error()
            if exact:
                t_exact += dt
            else:
                t_fast += dt
            key = (mod_type, exact)
            per_plan.setdefault(key, []).append(dt)
        bad += int((a != b).sum().item())
        total += a.numel()
        if call % 10 == 9:
            print("call %d: %d LLRs compared, %d differ" % (call + 1, total, bad), flush=True)
    print("TOTAL %d LLRs, %d differences between the fast path and the double-precision chain" % (total, bad))
    print("mean call (2048 streams, host-timed incl. the launch and one synchronisation): fast %.3f ms, exact %.3f ms" % (1e3 * t_fast / calls, 1e3 * t_exact / calls))
    for (mod_type, exact), v in sorted(per_plan.items()):
        print("  modType %d %s: %.3f ms" % (mod_type, "exact" if exact else "fast ", 1e3 * sum(v) / len(v)))
    dec.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
