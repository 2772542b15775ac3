"""PCIe-inclusive rate of the host-pointer entry point lnsfaid_decode (never the benchmarked number): 2048 groups from
pageable and from pinned host memory.  Run on the GPU box: gpurun -- 'python tools/gpu_host_path.py'."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_abi as oa  # noqa: E402

torch.cuda.init()
abi = oa.pyabi
lib = abi.load()
code = abi.Code50GPON(lib)
ng, N, K = 2048, code.N, code.K
cfg = abi.default_cfg(2, 10)
dec = abi.Decoder(code, cfg, 0, ng, lib)
fix = oa.synth_llr(ng, N, 3.0, seed=3)
ref = None
for name, pinned in (("pageable", False), ("pinned (torch)", True), ("registered (lnsfaid_host_register)", None)):
    src = torch.from_numpy(fix.copy())
    dst = torch.empty(ng * 32 * N, dtype=torch.int8)
    if pinned:
        src, dst = src.pin_memory(), dst.pin_memory()
    elif pinned is None:
        assert lib.lnsfaid_host_register(src.data_ptr(), src.numel()) == 0 and lib.lnsfaid_host_register(dst.data_ptr(), dst.numel()) == 0
    stats = np.zeros((ng, 2), dtype=np.int32)
    for rep in range(3):
        t0 = time.perf_counter()
        rc = lib.lnsfaid_decode(dec.ctx, src.data_ptr(), ng, dst.data_ptr(), stats.ctypes.data)
        dt = time.perf_counter() - t0
        assert rc == 0
    print("%s host buffers: %.1f ms per 65536 codewords = %.1f Gb/s of information (kernel alone %.1f ms)"
          % (name, dt * 1e3, ng * 32 * K / dt / 1e9, dec.kernel_time(reset=True)[0] / 3))
    if pinned is None:
        lib.lnsfaid_host_unregister(src.data_ptr()); lib.lnsfaid_host_unregister(dst.data_ptr())
    ref = dst.numpy().copy() if name == "pageable" else ref
    assert np.array_equal(dst.numpy(), ref), "pipelined path differs from the serial one"
dec.close()
