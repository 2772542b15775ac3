import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
torch.cuda.init()
import oracle_abi as oa
pyabi = oa.pyabi
lib = pyabi.load(); code = pyabi.Code50GPON(lib)
n = 256
dec = pyabi.Decoder(code, pyabi.default_cfg(2, 10, lib), 0, n, lib)
d_fix = torch.empty(n * 32 * code.N, dtype=torch.int8, device="cuda")
d_out = torch.empty(n * 32 * code.N, dtype=torch.int8, device="cuda")
torch.cuda.synchronize()
seeds = (C.c_uint32 * n)(*[101 + 2 * i for i in range(n)]); draws = (C.c_uint64 * n)(*([0] * n))
sigma = oa.load().lnsfaid_frontend_sigma(3.5, 2, 0.8444444)
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.lnsfaid_frontend_device(dec.ctx, seeds, draws, n, 2, sigma, 13.0, None, d_fix.data_ptr())
    t1 = time.perf_counter()
    dec.decode_device(d_fix.data_ptr(), n, d_out.data_ptr(), None)
    t2 = time.perf_counter()
    c = dec.count_errors_device(d_out.data_ptr(), None, n)
    t3 = time.perf_counter()
    print("frontend %.2f ms  decode %.2f ms  count %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), c, dec.kernel_time(True))
