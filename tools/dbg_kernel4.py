import sys, numpy as np
sys.path.insert(0, "tests")
import torch; torch.cuda.init()
import oracle_abi as oa
abi = oa.pyabi
code = abi.Code50GPON()
N = code.N
for method, mi in ((1, 1), (1, 2), (1, 10), (2, 10)):
    cfg = abi.default_cfg(method, mi)
    if method == 2: cfg.max_bf_iter = 0
    fix = oa.ReferenceChannel(code, 101, 13.0).groups(3.4, 1)
    ref, rs = oa.Oracle(code, cfg).decode(fix, 1)
    dec = abi.Decoder(code, cfg, device=0, max_groups=1)
    out, st = dec.decode(fix, 1)
    dec.select_kernel(2)
    out2, st2 = dec.decode(fix, 1)
    dec.close()
    d = (out != ref).reshape(32, N)
    print("method", method, "max_iter", mi, "stats", st.tolist(), rs.tolist(), "k2 ok", np.array_equal(out2, ref), "diff bits per frame", d.sum(axis=1)[:8].tolist(), "total", int(d.sum()))
    if d.sum():
        cols = d.sum(axis=0).reshape(-1, 256).sum(axis=1)
        print(" per block column:", cols.tolist())
        f0 = np.nonzero(d[0])[0]
        print(" frame0 first diffs:", f0[:20].tolist(), "ref", ref[:N][f0[:20]].tolist(), "out", out[:N][f0[:20]].tolist())
