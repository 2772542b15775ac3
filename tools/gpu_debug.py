import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_abi as oa
pyabi = oa.pyabi
lib = pyabi.load(); code = pyabi.Code50GPON(lib)
N = code.N
for method, eb, ng, maxg in [(2, 3.0, 48, 48), (2, 3.0, 48, 2048), (2, 3.6, 48, 48)]:
    cfg = pyabi.default_cfg(method, 10, lib)
    fix = oa.synth_llr(ng, N, eb, 99)
    ref, rst = oa.decode_mt(code, cfg, fix, ng)
    dec = pyabi.Decoder(code, cfg, 0, maxg, lib)
    for rep in range(2):
        out, st = dec.decode(fix, ng)
        diff = (out != ref).reshape(ng * 32, N)
        badf = np.nonzero(diff.any(axis=1))[0]
        print(method, eb, ng, maxg, "rep", rep, "bad frames", badf.size, badf[:20].tolist(), "stats equal", np.array_equal(st, rst))
        if badf.size:
            f = badf[0]; cols = np.nonzero(diff[f])[0]
            print("  frame", f, "n diff bits", cols.size, "cols", cols[:20].tolist(), "blockcols", sorted(set((cols // 256).tolist()))[:30])
            print("  stats gpu", st[f // 32].tolist(), "oracle", rst[f // 32].tolist())
    dec.close()
