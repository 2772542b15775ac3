"""Extended parity soak (not part of the test suite: minutes of CPU time): 65 536 frames per case, every DecodeMethod at two
Eb/N0 points, GPU (through the C ABI) against the vectorised CPU port, every frame and every per-group iteration count.
Run on the GPU box: gpurun -- 'python tools/gpu_soak.py [rounds]'.  Round r > 0 moves the two Eb/N0 points by 0.1 r dB, takes other seeds and cycles
through the kernel variants that share a configuration (r = 1: messages streamed through HBM, r = 2: two waves per codeword, r = 3:
two rows per lane where the configuration has that kernel)."""
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
ng = 2048
total = bad_cases = 0
# DecodeMethod 0 twice: two factors (two-rows-per-lane kernel) and one factor (four rows per lane, 16-level search)
cases = [(2, None), (5, None), (1, None), (0, (24, 26)), (0, (24, 24)), (4, None), (3, None)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for rnd, (method, factors) in [(r, c) for r in range(rounds) for c in cases]:
    for eb_n0 in (round(3.1 + 0.1 * rnd, 2), round(3.7 + 0.1 * rnd, 2)):
        cfg = abi.default_cfg(method, 10)
        if factors:
            cfg.factor_1, cfg.factor_2 = factors
        fix = oa.synth_llr(ng, code.N, eb_n0, seed=4242 + 31 * method + int(10 * eb_n0) + 1009 * rnd)
        d = abi.Decoder(code, cfg, 0, ng, lib)
        variant = ""
        try:
            if rnd % 4 == 1 and d.rows_per_lane() == 4:
                d.select_message_store(abi.MSG_HBM)
            elif rnd % 4 == 2:
                d.select_waves(2)
                variant = ", two waves per codeword"
            elif rnd % 4 == 3:
                d.select_kernel(2)
        except RuntimeError:
            pass  # the configuration has no such variant: the default kernel runs
        t0 = time.time()
        out, st = d.decode(fix, ng)
        t1 = time.time()
        rows, store = d.rows_per_lane(), d.message_store()
        d.close()
        ref, ref_st = oa.decode_mt(code, cfg, fix, ng, kind="avx2")
        t2 = time.time()
        bad = int((out != ref).reshape(ng * 32, code.N).any(axis=1).sum())
        ok = bad == 0 and np.array_equal(st, ref_st)
        total += ng * 32
        bad_cases += 0 if ok else 1
        print(("method %d%s  %.1f dB: %d frames, %d differ, stats %s  (%d rows per lane, messages %s" + variant + "; gpu incl. PCIe %.2f s, cpu port %.1f s, mean I/J %.2f/%.2f)")
              % (method, " factors %d/%d" % factors if factors else "", eb_n0, ng * 32, bad, "equal" if np.array_equal(st, ref_st) else "DIFFER",
                 rows, {1: "in registers", 2: "through HBM"}[store], t1 - t0, t2 - t1, st[:, 0].mean(), st[:, 1].mean()), flush=True)
print("soak: %d frames in %d cases, %d cases with differences" % (total, 2 * len(cases) * rounds, bad_cases))
sys.exit(1 if bad_cases else 0)
