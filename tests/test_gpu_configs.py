"""BASELINE.json configurations end to end on the GPU (VERDICT r01 "next" item 5):
  config 1  BPSK / DecodeMethod 1 plumbing through the CLDPC / CSimulate driver (its MKL noise stream cannot be reproduced, so
            the driver dumps the fixInput it decoded and the oracle is fed with exactly that)
  config 3  FER sweep 3.3 ... 3.8 dB with the reference's stop rule, counters per point against the CPU port
  config 4  counters summed over RCCL through the C ABI (world of one on a one-GPU box) and the driver's one-process-per-GPU mode
  config 5  DecodeMethod 5 (hybrid 2B1C), 16-QAM, scale 12.5 at batch size against the CPU port
"""
import ctypes as C
import os
import re
import subprocess
import threading

import numpy as np
import pytest

import oracle_abi as oa

pytestmark = pytest.mark.gpu

EXE = os.path.join(oa.PKG_DIR, "host", "lnsfaid_sim")
SEEDS = [101, 103, 107, 109, 113, 127, 131, 137]  # reference CSimulate.cpp:11-17, first entries


def _profile(tmp_path, start, end, method, mod_type=2, scale=13.0, step=0.1):
    prof = open(os.path.join(oa.PKG_DIR, "host", "Profile.txt")).read()
    prof = prof.replace("StartSNR: 3.3", "StartSNR: %g" % start).replace("EndSNR: 3.85", "EndSNR: %g" % end)
    prof = prof.replace("SNRPass: 0.1", "SNRPass: %g" % step)
    prof = prof.replace("DecodeMethod: 2", "DecodeMethod: %d" % method).replace("modType: 2", "modType: %d" % mod_type)
    prof = prof.replace("scale: 13", "scale: %g" % scale)
    (tmp_path / "Profile.txt").write_text(prof)


def _rows(stdout):
    """Eb/N0 -> [TestFrame, ErrorFrame, ErrorBits, LT3ErrBitFrame] from the driver's table (same columns as Result.txt)."""
    out = {}
    for l in stdout.splitlines():
        f = l.split()
        if len(f) >= 8 and re.match(r"^\d+(\.\d+)?$", f[0]):
            out[round(float(f[0]), 2)] = [int(f[1]), int(f[2]), int(f[3]), int(f[6])]
    return out


def test_config1_bpsk_oms_through_the_driver(abi, code50, tmp_path):
    """modType 1 (BPSK, reference CSimulate.cpp:119-124), DecodeMethod 1, 10 iterations: the driver's counters for one stream
    and one round equal the oracle's on the fixInput the driver dumped (50 calls of one group)."""
    _profile(tmp_path, 3.45, 3.5, method=1, mod_type=1)
    dump = tmp_path / "fix.bin"
    res = subprocess.run([EXE, "--streams", "1", "--max-rounds", "1", "--dump-fixinput", str(dump)], cwd=tmp_path,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    got = _rows(res.stdout)[3.45]
    fix = np.fromfile(dump, dtype=np.int8)
    assert fix.size == 50 * 32 * code50.N and fix.min() >= -7 and fix.max() <= 7
    cfg = abi.default_cfg(1, 10)
    dec, _ = oa.decode_mt(code50, cfg, fix, 50)  # the scalar oracle
    want = oa.Oracle(code50, cfg).count_errors(dec, None, 50)
    assert got == want and got[0] == 1600
    # BPSK at 3.45 dB with OMS: most frames decode, some do not (a vacuous 0 == 0 would prove nothing)
    assert 0 < got[1] < 1600


def test_config3_fer_sweep_with_the_reference_stop_rule(abi, code50, tmp_path):
    """DecodeMethod 2, Eb/N0 3.3 ... 3.8 step 0.1, 4 streams (reference threads 0..3): every point runs rounds of 50 calls per
    stream until TestFrame >= 1000 and ErrorFrame >= 20 (reference main.cpp:164, :209), capped at 2 rounds where the error
    rate is too low for that; counters of every point, as printed and as appended to Result.txt, against the CPU port fed by
    the restated reference channel (whose generators run on across the points exactly like the driver's)."""
    streams, cap = 4, 2
    _profile(tmp_path, 3.3, 3.85, method=2)
    res = subprocess.run([EXE, "--streams", str(streams), "--max-rounds", str(cap)], cwd=tmp_path, capture_output=True, text=True,
                         timeout=1500)
    assert res.returncode == 0, res.stderr
    got = _rows(res.stdout)
    points = [3.3, 3.4, 3.5, 3.6, 3.7, 3.8]
    assert sorted(got) == points
    cfg = abi.default_cfg(2, 10)
    chans = [oa.ReferenceChannel(code50, SEEDS[s], 13.0) for s in range(streams)]
    eb32 = np.float32(3.3)  # the driver's loop variable is a float that accumulates SNRPass (reference main.cpp:136)
    for eb in points:
        total = [0, 0, 0, 0]
        running = [[0, 0, 0, 0] for _ in range(streams)]  # CSimulate's counters run on over the rounds of a point ...
        rounds = 0
        while total[0] < 1000 or total[1] < 20:           # reference main.cpp:164
            per_stream = [None] * streams

            def work(s):
                fix = chans[s].groups(float(eb32), 50)
                dec, _ = oa.decode_mt(code50, cfg, fix, 50, threads=4, kind="avx2")
                per_stream[s] = oa.Oracle(code50, cfg).count_errors(dec, None, 50)

            th = [threading.Thread(target=work, args=(s,)) for s in range(streams)]
            [t.start() for t in th]
            [t.join() for t in th]
            for s in range(streams):
                running[s] = [a + b for a, b in zip(running[s], per_stream[s])]
                total = [a + b for a, b in zip(total, running[s])]  # ... and main adds the running totals every round (:174-182)
            rounds += 1
            if total[0] > 1000 and total[1] > 20:         # reference main.cpp:209
                break
            if rounds >= cap:
                break
        assert got[eb] == total, (eb, got[eb], total)
        eb32 = np.float32(eb32 + np.float32(0.1))
    # Result.txt carries the same rows
    res_rows = _rows(open(tmp_path / "Result.txt").read())
    assert res_rows == got
    assert got[3.3][1] > got[3.5][1] > got[3.7][1]  # the waterfall is inside the sweep


def test_config4_counters_over_rccl_world_of_one(abi, lib, code50):
    """lnsfaid_comm_unique_id / lnsfaid_comm_init / lnsfaid_allreduce_counters on a communicator of one rank: the sum over
    the ranks is the rank's own counters (on an 8-GPU node the same calls sum over the 8 contexts)."""
    cfg = abi.default_cfg(2, 10)
    dec = abi.Decoder(code50, cfg, device=0, max_groups=2)
    with pytest.raises(RuntimeError):
        dec.allreduce_counters([1, 2, 3, 4])  # no communicator yet
    cid = (C.c_uint8 * 128)()
    assert lib.lnsfaid_comm_unique_id(cid) == 0
    dec.comm_init(1, 0, cid)
    fix = oa.ReferenceChannel(code50, 101, 13.0).groups(3.5, 2)
    out, _ = dec.decode(fix, 2)
    local = dec.count_errors(out, None, 2)
    assert dec.allreduce_counters(local) == local
    assert dec.allreduce_counters([2**40 + 5, 0, 2**63, 7]) == [2**40 + 5, 0, 2**63, 7]
    assert lib.lnsfaid_comm_destroy(dec.ctx) == 0
    dec.close()


def test_config4_driver_one_process_per_gpu(abi, code50, tmp_path):
    """`lnsfaid_sim --ranks N --rank r --comm-file F`: N = number of visible GPUs (1 on the test box, where the all-reduce is
    over a communicator of one).  The printed counters equal the single-process run's over the same 4 streams."""
    import torch
    n = max(1, min(torch.cuda.device_count(), 4))
    _profile(tmp_path, 3.5, 3.55, method=2)
    ref = subprocess.run([EXE, "--streams", "4", "--max-rounds", "1", "--device-frontend"], cwd=tmp_path, capture_output=True,
                         text=True, timeout=600)
    assert ref.returncode == 0, ref.stderr
    want = _rows(ref.stdout)[3.5]
    procs = [subprocess.Popen([EXE, "--streams", "4", "--max-rounds", "1", "--device-frontend", "--ranks", str(n), "--rank", str(r),
                               "--comm-file", str(tmp_path / "rccl.id")], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(n)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1] for o in outs]
    assert _rows(outs[0][0])[3.5] == want
    assert want[0] == 4 * 50 * 32


def test_config5_hybrid_2b1c_16qam_at_batch_size(abi, code50):
    """DecodeMethod 5, 16-QAM LLR statistics (per symbol r, i, |r| - c, |i| - c; reference CModulate.cpp:283-293), scale 12.5
    (reference README.md:20), 1024 groups = 32 768 frames around the waterfall: every frame and every group's (I, J) against
    the CPU port."""
    ng, N = 1024, code50.N
    cfg = abi.default_cfg(5, 10)
    rng = np.random.default_rng(55)
    for eb in (8.1, 8.5):
        sigma = 1.0 / np.sqrt(oa.ReferenceChannel.RATE * 4 * 10.0 ** (0.1 * eb)) / np.sqrt(2.0)
        ri = rng.standard_normal((ng * 32 * N // 4, 2), dtype=np.float32) * np.float32(sigma) - np.float32(0.316228)
        x = np.concatenate([ri, np.abs(ri) - np.float32(0.6324555)], axis=1).reshape(-1)
        fix = np.clip(np.trunc(x * np.float32(12.5)), -7, 7).astype(np.int8)
        d = abi.Decoder(code50, cfg, 0, ng)
        out, st = d.decode(fix, ng)
        d.close()
        ref, ref_st = oa.decode_mt(code50, cfg, fix, ng, kind="avx2")
        bad = np.nonzero((out != ref).reshape(ng * 32, N).any(axis=1))[0]
        assert bad.size == 0, "frames differ from the CPU port at %.1f dB: %s" % (eb, bad[:16].tolist())
        assert np.array_equal(st, ref_st)
        assert 0 < st[:, 0].mean() < 10  # inside the waterfall: early stop active, not everything converges at once


def test_contexts_decoding_concurrently_through_the_call_combiner(abi, code50):
    """The reference's call shape (one CLDPC per worker thread, one group of 32 frames per call: CSimulate.cpp:136-164,
    main.cpp:164-172) through the binding of INTEGRATION.md 2: several one-group contexts on one GPU, every one called from its
    own host thread at the same time.  The library combines such calls into common launches (lnsfaid_capi.hip, call combiner);
    every call must still return exactly its own group's oracle result - also when the threads use different DecodeMethods
    (batches are formed per configuration) and when a context is created or destroyed while the others are decoding."""
    import threading
    n_threads, n_calls = 6, 5
    methods = [2, 2, 2, 5, 1, 2]
    fixes, refs = [], []
    for t in range(n_threads):
        cfg = abi.default_cfg(methods[t], 10)
        fix = oa.ReferenceChannel(code50, 300 + t, 13.0).groups(3.5 if t % 2 else 4.0, n_calls)
        ref, ref_stats = oa.Oracle(code50, cfg).decode(fix, n_calls)
        fixes.append(fix.reshape(n_calls, -1))
        refs.append((ref.reshape(n_calls, -1), ref_stats))
    errors = []
    start = threading.Barrier(n_threads)

    def worker(t):
        try:
            dec = abi.Decoder(code50, abi.default_cfg(methods[t], 10), device=0, max_groups=1)
            start.wait()
            for rep in range(2):
                for c in range(n_calls):
                    out, st = dec.decode(np.ascontiguousarray(fixes[t][c]), 1)
                    if not np.array_equal(out, refs[t][0][c]) or st.tolist() != [refs[t][1][c].tolist()]:
                        errors.append((t, rep, c, st.tolist(), refs[t][1][c].tolist()))
                if t == 0 and rep == 0:  # a member leaves and a new one joins while the others keep decoding
                    dec.close()
                    dec = abi.Decoder(code50, abi.default_cfg(methods[t], 10), device=0, max_groups=1)
            dec.close()
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not any(th.is_alive() for th in threads), "a decode call did not return"
    assert not errors, errors[:4]
