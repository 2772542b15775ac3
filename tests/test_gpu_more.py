"""GPU parity, part 2: committed golden vectors, reference-anchored digests, the device-pointer entry points,
BASELINE-size batches through size-independent properties, edge cases, and the C++ host driver."""
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_abi as oa
from test_oracle import ANCHORS, GOLD, GOLDEN, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", GOLDEN)
def test_golden_vectors(abi, code50, name):
    z, fix, dec = load_golden(name, code50.N)
    d = abi.Decoder(code50, abi.default_cfg(int(z["method"]), int(z["max_iter"])), 0, 1)
    out, st = d.decode(fix, 1)
    inp = None
    if int(z["known_codeword"]):
        cw = np.unpackbits(np.fromfile(os.path.join(GOLD, "codeword_50gpon.bin"), dtype=np.uint8))[:code50.N].astype(np.int8)
        inp = np.ascontiguousarray(np.tile(cw[:code50.K], 32))
    cnt = d.count_errors(out, inp, 1)
    d.close()
    assert np.array_equal(out, dec)
    assert np.array_equal(st, z["stats"])
    assert cnt == [int(x) for x in z["counters"]]


@pytest.mark.parametrize("a", [ANCHORS[1], ANCHORS[2], ANCHORS[5], ANCHORS[6], ANCHORS[7]],
                         ids=lambda a: "m%d_%.1fdB_it%d" % (a["method"], a["eb_n0"], a["max_iter"]))
def test_reference_anchored_runs(abi, code50, a):
    """30 calls of seed 101 at the survey's Eb/N0 points: the GPU must give the reference's recorded counters
     and the oracle's digest of all hard decisions."""
    fix = oa.ReferenceChannel(code50, a["seed"], 13.0).groups(a["eb_n0"], a["groups"])
    d = abi.Decoder(code50, abi.default_cfg(a["method"], a["max_iter"]), 0, a["groups"])
    out, st = d.decode(fix, a["groups"])
    cnt = d.count_errors(out, None, a["groups"])
    d.close()
    assert cnt == [960, a["survey_frame_errors"], a["survey_bit_errors"], a["oracle_counters"][3]]
    assert hashlib.sha256(out.tobytes()).hexdigest() == a["oracle_sha256"]
    assert (int(st[:, 0].sum()), int(st[:, 1].sum())) == (a["oracle_sum_I"], a["oracle_sum_J"])


@pytest.mark.parametrize("method,eb_n0", [(2, 3.6), (5, 3.55), (1, 3.7)])
def test_large_batch_device_pointers(abi, code50, method, eb_n0):
    """96 groups (3072 codewords, more than one wave of workgroups) through lnsfaid_decode_device /
    lnsfaid_count_errors_device with torch-owned device memory, against the threaded oracle."""
    import torch
    ng, N = 96, code50.N
    cfg = abi.default_cfg(method, 10)
    fix = oa.synth_llr(ng, N, eb_n0, seed=7 + method)
    ref, ref_st = oa.decode_mt(code50, cfg, fix, ng)
    ref_cnt = oa.Oracle(code50, cfg).count_errors(ref, None, ng)
    d_fix = torch.from_numpy(fix).cuda()
    d_out = torch.empty(ng * 32 * N, dtype=torch.int8, device="cuda")
    d_st = torch.zeros((ng, 2), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    d = abi.Decoder(code50, cfg, 0, 128)  # context larger than the batch
    for _ in range(2):  # a context is reusable: same answer on the second call
        d.decode_device(d_fix.data_ptr(), ng, d_out.data_ptr(), d_st.data_ptr())
        cnt = d.count_errors_device(d_out.data_ptr(), None, ng)
        assert np.array_equal(d_out.cpu().numpy(), ref)
        assert np.array_equal(d_st.cpu().numpy(), ref_st)
        assert cnt == ref_cnt
    ms, launches = d.kernel_time()
    assert ms > 0 and launches >= 2
    d.close()


@pytest.mark.parametrize("off_in,off_out", [(1, 0), (0, 4), (3, 9)])
def test_unaligned_device_buffers(abi, code50, off_in, off_out):
    """lnsfaid_decode_device takes any alignment: a fixInput that is not dword aligned goes through the byte-load staging, a
    decodedBits that is not 16-byte aligned through dword stores; same output as with aligned buffers."""
    import torch
    ng, N = 6, code50.N
    cfg = abi.default_cfg(2, 10)
    fix = oa.synth_llr(ng, N, 3.6, seed=31)
    ref, ref_st = oa.decode_mt(code50, cfg, fix, ng)
    raw_in = torch.zeros(fix.size + 64, dtype=torch.int8, device="cuda")
    raw_in[off_in:off_in + fix.size] = torch.from_numpy(fix.reshape(-1)).cuda()
    raw_out = torch.zeros(ng * 32 * N + 64, dtype=torch.int8, device="cuda")
    d_st = torch.zeros((ng, 2), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert (raw_in.data_ptr() + off_in) % 4 == off_in % 4 and (raw_out.data_ptr() + off_out) % 16 == off_out % 16
    d = abi.Decoder(code50, cfg, 0, ng)
    d.decode_device(raw_in.data_ptr() + off_in, ng, raw_out.data_ptr() + off_out, d_st.data_ptr())
    got = raw_out[off_out:off_out + ng * 32 * N].cpu().numpy().reshape(ref.shape)
    assert np.array_equal(got, ref)
    assert np.array_equal(d_st.cpu().numpy(), ref_st)
    assert int(raw_out[:off_out].abs().sum()) == 0 and int(raw_out[off_out + ng * 32 * N:].abs().sum()) == 0  # nothing outside
    d.close()


def test_baseline_size_batch_properties(abi, code50):
    """BASELINE.json configs[1] size (2048 groups = 65 536 codewords), checked by properties that need no
    oracle run: noiseless codewords are fixed points (zero and the reference's known codeword, mixed per lane),
    the error counters agree with a host recount, decoding is deterministic, and a sample of groups matches
    the oracle."""
    ng, N, K = 2048, code50.N, code50.K
    cw = np.unpackbits(np.fromfile(os.path.join(GOLD, "codeword_50gpon.bin"), dtype=np.uint8))[:N].astype(np.int8)
    cfg = abi.default_cfg(2, 10)
    d = abi.Decoder(code50, cfg, 0, ng)
    # (1) fixed points, with a few noisy groups mixed in so that groups stop at different points
    frames = np.zeros((ng, 32, N), dtype=np.int8)
    frames[:, 1::2] = cw
    llr = np.where(frames > 0, 7, -7).astype(np.int8)
    fix = np.concatenate([llr[:, :, :K].reshape(ng, -1), llr[:, :, K:].reshape(ng, -1)], axis=1)
    noisy = oa.synth_llr(8, N, 3.5, seed=5).reshape(8, -1)
    fix[::256] = noisy
    fix = np.ascontiguousarray(fix.reshape(-1))
    out, st = d.decode(fix, ng)
    out3 = out.reshape(ng, 32, N)
    clean = np.ones(ng, dtype=bool)
    clean[::256] = False
    assert np.array_equal(out3[clean], frames[clean])
    assert (st[clean] == [1, 0]).all()
    ref, ref_st = oa.decode_mt(code50, cfg, np.ascontiguousarray(noisy.reshape(-1)), 8)
    assert np.array_equal(out3[::256].reshape(-1), ref) and np.array_equal(st[::256], ref_st)
    # (2) counters == host recount of the same bits; (3) determinism
    fix2 = oa.synth_llr(ng, N, 3.6, seed=11)
    out2, st2 = d.decode(fix2, ng)
    cnt = d.count_errors(out2, None, ng)
    err = out2.reshape(ng * 32, N)[:, :K].astype(np.int64).sum(axis=1)
    assert cnt == [ng * 32, int((err > 0).sum()), int(err.sum()), int(((err > 0) & (err < 3)).sum())]
    out2b, st2b = d.decode(fix2, ng)
    assert np.array_equal(out2, out2b) and np.array_equal(st2, st2b)
    # (4) every decoded frame the decoder reports as converged satisfies all checks (syndrome of the output)
    pos = np.ctypeslib.as_array(code50.pos_vn).astype(np.int64)
    row_deg = np.repeat(np.array(list(code50.deg)), np.array(list(code50.deg_rows)))
    starts = np.concatenate([[0], np.cumsum(row_deg)[:-1]])
    early = np.nonzero(st2[:, 0] < 10)[0][:64]  # groups that stopped early: all 32 lanes were clean
    assert early.size > 0
    sample = out2.reshape(ng, 32, N)[early].reshape(-1, N)
    synd = np.add.reduceat(sample[:, pos].astype(np.int64), starts, axis=1) & 1
    assert not synd.any()
    # (5) every one of the 65 536 frames against the vectorised CPU port (itself pinned to the oracle by
    # tests/test_oracle.py::test_avx2_port_equals_oracle)
    ref_all, ref_all_st = oa.decode_mt(code50, cfg, fix2, ng, kind="avx2")
    assert np.array_equal(out2, ref_all) and np.array_equal(st2, ref_all_st)
    # sample parity with the oracle at this size
    pick = [0, 777, 2047]
    sub = np.concatenate([fix2.reshape(ng, -1)[g] for g in pick])
    ref2, ref2_st = oa.decode_mt(code50, cfg, sub, len(pick))
    assert np.array_equal(np.concatenate([out2.reshape(ng, -1)[g] for g in pick]), ref2)
    assert np.array_equal(st2[pick], ref2_st)
    d.close()


def test_edge_cases(abi, lib, code50):
    cfg = abi.default_cfg(2, 10)
    d = abi.Decoder(code50, cfg, 0, 2)
    assert lib.lnsfaid_decode(d.ctx, None, 0, None, None) == 0                      # empty batch
    fix = oa.synth_llr(3, code50.N, 3.6, seed=1)
    out = np.empty_like(fix)
    assert lib.lnsfaid_decode(d.ctx, fix.ctypes.data, 3, out.ctypes.data, None) == -1   # more groups than the context holds
    assert lib.lnsfaid_decode(d.ctx, None, 1, None, None) == -1
    # all-erased input (every LLR 0): hard decisions are 0 everywhere, the group is clean at once
    zeros = np.zeros(32 * code50.N, dtype=np.int8)
    o, st = d.decode(zeros, 1)
    assert not o.any() and st.tolist() == [[0, 0]]
    # saturated, conflicting input (+7 everywhere = all-ones word, not a codeword): runs to the caps like the oracle
    sevens = np.full(32 * code50.N, 7, dtype=np.int8)
    o, st = d.decode(sevens, 1)
    ro, rst = oa.Oracle(code50, cfg).decode(sevens, 1)
    assert np.array_equal(o, ro) and np.array_equal(st, rst)
    # set_cfg switches iteration cap and factors without a new context
    cfg6 = abi.default_cfg(2, 6)
    d.set_cfg(cfg6)
    fx = oa.ReferenceChannel(code50, 107, 13.0).groups(3.6, 2)
    o, st = d.decode(fx, 2)
    ro, rst = oa.Oracle(code50, cfg6).decode(fx, 2)
    assert np.array_equal(o, ro) and np.array_equal(st, rst)
    d.close()


def test_oms_factors_and_nondefault_tables(abi, code50):
    """Profile.txt Factor_1/Factor_2 variants (OMS) and the reference's alternative LUT sets
    (FAID32, reference CDecoder_FAID.cpp:51-88) go through the same kernels."""
    fx = oa.ReferenceChannel(code50, 109, 13.0).groups(3.5, 2)
    cfg = abi.default_cfg(1, 10)
    cfg.factor_1, cfg.factor_2 = 2, 5
    d = abi.Decoder(code50, cfg, 0, 2)
    o, st = d.decode(fx, 2)
    ro, rst = oa.Oracle(code50, cfg).decode(fx, 2)
    assert np.array_equal(o, ro) and np.array_equal(st, rst)
    d.close()
    faid32 = [[0, 1, 1, 2, 3, 3, 3, 3], [0, 1, 1, 2, 3, 3, 3, 3], [0, 1, 1, 2, 4, 4, 4, 4],
              [1, 1, 1, 1, 4, 4, 4, 4], [1, 1, 1, 1, 5, 5, 5, 5], [1, 1, 1, 1, 6, 6, 6, 6]]
    cfg = abi.default_cfg(2, 10)
    for it in range(6):
        for w in range(4):
            for a in range(8):
                cfg.v2c_map[it][w][a] = faid32[it][a] + (1 if (w == 1 and a == 7 and it == 5) else 0)  # weight-6 row differs
    d = abi.Decoder(code50, cfg, 0, 2)
    o, st = d.decode(fx, 2)
    ro, rst = oa.Oracle(code50, cfg).decode(fx, 2)
    assert np.array_equal(o, ro) and np.array_equal(st, rst)
    d.close()


@pytest.mark.parametrize("mod_type,method,scale,eb_n0,extra,interleave",
                         [(2, 2, 13.0, 3.5, [], 1), (4, 5, 12.5, 8.1, [], 1), (2, 2, 13.0, 3.5, ["--device-frontend"], 1),
                          (6, 2, 12.5, 12.6, [], 3), (6, 2, 12.5, 12.6, ["--device-frontend"], 3)],
                         ids=["qpsk_faid", "16qam_2b1c", "qpsk_faid_device_frontend", "64qam_il3", "64qam_il3_device_frontend"])
def test_host_driver_sweep_point_matches_oracle(abi, code50, tmp_path, mod_type, method, scale, eb_n0, extra, interleave):
    """The CLDPC/CSimulate-shaped C++ driver (host/lnsfaid_sim): 4 streams (reference threads 0..3, seeds
    101, 103, 107, 109), one round of 50 calls at one Eb/N0 point; counters against the oracle fed by the restated
    channel with the same seeds.  Covers Profile.txt parsing, the DecodeMethod switch and the host front-end."""
    exe = os.path.join(oa.PKG_DIR, "host", "lnsfaid_sim")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(oa.PKG_DIR, "host")])
    prof = open(os.path.join(oa.PKG_DIR, "host", "Profile.txt")).read()
    prof = prof.replace("StartSNR: 3.3", "StartSNR: %g" % eb_n0).replace("EndSNR: 3.85", "EndSNR: %g" % (eb_n0 + 0.05))
    prof = prof.replace("DecodeMethod: 2", "DecodeMethod: %d" % method).replace("modType: 2", "modType: %d" % mod_type)
    prof = prof.replace("scale: 13", "scale: %g" % scale).replace("InterleaveModType: 1", "InterleaveModType: %d" % interleave)
    assert "InterleaveModType: %d" % interleave in prof
    (tmp_path / "Profile.txt").write_text(prof)
    res = subprocess.run([exe, "--streams", "4", "--gpus", "1", "--max-rounds", "1"] + extra, cwd=tmp_path, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    row = [l for l in res.stdout.splitlines() if re.match(r"\s*%g\s" % eb_n0, l)][-1].split()
    got = [int(row[1]), int(row[2]), int(row[3]), int(row[6])]
    cfg = abi.default_cfg(method, 10)
    want = [0, 0, 0, 0]
    for s, seed in enumerate([101, 103, 107, 109]):
        fix = oa.ReferenceChannel(code50, seed, scale, mod_type=mod_type, interleave=interleave).groups(eb_n0, 50)
        dec, _ = oa.decode_mt(code50, cfg, fix, 50, kind="avx2")
        c = oa.Oracle(code50, cfg).count_errors(dec, None, 50)
        want = [w + x for w, x in zip(want, c)]
    assert got == want, (got, want, res.stdout)
    assert (tmp_path / "Result.txt").exists() and (tmp_path / "Temp.txt").exists()


@pytest.mark.parametrize("extra", [[], ["--device-frontend"]], ids=["host_frontend", "device_frontend"])
def test_host_driver_result_files(abi, code50, tmp_path, extra):
    """N4: iterCount.txt (BF-iteration histogram of DecodeMethod 3 / 4, reference CSimulate.cpp:146-178) and the
    collect-flag dumps of CalculateErrors (errorindex / errorfloat / errordecode.txt, reference CLDPC.cpp:4877-4983),
    forced on from the first call with --collect (the reference switches them on once FER < 1e-5)."""
    exe = os.path.join(oa.PKG_DIR, "host", "lnsfaid_sim")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(oa.PKG_DIR, "host")])
    eb_n0, method = 3.45, 4
    prof = open(os.path.join(oa.PKG_DIR, "host", "Profile.txt")).read()
    prof = prof.replace("StartSNR: 3.3", "StartSNR: %g" % eb_n0).replace("EndSNR: 3.85", "EndSNR: %g" % (eb_n0 + 0.05))
    prof = prof.replace("DecodeMethod: 2", "DecodeMethod: %d" % method)
    (tmp_path / "Profile.txt").write_text(prof)
    res = subprocess.run([exe, "--streams", "2", "--gpus", "1", "--max-rounds", "1", "--collect"] + extra, cwd=tmp_path,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    cfg = abi.default_cfg(method, 10)
    hist, records = {}, []
    for seed in [101, 103]:
        fix = oa.ReferenceChannel(code50, seed, 13.0).groups(eb_n0, 50)
        dec, st = oa.decode_mt(code50, cfg, fix, 50, kind="avx2")
        for j in st[:, 1]:
            if j:
                hist[int(j)] = hist.get(int(j), 0) + 1
        d = dec.reshape(50, 32, code50.N)
        for g in range(50):
            for i in range(32):
                bad = np.nonzero(d[g, i, :code50.K])[0]
                if bad.size:
                    records.append((seed, g, i, bad, np.nonzero(d[g, i, code50.K:])[0] + code50.K, d[g, i]))
    lines = (tmp_path / "iterCount.txt").read_text().splitlines()
    assert lines[0].startswith("Eb/N0:")
    # demod.txt (reference main.cpp:75-83, :224-227): header + one row per Eb/N0 point; its counters are never incremented in the
    # reference either (CSimulate.cpp:129 is commented out), so the rates are zeros
    demod = (tmp_path / "demod.txt").read_text().splitlines()
    assert demod[0].split() == ["Eb/N0", "ModFER", "ModBER", "ModSER"] and len(demod) == 2
    assert abs(float(demod[1].split()[0]) - eb_n0) < 1e-6 and [float(x) for x in demod[1].split()[1:]] == [0.0, 0.0, 0.0]
    got = {}
    for l in lines[1:]:
        k, v = l.split(":")
        got[int(k)] = got.get(int(k), 0) + int(v)
    assert got == hist
    if extra:  # frames stay on the device: headers only, and a warning
        assert "no error dumps" in res.stderr
        assert len((tmp_path / "errorindex.txt").read_text().splitlines()) == 1
        return
    idx = (tmp_path / "errorindex.txt").read_text().splitlines()[1:]
    assert len(idx) == 7 * len(records)
    # the driver walks call by call with both streams inside: order records by (call, stream, frame)
    records.sort(key=lambda r: (r[1], [101, 103].index(r[0]), r[2]))
    for n, (seed, g, i, bad, badc, bits) in enumerate(records):
        rec = idx[7 * n:7 * n + 7]
        assert rec[0] == "ErrorFrame: %d" % i and rec[1] == "ErrorBit Num: %d" % bad.size
        assert [int(x) for x in rec[2].split(":")[1].split()] == (bad // 256 + 1).tolist()
        assert [int(x) for x in rec[3].split(":")[1].split()] == (bad % 256).tolist()
        assert rec[4] == "Errorcheck Num: %d" % badc.size
        assert [int(x) for x in rec[5].split(":")[1].split()] == (badc // 256 + 1).tolist()
    dec_lines = [l for l in (tmp_path / "errordecode.txt").read_text().splitlines() if l.startswith("Decodedbits=[")]
    assert len(dec_lines) == len(records)
    assert [int(x) for x in dec_lines[0][len("Decodedbits=["):-2].split()] == records[0][5].tolist()
    fl = [l for l in (tmp_path / "errorfloat.txt").read_text().splitlines() if l.startswith("ErrorChar=[")]
    assert len(fl) == len(records)
    seed, g, i = records[0][:3]
    fix = oa.ReferenceChannel(code50, seed, 13.0).groups(eb_n0, 50).reshape(50, -1)[g]
    want = np.concatenate([fix[i * code50.K:(i + 1) * code50.K], fix[32 * code50.K + i * code50.M:32 * code50.K + (i + 1) * code50.M]])
    assert [int(x) for x in fl[0][len("ErrorChar=["):-2].split()] == want.tolist()


@pytest.mark.parametrize("extra", [[], ["--device-frontend"]], ids=["host_frontend", "device_frontend"])
def test_host_driver_with_encoder(abi, code50, encoder, tmp_path, extra):
    """N2: `lnsfaid_sim --encode` sends random information bits (libc rand() % 2, reference CLDPC.cpp:60-66) through the
    systematic encoder the driver derives from the code table (host/CEncoder.cpp; the reference's GenMatrix is not shipped).
    Rebuilt here with glibc's rand(), the test encoder (tests/gf2_encoder.py) and the restated channel: the counters
    must match the oracle's, and every sent frame must be a codeword."""
    import ctypes
    exe = os.path.join(oa.PKG_DIR, "host", "lnsfaid_sim")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(oa.PKG_DIR, "host")])
    eb_n0 = 3.55
    prof = open(os.path.join(oa.PKG_DIR, "host", "Profile.txt")).read()
    prof = prof.replace("StartSNR: 3.3", "StartSNR: %g" % eb_n0).replace("EndSNR: 3.85", "EndSNR: %g" % (eb_n0 + 0.05))
    (tmp_path / "Profile.txt").write_text(prof)
    res = subprocess.run([exe, "--streams", "2", "--gpus", "1", "--max-rounds", "1", "--encode", "--collect"] + extra, cwd=tmp_path,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    row = [l for l in res.stdout.splitlines() if re.match(r"\s*%g\s" % eb_n0, l)][-1].split()
    got = [int(row[1]), int(row[2]), int(row[3]), int(row[6])]
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)  # a fresh process starts from seed 1; the driver never calls srand (nor does the reference)
    K, N = code50.K, code50.N
    info = np.array([libc.rand() % 2 for _ in range(2 * 32 * K)], dtype=np.uint8).reshape(2, 32, K)
    cfg = abi.default_cfg(2, 10)
    want = [0, 0, 0, 0]
    sent = []
    for s, seed in enumerate([101, 103]):
        frames = encoder.encode(info[s])
        sent.append(frames)
        fix = oa.ReferenceChannel(code50, seed, 13.0).groups(eb_n0, 50, frames=frames)
        dec, _ = oa.decode_mt(code50, cfg, fix, 50, kind="avx2")
        c = oa.Oracle(code50, cfg).count_errors(dec, np.ascontiguousarray(np.tile(info[s].reshape(-1).astype(np.int8), 50)), 50)
        want = [w + x for w, x in zip(want, c)]
    assert got == want, (got, want, res.stdout)
    assert want[1] > 0  # the point has frame errors, so the dumps below exist
    if extra:  # lnsfaid_frontend_set_frames: the GPU sent the same frames (the counters above prove it); no dumps in this mode
        return
    # the frames the driver sent (outputbits of the dump) are the encoder's, i.e. codewords of H
    ob = [l for l in (tmp_path / "errordecode.txt").read_text().splitlines() if l.startswith("outputbits=[")]
    first = np.array([int(x) for x in ob[0][len("outputbits=["):-2].split()], dtype=np.int8)
    assert any(np.array_equal(first, f) for f in np.concatenate(sent))


@pytest.mark.parametrize("extra", [[], ["--device-frontend"]], ids=["host_frontend", "device_frontend"])
def test_host_driver_resume_from_temp_txt(abi, code50, tmp_path, extra):
    """The lastSeed table of Temp.txt (reference main.cpp:200-207) fed back with --resume (the reference compiles it in
    under CONTINUE_SEED, CChannel.cpp:4-41): the resumed round continues the noise streams exactly where the first round
    stopped, on the host and on the device front-end."""
    exe = os.path.join(oa.PKG_DIR, "host", "lnsfaid_sim")
    eb_n0 = 3.5
    prof = open(os.path.join(oa.PKG_DIR, "host", "Profile.txt")).read()
    prof = prof.replace("StartSNR: 3.3", "StartSNR: %g" % eb_n0).replace("EndSNR: 3.85", "EndSNR: %g" % (eb_n0 + 0.05))
    (tmp_path / "Profile.txt").write_text(prof)
    base = [exe, "--streams", "2", "--gpus", "1", "--max-rounds", "1"] + extra
    res1 = subprocess.run(base, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res1.returncode == 0, res1.stderr
    table = re.findall(r"\{(\d+),(\d+),(\d+)\}", (tmp_path / "Temp.txt").read_text())
    assert len(table) == 2
    os.rename(tmp_path / "Temp.txt", tmp_path / "Saved.txt")
    res2 = subprocess.run(base + ["--resume", "Saved.txt"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res2.returncode == 0, res2.stderr
    row = [l for l in res2.stdout.splitlines() if re.match(r"\s*%g\s" % eb_n0, l)][-1].split()
    got = [int(row[1]), int(row[2]), int(row[3]), int(row[6])]
    cfg = abi.default_cfg(2, 10)
    want = [0, 0, 0, 0]
    for s, seed in enumerate([101, 103]):
        ch = oa.ReferenceChannel(code50, seed, 13.0)
        ch.groups(eb_n0, 50)  # the first round's 50 calls
        assert (ch.fe.IX, ch.fe.IY, ch.fe.IZ) == tuple(int(x) for x in table[s])  # the table is the generator state
        fix = ch.groups(eb_n0, 50)  # the resumed round
        dec, _ = oa.decode_mt(code50, cfg, fix, 50, kind="avx2")
        want = [w + x for w, x in zip(want, oa.Oracle(code50, cfg).count_errors(dec, None, 50))]
    assert got == want, (got, want, res2.stdout)


def _derived_code(abi, lib, drop_cols, from_block_row, keep_edges=None):
    """A second quasi-cyclic code for the generic code paths: the 50G-PON table with the circulants of the
    block columns `drop_cols` removed from block rows >= from_block_row (degree 23 -> 23 - len(drop_cols)); `keep_edges`
    {block row: n} keeps only the first n circulants of a block row."""
    import ctypes as C
    base = abi.Code50GPON(lib)
    pos = np.ctypeslib.as_array(base.pos_vn)
    out, e = [], 0
    degs = []
    for r in range(3072):
        d = 22 if 256 <= r < 512 else 23
        row = pos[e:e + d]
        e += d
        if r // 256 >= from_block_row:
            row = row[~np.isin(row // 256, drop_cols)]
        if keep_edges and r // 256 in keep_edges:
            row = row[:keep_edges[r // 256]]
        out.append(row)
        degs.append(len(row))
    classes, rows = [], []
    for d in degs:
        if classes and classes[-1] == d:
            rows[-1] += 1
        else:
            classes.append(d)
            rows.append(1)
    flat = np.concatenate(out).astype(np.uint16)

    class Derived:
        pass
    dc = Derived()
    dc.pos_vn = (C.c_uint16 * flat.size)(*flat.tolist())
    dc.deg = (C.c_int32 * len(classes))(*classes)
    dc.deg_rows = (C.c_int32 * len(rows))(*rows)
    dc.code = abi.Code()
    dc.code.n_var, dc.code.n_check, dc.code.n_edges, dc.code.z = 17664, 3072, int(flat.size), 256
    dc.code.puncture_tail, dc.code.nb_degres = 384, len(classes)
    dc.code.deg, dc.code.deg_rows, dc.code.pos_vn = dc.deg, dc.deg_rows, dc.pos_vn
    dc.N, dc.M, dc.K = 17664, 3072, 17664 - 3072
    return dc


@pytest.mark.parametrize("method", [2, 1, 5])
def test_other_code_with_runtime_row_degree(abi, lib, method):
    """Row degrees 23 / 22 / 21: the degree-21 layers take the kernel's run-time-degree path."""
    dc = _derived_code(abi, lib, [67, 68], 2)
    assert list(dc.deg) == [23, 22, 21]
    cfg = abi.default_cfg(method, 10)
    fix = oa.synth_llr(3, dc.N, 3.9, seed=21 + method)  # a weaker code: decode a little above the waterfall
    ref, rst = oa.decode_mt(dc, cfg, fix, 3)
    d = abi.Decoder(dc, cfg, 0, 3)
    out, st = d.decode(fix, 3)
    assert np.array_equal(out, ref) and np.array_equal(st, rst)
    d.select_waves(2)  # the experimental two-waves-per-codeword kernel: its run-time-degree layer step, edges dealt by parity
    out, st = d.decode(fix, 3)
    d.close()
    assert np.array_equal(out, ref) and np.array_equal(st, rst)


@pytest.mark.parametrize("method", [2, 1])
def test_other_code_with_low_row_degrees(abi, lib, method):
    """Block rows of degree 12, 7 and 3 next to the full ones: the run-time-degree layer step with no edge in 16..23 / 8..15 (the
    arg-min index takes its bits 4 and 3 from those ranges), and rows shorter than one 8-edge sign word."""
    dc = _derived_code(abi, lib, [], 12, keep_edges={9: 12, 10: 7, 11: 3})
    assert sorted(set(dc.deg)) == [3, 7, 12, 22, 23]
    cfg = abi.default_cfg(method, 10)
    fix = oa.synth_llr(3, dc.N, 4.6, seed=77 + method)
    ref, rst = oa.decode_mt(dc, cfg, fix, 3)
    d = abi.Decoder(dc, cfg, 0, 3)
    assert d.rows_per_lane() == 4
    out, st = d.decode(fix, 3)
    assert np.array_equal(out, ref) and np.array_equal(st, rst)
    d.select_waves(2)  # rows of 3 edges: one wave of the pair has a single edge of them, index accumulators of either stay untouched
    out, st = d.decode(fix, 3)
    d.close()
    assert np.array_equal(out, ref) and np.array_equal(st, rst)
    assert (ref != 0).any()  # (the weakened code leaves errors: the comparison is not all-zero against all-zero)


@pytest.mark.parametrize("method,alpha,W,L0,delta", [(2, 2, 3, 50, 1), (2, 1, 6, 3, 1), (5, 0, 3, 2, 2), (5, 3, 3, 100, 1)])
def test_bit_flipping_parameter_variants(abi, code50, method, alpha, W, L0, delta):
    """Non-shipped DTBF / 2B1C constants: alpha outside {0, 1} or a column weight other than 3 take the
    per-variable-node flip path; alpha = 0 stays on the bit-sliced one."""
    cfg = abi.default_cfg(method, 4)  # few layered iterations so that the bit-flipping stage has work to do
    cfg.bf_alpha, cfg.regular_col_weight, cfg.bf_L0, cfg.bf_delta = alpha, W, L0, delta
    fix = oa.ReferenceChannel(code50, 113, 13.0).groups(3.7, 2)
    ref, rst = oa.Oracle(code50, cfg).decode(fix, 2)
    d = abi.Decoder(code50, cfg, 0, 2)
    out, st = d.decode(fix, 2)
    d.close()
    assert rst[:, 1].max() > 0
    assert np.array_equal(out, ref) and np.array_equal(st, rst)


@pytest.mark.parametrize("eb_n0", [8.1, 8.6])
def test_hybrid_2b1c_on_16qam(abi, code50, eb_n0):
    """BASELINE.json configs[4]: DecodeMethod 5 (hybrid-precision FAID + 2B1C), 16-QAM, scale 12.5 (reference
    README.md:20), LLRs from the restated reference mapper / max-log demapper."""
    cfg = abi.default_cfg(5, 10)
    fix = oa.ReferenceChannel(code50, 101, 12.5, mod_type=4).groups(eb_n0, 6)
    ref, rst = oa.decode_mt(code50, cfg, fix, 6)
    d = abi.Decoder(code50, cfg, 0, 6)
    out, st = d.decode(fix, 6)
    d.close()
    assert np.array_equal(out, ref) and np.array_equal(st, rst)


@pytest.mark.parametrize("method,eb_n0", [(2, 3.5), (2, 3.65), (1, 3.6), (5, 3.55)])
def test_random_codewords(abi, code50, encoder, method, eb_n0):
    """64 groups of per-frame different random codewords (systematic encoder of tests/gf2_encoder.py; the reference
    can only send one fixed word because its GenMatrix data is not shipped): decoder parity on signs of both
    polarities, and the counter pass against the transmitted information bits."""
    import gf2_encoder as ge
    enc = encoder
    ng = 64
    rng = np.random.default_rng(100 + method)
    info = rng.integers(0, 2, size=(ng * 32, code50.K), dtype=np.uint8)
    cw = enc.encode(info)
    fix = ge.qpsk_llr(cw, eb_n0, seed=method)
    cfg = abi.default_cfg(method, 10)
    ref, rst = oa.decode_mt(code50, cfg, fix, ng, kind="avx2")
    inp = np.ascontiguousarray(info.astype(np.int8).reshape(-1))
    d = abi.Decoder(code50, cfg, 0, ng)
    out, st = d.decode(fix, ng)
    cnt = d.count_errors(out, inp, ng)
    d.close()
    assert np.array_equal(out, ref) and np.array_equal(st, rst)
    assert cnt == oa.Oracle(code50, cfg).count_errors(ref, inp, ng)
    assert cnt[1] < ng * 32  # not everything fails: the words really are codewords


@pytest.mark.parametrize("eb_n0", [3.2, 3.6, 4.0])
@pytest.mark.parametrize("method", [2, 5, 1, 0, -1, 4, 3])
def test_soak_against_cpu_port(abi, code50, method, eb_n0):
    """16 384 frames per case (512 groups, eight dispatch rounds of workgroups) against the vectorised CPU port, every
    frame and every per-group iteration count: the place where rare message patterns (ties, zero messages on the argmin
    edge, saturated rows) turn up.  (tools/gpu_soak.py is the long version: 65 536 frames per case, every kernel variant.)"""
    ng = 512
    uni = method == -1  # DecodeMethod 0 with one normalisation factor: the kernel's patch path
    method = 0 if uni else method
    cfg = abi.default_cfg(method, 10)
    if method == 0:
        cfg.factor_1, cfg.factor_2 = (24, 24) if uni else (24, 26)  # the shipped Profile.txt factors are OMS offsets
    fix = oa.synth_llr(ng, code50.N, eb_n0, seed=1000 + 17 * method + int(eb_n0 * 10))
    d = abi.Decoder(code50, cfg, 0, ng)
    out, st = d.decode(fix, ng)
    d.close()
    ref, ref_st = oa.decode_mt(code50, cfg, fix, ng, kind="avx2")
    bad = np.nonzero((out != ref).reshape(ng * 32, code50.N).any(axis=1))[0]
    assert bad.size == 0, "frames differ from the CPU port: %s" % bad[:16].tolist()
    assert np.array_equal(st, ref_st)


def test_pinned_host_buffers_take_the_pipelined_path(abi, lib, code50):
    """lnsfaid_decode with page-locked host buffers (lnsfaid_host_register) cuts the batch into pieces of whole groups and
    overlaps copies and decode; results and per-group statistics must equal the single-piece path, also when the last
    piece is ragged."""
    ng = 200  # pieces of 64 groups: 64 + 64 + 64 + 8
    cfg = abi.default_cfg(2, 10)
    fix = oa.synth_llr(ng, code50.N, 3.6, seed=77)
    d = abi.Decoder(code50, cfg, 0, ng)
    ref, ref_st = d.decode(fix, ng)  # pageable numpy buffers
    src, dst = fix.copy(), np.zeros_like(fix)
    st = np.zeros((ng, 2), dtype=np.int32)
    assert lib.lnsfaid_host_register(src.ctypes.data, src.size) == 0 and lib.lnsfaid_host_register(dst.ctypes.data, dst.size) == 0
    try:
        assert lib.lnsfaid_decode(d.ctx, src.ctypes.data, ng, dst.ctypes.data, st.ctypes.data) == 0
    finally:
        assert lib.lnsfaid_host_unregister(src.ctypes.data) == 0 and lib.lnsfaid_host_unregister(dst.ctypes.data) == 0
    d.close()
    assert np.array_equal(dst, ref) and np.array_equal(st, ref_st)
    assert lib.lnsfaid_host_register(None, 16) != 0


@pytest.mark.parametrize("mode", [1, 2])
def test_ef_elimination_variants_of_decode_faid(abi, lib, code50, mode):
    """SURVEY.md 8(f) N3, last item: EF_ELIMINATION 1 (error-floor tables) and 2 (+ erasure of the V2C of weight-3 nodes with
    three unsatisfied checks, CDecoder_FAID.cpp:673-680) of Decode_FAID on the GPU against the oracle: channel batches and nearly
    clean frames with a few confident errors; hard decisions and (I, J) per group.  No reference output exists for these
    compile-time variants (parity unpinned)."""
    rng = np.random.default_rng(70 + mode)
    N, K = code50.N, code50.K
    ng = 4
    llr = np.full((ng * 32, N), -3, dtype=np.int16) + rng.integers(-2, 3, size=(ng * 32, N))
    for l in range(ng * 32):
        nf = int(rng.integers(2, 24))
        llr[l, rng.integers(17 * 256, 67 * 256, size=nf)] = rng.integers(3, 8, size=nf)
    llr = np.clip(llr, -7, 7).astype(np.int8).reshape(ng, 32, N)
    synth = np.concatenate([np.concatenate([g[:, :K].reshape(-1), g[:, K:].reshape(-1)]) for g in llr])
    for max_iter, fix in ((10, oa.ReferenceChannel(code50, 137, 13.0).groups(3.55, ng)), (6, synth), (7, synth), (10, synth)):
        cfg = abi.default_cfg(2, max_iter)
        assert lib.lnsfaid_cfg_ef_elimination(cfg, mode) == 0
        ref, rst = oa.decode_mt(code50, cfg, fix, ng)
        d = abi.Decoder(code50, cfg, 0, ng)
        assert d.rows_per_lane() == 4
        with pytest.raises(RuntimeError):
            d.select_kernel(2)  # these variants exist in the four-rows-per-lane kernel only
        out, st = d.decode(fix, ng)
        assert np.array_equal(out, ref) and np.array_equal(st, rst)
        if mode == 1:  # the experimental two-waves-per-codeword kernel has the _ef tables too (not the erasure)
            d.select_waves(2)
            out, st = d.decode(fix, ng)
            assert d.kernel_waves() == 2 and np.array_equal(out, ref) and np.array_equal(st, rst)
        d.close()
    if mode == 2:  # the erasure changes what is decoded on the synthetic batch (else the test would prove nothing about it)
        cfg = abi.default_cfg(2, 6)
        assert lib.lnsfaid_cfg_ef_elimination(cfg, 2) == 0
        with_erasure, st2 = oa.decode_mt(code50, cfg, synth, ng)
        cfg.ef_elimination = 1
        without, st1 = oa.decode_mt(code50, cfg, synth, ng)
        assert not (np.array_equal(with_erasure, without) and np.array_equal(st1, st2))
    cfg = abi.default_cfg(2, 10)
    assert lib.lnsfaid_cfg_ef_elimination(cfg, 2) == 0
    cfg.v2c_map_ef[0][2][5] = 7  # error-floor tables must be uniform over the weight classes for these variants
    with pytest.raises(RuntimeError):
        abi.Decoder(code50, cfg, 0, 1)


def test_one_group_contexts_of_different_codes_and_changing_configurations(abi, lib, code50):
    """One-group contexts share a call combiner only with contexts of the SAME code (lnsfaid_capi.hip): a context of another code
    beside them decodes on its own, a context whose REGULAR_COL_WEIGHT is changed by lnsfaid_set_cfg (which changes the
    bit-flipping column list of its code) leaves the common path, and a configuration change between two calls of a member is
    picked up by its next batch.  All of it from concurrent threads, every result against the oracle / the CPU port."""
    import threading
    dc = _derived_code(abi, lib, [67, 68], 2)
    n_calls = 4
    plans = [  # (code, cfgs to cycle through)
        (code50, [abi.default_cfg(2, 10)]),
        (code50, [abi.default_cfg(2, 10), abi.default_cfg(2, 6)]),        # MaxIteration changes between calls
        (code50, [abi.default_cfg(5, 10)]),
        (code50, [abi.default_cfg(1, 10)]),
        (dc, [abi.default_cfg(2, 10)]),                                    # another code: never in a common batch
        (code50, [abi.default_cfg(2, 10), abi.default_cfg(2, 10)]),        # second cfg gets another column weight below
    ]
    plans[5][1][1].regular_col_weight = 6
    plans[5][1][1].bf_alpha = 1
    jobs = []
    for t, (code, cfgs) in enumerate(plans):
        fix = oa.synth_llr(n_calls, code.N, 3.9 if code is dc else 3.55, seed=500 + t).reshape(n_calls, -1)
        want = []
        for c in range(n_calls):
            cfg = cfgs[c % len(cfgs)]
            ref, rst = oa.decode_mt(code, cfg, np.ascontiguousarray(fix[c]), 1)
            want.append((ref, rst))
        jobs.append((code, cfgs, fix, want))
    errors = []
    start = threading.Barrier(len(jobs))

    def worker(t):
        code, cfgs, fix, want = jobs[t]
        try:
            dec = abi.Decoder(code, cfgs[0], device=0, max_groups=1)
            start.wait()
            for c in range(n_calls):
                if len(cfgs) > 1:
                    dec.set_cfg(cfgs[c % len(cfgs)])
                out, st = dec.decode(np.ascontiguousarray(fix[c]), 1)
                if not np.array_equal(out, want[c][0]) or not np.array_equal(st, want[c][1]):
                    errors.append((t, c, st.tolist(), want[c][1].tolist()))
            dec.close()
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not any(th.is_alive() for th in threads), "a decode call did not return"
    assert not errors, errors[:4]
