"""ctypes view of oracle/liblnsfaid_oracle.so (the CPU oracle: test infrastructure only)."""
import ctypes as C
import importlib.util
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liblnsfaid_oracle.so")
PKG_DIR = os.path.join(ROOT, "mod-interleaveavx_multithreads-faid_amd")


def load_pyabi():
    """The package directory name is not a Python identifier, so import its ctypes module by path."""
    import sys
    if "lnsfaid_pyabi" in sys.modules:  # bench.py has loaded it already: one instance per process (ctypes classes compare by identity)
        return sys.modules["lnsfaid_pyabi"]
    spec = importlib.util.spec_from_file_location("lnsfaid_pyabi", os.path.join(PKG_DIR, "pyabi.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["lnsfaid_pyabi"] = mod
    spec.loader.exec_module(mod)
    return mod


pyabi = load_pyabi()


class Frontend(C.Structure):
    _fields_ = [("IX", C.c_ulong), ("IY", C.c_ulong), ("IZ", C.c_ulong)]


_SYMS = {
    "lnsfaid_oracle_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(pyabi.Code), C.POINTER(pyabi.Cfg)]),
    "lnsfaid_oracle_destroy": (None, [C.c_void_p]),
    "lnsfaid_oracle_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "lnsfaid_oracle_count_errors": (C.c_int, [C.POINTER(pyabi.Code), C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "lnsfaid_cpu_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(pyabi.Code), C.POINTER(pyabi.Cfg)]),
    "lnsfaid_cpu_destroy": (None, [C.c_void_p]),
    "lnsfaid_cpu_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "lnsfaid_frontend_seed": (None, [C.POINTER(Frontend), C.c_int]),
    "lnsfaid_frontend_sigma": (C.c_float, [C.c_float, C.c_int, C.c_double]),
    "lnsfaid_frontend_qpsk_group": (None, [C.POINTER(Frontend), C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p]),
    "lnsfaid_frontend_qpsk_frames": (None, [C.POINTER(Frontend), C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p]),
    "lnsfaid_frontend_group": (C.c_int, [C.POINTER(Frontend), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "lnsfaid_frontend_qam16_group": (None, [C.POINTER(Frontend), C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR])
        _lib = pyabi.bind(C.CDLL(ORACLE_LIB), _SYMS)
    return _lib


class Oracle:
    """kind="oracle": the pinned scalar restatement; kind="avx2": the vectorised CPU port (same results, faster)."""

    def __init__(self, code50, cfg, kind="oracle"):
        self.lib = load()
        self.code50 = code50
        self.h = C.c_void_p()
        self._create, self._decode, self._destroy = {
            "oracle": (self.lib.lnsfaid_oracle_create, self.lib.lnsfaid_oracle_decode, self.lib.lnsfaid_oracle_destroy),
            "avx2": (self.lib.lnsfaid_cpu_create, self.lib.lnsfaid_cpu_decode, self.lib.lnsfaid_cpu_destroy),
        }[kind]
        rc = self._create(C.byref(self.h), C.byref(code50.code), C.byref(cfg))
        if rc != 0:
            raise RuntimeError("oracle create (%s) failed: %d" % (kind, rc))

    def decode(self, fix_input, n_groups):
        N = self.code50.N
        assert fix_input.dtype == np.int8 and fix_input.size == n_groups * 32 * N
        fix_input = np.ascontiguousarray(fix_input)
        out = np.empty(n_groups * 32 * N, dtype=np.int8)
        stats = np.zeros((n_groups, 2), dtype=np.int32)
        rc = self._decode(self.h, fix_input.ctypes.data, n_groups, out.ctypes.data, stats.ctypes.data)
        if rc != 0:
            raise RuntimeError("oracle decode failed: %d" % rc)
        return out, stats

    def count_errors(self, decoded, input_bits, n_groups):
        out = (C.c_uint64 * 4)()
        ip = input_bits.ctypes.data if input_bits is not None else None
        rc = self.lib.lnsfaid_oracle_count_errors(C.byref(self.code50.code), decoded.ctypes.data, ip, n_groups, out)
        if rc != 0:
            raise RuntimeError("lnsfaid_oracle_count_errors failed: %d" % rc)
        return list(out)

    def close(self):
        if self.h:
            self._destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReferenceChannel:
    """The reference's per-thread QPSK/AWGN front-end (oracle/frontend_oracle.c)."""

    RATE = 0.8444444  # m_Rate, reference CLDPC.cpp:4780

    def __init__(self, code50, seed=101, scale=13.0, mod_type=2, interleave=1):
        self.lib = load()
        self.code50 = code50
        self.fe = Frontend()
        self.scale = scale
        self.mod_type = mod_type  # 2 QPSK, 4 16-QAM, 6 64-QAM, 8 256-QAM (Profile.txt modType)
        self.interleave = interleave  # Profile.txt InterleaveModType
        self.lib.lnsfaid_frontend_seed(C.byref(self.fe), seed)

    def groups(self, eb_n0_db, n_groups, codeword=None, frames=None):
        """codeword: one [N] codeword sent in every frame (FakeEncoder); frames: [32, N] bits, 32 different frames sent in
        every group (the driver with a real encoder: encoded once per 50 calls, reference CSimulate.cpp:103-116)."""
        N, M = self.code50.N, self.code50.M
        sigma = self.lib.lnsfaid_frontend_sigma(eb_n0_db, self.mod_type, self.RATE)
        out = np.empty((n_groups, 32 * N), dtype=np.int8)
        if self.mod_type > 4 or self.interleave != 1:  # the general restatement (also covers 2 / 4, see test_oracle.py)
            bits, stride = None, 0
            if frames is not None:
                frames = np.ascontiguousarray(frames, dtype=np.int8).reshape(32, N)
                bits, stride = frames.ctypes.data, N
            elif codeword is not None:
                codeword = np.ascontiguousarray(codeword, dtype=np.int8)
                bits = codeword.ctypes.data
            for g in range(n_groups):
                rc = self.lib.lnsfaid_frontend_group(C.byref(self.fe), N, M, bits, stride, self.mod_type, self.interleave, sigma,
                                                     self.scale, out[g].ctypes.data)
                assert rc == 0
            return out.reshape(-1)
        gen = {2: self.lib.lnsfaid_frontend_qpsk_group, 4: self.lib.lnsfaid_frontend_qam16_group}[self.mod_type]
        if frames is not None:
            assert self.mod_type == 2
            frames = np.ascontiguousarray(frames, dtype=np.int8).reshape(32, N)
            for g in range(n_groups):
                self.lib.lnsfaid_frontend_qpsk_frames(C.byref(self.fe), N, M, frames.ctypes.data, sigma, self.scale, out[g].ctypes.data)
            return out.reshape(-1)
        cw = None
        if codeword is not None:
            codeword = np.ascontiguousarray(codeword, dtype=np.int8)
            cw = codeword.ctypes.data
        for g in range(n_groups):
            gen(C.byref(self.fe), N, M, cw, sigma, self.scale, out[g].ctypes.data)
        return out.reshape(-1)


def decode_mt(code50, cfg, fix_input, n_groups, threads=None, kind="oracle"):
    """Oracle over n_groups groups with a pool of host threads (one oracle instance per thread)."""
    import concurrent.futures
    threads = max(1, min(threads or (os.cpu_count() or 1), n_groups, 16))
    per = 32 * code50.N
    oracles = [Oracle(code50, cfg, kind) for _ in range(threads)]
    outs = [None] * n_groups

    def work(t):
        for g in range(t, n_groups, threads):
            outs[g] = oracles[t].decode(fix_input[g * per:(g + 1) * per], 1)

    with concurrent.futures.ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    return np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])


def synth_llr(n_groups, n_var, eb_n0, seed, scale=13.0, rate=ReferenceChannel.RATE):
    """iid QPSK/AWGN LLRs of the all-zero codeword, quantised like float2LimitChar_4bit (numpy generator)."""
    sigma = 1.0 / np.sqrt(rate * 2 * 10.0 ** (0.1 * eb_n0))
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n_groups * 32 * n_var, dtype=np.float32) * np.float32(sigma / np.sqrt(2.0)) - np.float32(0.707107)
    return np.clip(np.trunc(x * np.float32(scale)), -7, 7).astype(np.int8)
