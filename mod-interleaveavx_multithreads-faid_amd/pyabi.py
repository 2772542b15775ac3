"""ctypes view of the C ABI in include/lnsfaid.h.

This is plumbing for tests/, bench.py and __graft_entry__.py: it declares the structs and the
entry points of liblnsfaid.so (the HIP product library) one-to-one and adds no logic of its own.
The library is loaded from csrc/ next to this file; a missing library is a hard error (there is
no CPU fallback in the product path).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "liblnsfaid.so")

GROUP = 32
MSG_REGISTERS, MSG_HBM = 1, 2  # lnsfaid_select_message_store


class Code(C.Structure):
    _fields_ = [
        ("n_var", C.c_int32),
        ("n_check", C.c_int32),
        ("n_edges", C.c_int32),
        ("z", C.c_int32),
        ("puncture_tail", C.c_int32),
        ("nb_degres", C.c_int32),
        ("deg", C.POINTER(C.c_int32)),
        ("deg_rows", C.POINTER(C.c_int32)),
        ("pos_vn", C.POINTER(C.c_uint16)),
    ]


class Cfg(C.Structure):
    _fields_ = [
        ("decode_method", C.c_int32),
        ("max_iteration", C.c_int32),
        ("factor_1", C.c_int32),
        ("factor_2", C.c_int32),
        ("floor_err_count", C.c_int32),
        ("floor_iter_thresh", C.c_int32),
        ("ef_elimination", C.c_int32),
        ("max_bf_iter", C.c_int32),
        ("bf_L0", C.c_int32),
        ("bf_L1", C.c_int32),
        ("bf_alpha", C.c_int32),
        ("bf_delta", C.c_int32),
        ("regular_col_weight", C.c_int32),
        ("hard2_threshold", C.c_int32),
        ("bf_vote_cap", C.c_int32),
        ("v2c_map", C.c_int8 * 8 * 4 * 6),
        ("v2c_map_ef", C.c_int8 * 8 * 4 * 6),
    ]


class GroupStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("bf_iterations", C.c_int32)]


# every symbol include/lnsfaid.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "lnsfaid_code_50gpon": (C.c_int, [C.POINTER(Code), C.POINTER(C.c_uint16), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lnsfaid_cfg_default": (C.c_int, [C.POINTER(Cfg), C.c_int32, C.c_int32]),
    "lnsfaid_cfg_table_preset": (C.c_int, [C.POINTER(Cfg), C.c_int32]),
    "lnsfaid_cfg_ef_elimination": (C.c_int, [C.POINTER(Cfg), C.c_int32]),
    "lnsfaid_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Code), C.POINTER(Cfg), C.c_int32, C.c_size_t]),
    "lnsfaid_destroy": (None, [C.c_void_p]),
    "lnsfaid_set_cfg": (C.c_int, [C.c_void_p, C.POINTER(Cfg)]),
    "lnsfaid_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "lnsfaid_decode_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "lnsfaid_count_errors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "lnsfaid_count_errors_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "lnsfaid_frontend_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_size_t, C.c_int32,
                                          C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "lnsfaid_frontend_device_states": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_size_t, C.c_int32,
                                                 C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "lnsfaid_frontend_draws_per_group": (C.c_uint64, [C.c_void_p, C.c_int32]),
    "lnsfaid_frontend_set_interleave": (C.c_int, [C.c_void_p, C.c_int32]),
    "lnsfaid_frontend_set_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "lnsfaid_frontend_input_bits": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "lnsfaid_io_buffers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "lnsfaid_read_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "lnsfaid_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "lnsfaid_host_unregister": (C.c_int, [C.c_void_p]),
    "lnsfaid_comm_unique_id": (C.c_int, [C.c_void_p]),
    "lnsfaid_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "lnsfaid_comm_attach": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lnsfaid_comm_destroy": (C.c_int, [C.c_void_p]),
    "lnsfaid_allreduce_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "lnsfaid_select_kernel": (C.c_int, [C.c_void_p, C.c_int32]),
    "lnsfaid_kernel_rows_per_lane": (C.c_int, [C.c_void_p]),
    "lnsfaid_select_waves": (C.c_int, [C.c_void_p, C.c_int32]),
    "lnsfaid_frontend_set_exact": (C.c_int, [C.c_void_p, C.c_int32]),
    "lnsfaid_frontend_fastpath_bounds": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "lnsfaid_kernel_waves": (C.c_int, [C.c_void_p]),
    "lnsfaid_select_message_store": (C.c_int, [C.c_void_p, C.c_int32]),
    "lnsfaid_message_store": (C.c_int, [C.c_void_p]),
    "lnsfaid_kernel_residency": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lnsfaid_kernel_time": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int32]),
    "lnsfaid_stream": (C.c_void_p, [C.c_void_p]),
    "lnsfaid_strerror": (C.c_char_p, [C.c_int]),
    "lnsfaid_last_hip_error": (C.c_char_p, []),
    "lnsfaid_version": (C.c_char_p, []),
}


def bind(lib, symbols):
    for name, (res, args) in symbols.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def load():
    """Load liblnsfaid.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(no CPU fallback exists in the product path)" % LIB_PATH)
        _lib = bind(C.CDLL(LIB_PATH), SYMBOLS)
    return _lib


class Code50GPON:
    """The built-in 50G-PON code, expanded to the Constants_SSE.h table format, with its buffers kept alive."""

    def __init__(self, lib=None):
        lib = lib or load()
        self.pos_vn = (C.c_uint16 * 70400)()
        self.deg = (C.c_int32 * 3)()
        self.deg_rows = (C.c_int32 * 3)()
        self.code = Code()
        rc = lib.lnsfaid_code_50gpon(C.byref(self.code), self.pos_vn, self.deg, self.deg_rows)
        if rc != 0:
            raise RuntimeError("lnsfaid_code_50gpon failed: %d" % rc)

    @property
    def N(self):
        return self.code.n_var

    @property
    def M(self):
        return self.code.n_check

    @property
    def K(self):
        return self.code.n_var - self.code.n_check


def default_cfg(method, max_iter, lib=None):
    lib = lib or load()
    cfg = Cfg()
    rc = lib.lnsfaid_cfg_default(C.byref(cfg), method, max_iter)
    if rc != 0:
        raise ValueError("lnsfaid_cfg_default(%d, %d) failed: %d" % (method, max_iter, rc))
    return cfg


class Decoder:
    """Thin RAII wrapper over lnsfaid_create / lnsfaid_decode* / lnsfaid_destroy."""

    def __init__(self, code50, cfg, device=0, max_groups=64, lib=None):
        self.lib = lib or load()
        self.code50 = code50
        self.ctx = C.c_void_p()
        rc = self.lib.lnsfaid_create(C.byref(self.ctx), C.byref(code50.code), C.byref(cfg), device, max_groups)
        if rc != 0:
            raise RuntimeError("lnsfaid_create failed: %d (%s; hip: %s)" % (
                rc, self.lib.lnsfaid_strerror(rc).decode(), self.lib.lnsfaid_last_hip_error().decode()))

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %d (%s; hip: %s)" % (
                what, rc, self.lib.lnsfaid_strerror(rc).decode(), self.lib.lnsfaid_last_hip_error().decode()))

    def set_cfg(self, cfg):
        self._check(self.lib.lnsfaid_set_cfg(self.ctx, C.byref(cfg)), "lnsfaid_set_cfg")

    def decode(self, fix_input, n_groups):
        """fix_input: numpy int8 array, reference layout. Returns (decodedBits int8 array, stats array)."""
        import numpy as np
        N = self.code50.N
        assert fix_input.dtype == np.int8 and fix_input.size == n_groups * GROUP * N and fix_input.flags.c_contiguous
        out = np.empty(n_groups * GROUP * N, dtype=np.int8)
        stats = np.zeros((n_groups, 2), dtype=np.int32)
        self._check(self.lib.lnsfaid_decode(self.ctx, fix_input.ctypes.data, n_groups, out.ctypes.data,
                                            stats.ctypes.data), "lnsfaid_decode")
        return out, stats

    def decode_device(self, d_fix_ptr, n_groups, d_out_ptr, d_stats_ptr=None):
        self._check(self.lib.lnsfaid_decode_device(self.ctx, d_fix_ptr, n_groups, d_out_ptr, d_stats_ptr),
                    "lnsfaid_decode_device")

    def count_errors(self, decoded, input_bits, n_groups):
        out = (C.c_uint64 * 4)()
        ip = input_bits.ctypes.data if input_bits is not None else None
        self._check(self.lib.lnsfaid_count_errors(self.ctx, decoded.ctypes.data, ip, n_groups, out), "lnsfaid_count_errors")
        return list(out)

    def count_errors_device(self, d_decoded_ptr, d_input_ptr, n_groups):
        out = (C.c_uint64 * 4)()
        self._check(self.lib.lnsfaid_count_errors_device(self.ctx, d_decoded_ptr, d_input_ptr, n_groups, out),
                    "lnsfaid_count_errors_device")
        return list(out)

    def comm_init(self, n_ranks, rank, comm_id):
        self._check(self.lib.lnsfaid_comm_init(self.ctx, n_ranks, rank, comm_id), "lnsfaid_comm_init")

    def allreduce_counters(self, counters):
        buf = (C.c_uint64 * 4)(*[int(c) for c in counters])
        self._check(self.lib.lnsfaid_allreduce_counters(self.ctx, buf), "lnsfaid_allreduce_counters")
        return list(buf)

    def select_kernel(self, rows_per_lane):
        self._check(self.lib.lnsfaid_select_kernel(self.ctx, rows_per_lane), "lnsfaid_select_kernel")

    def rows_per_lane(self):
        return self.lib.lnsfaid_kernel_rows_per_lane(self.ctx)

    def select_waves(self, waves_per_codeword):
        self._check(self.lib.lnsfaid_select_waves(self.ctx, waves_per_codeword), "lnsfaid_select_waves")

    def kernel_waves(self):
        return self.lib.lnsfaid_kernel_waves(self.ctx)

    def select_message_store(self, where):
        """0 default, MSG_REGISTERS, MSG_HBM (lnsfaid_select_message_store)"""
        self._check(self.lib.lnsfaid_select_message_store(self.ctx, where), "lnsfaid_select_message_store")

    def message_store(self):
        return self.lib.lnsfaid_message_store(self.ctx)

    def kernel_residency(self):
        """(workgroups per CU of the selected kernel, what its LDS alone allows)"""
        wg, lim = C.c_int32(), C.c_int32()
        self._check(self.lib.lnsfaid_kernel_residency(self.ctx, C.byref(wg), C.byref(lim)), "lnsfaid_kernel_residency")
        return wg.value, lim.value

    def kernel_time(self, reset=False):
        ms = C.c_double()
        n = C.c_uint64()
        self._check(self.lib.lnsfaid_kernel_time(self.ctx, C.byref(ms), C.byref(n), 1 if reset else 0), "lnsfaid_kernel_time")
        return ms.value, n.value

    def close(self):
        if self.ctx:
            self.lib.lnsfaid_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
