"""Systematic encoder for the 50G-PON code, test infrastructure only.

The reference's encoder needs `GenMatrix`, whose data is not shipped (reference Constants_SSE.h:3106-3107 is empty,
.MISSING_LARGE_BLOBS), so its driver can only send the fixed `CodeWord_sym`.  The codeword that carries given
information bits is nevertheless unique: with H = [A | B] (A: 3072 x 14592 information part, B: 3072 x 3072 parity part,
B invertible) the parity bits solve B p = A u over GF(2).  This module solves that system once with bit-packed Gaussian
elimination and encodes random information words, so that the parity tests can also run on non-zero, per-frame different
codewords (the decoders are not sign-symmetric: hard decision is En > 0, zero messages take the sign of En)."""
import numpy as np


class Encoder:
    def __init__(self, code50):
        N, M = code50.N, code50.M
        K = N - M
        pos = np.ctypeslib.as_array(code50.pos_vn).astype(np.int64)
        row_deg = np.repeat(np.array(list(code50.deg)), np.array(list(code50.deg_rows)))
        rows = np.repeat(np.arange(M), row_deg)
        H = np.zeros((M, N), dtype=np.uint8)
        H[rows, pos] = 1
        self.N, self.M, self.K = N, M, K
        self.A = H[:, :K]
        # invert B with Gauss-Jordan on bit-packed rows of [B | I]
        aug = np.concatenate([H[:, K:], np.eye(M, dtype=np.uint8)], axis=1)
        packed = np.packbits(aug, axis=1)
        for col in range(M):
            byte, bit = col >> 3, 7 - (col & 7)
            piv = np.nonzero((packed[col:, byte] >> bit) & 1)[0]
            if piv.size == 0:
                raise ValueError("parity part of H is singular")
            p = col + int(piv[0])
            if p != col:
                packed[[col, p]] = packed[[p, col]]
            mask = ((packed[:, byte] >> bit) & 1).astype(bool)
            mask[col] = False
            packed[mask] ^= packed[col]
        self.Binv = np.unpackbits(packed, axis=1)[:, M:2 * M]
        self.At = np.ascontiguousarray(self.A.T.astype(np.float32))
        self.Binvt = np.ascontiguousarray(self.Binv.T.astype(np.float32))

    def encode(self, info):
        """info: [n, K] bits -> codewords [n, N]."""
        info = np.asarray(info, dtype=np.uint8)
        # float32 BLAS products are exact here: every dot product is an integer below 2^24
        s = np.rint(info.astype(np.float32) @ self.At).astype(np.int64) & 1          # A u
        p = np.rint(s.astype(np.float32) @ self.Binvt).astype(np.int64) & 1          # B^-1 A u
        return np.concatenate([info, p.astype(np.uint8)], axis=1).astype(np.int8)


def qpsk_llr(codewords, eb_n0, seed, scale=13.0, rate=0.8444444):
    """Per-frame codewords [n_frames, N] (n_frames multiple of 32) through QPSK + AWGN + the 4-bit quantiser, in the
    decoder's group layout ([32][K] then [32][M] per group)."""
    cw = np.asarray(codewords, dtype=np.int8)
    n, N = cw.shape
    assert n % 32 == 0
    sigma = 1.0 / np.sqrt(rate * 2 * 10.0 ** (0.1 * eb_n0))
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, N), dtype=np.float32) * np.float32(sigma / np.sqrt(2.0)) + np.where(cw > 0, np.float32(0.707107), np.float32(-0.707107))
    q = np.clip(np.trunc(x * np.float32(scale)), -7, 7).astype(np.int8)
    return to_group_layout(q, N - 3072)


def to_group_layout(frames, k_info):
    """[n_frames, N] frame-major values -> flat array of groups, each [32][K] followed by [32][M]."""
    n, N = frames.shape
    g = frames.reshape(n // 32, 32, N)
    return np.ascontiguousarray(np.concatenate([g[:, :, :k_info].reshape(n // 32, -1), g[:, :, k_info:].reshape(n // 32, -1)], axis=1).reshape(-1))
