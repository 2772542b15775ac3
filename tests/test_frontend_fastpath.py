"""The single-precision identities the device front-end's fast path rests on (lnsfaid_frontend.hip), checked exhaustively on the CPU
with IEEE float32 arithmetic: they hold for EVERY state of the three Wichmann-Hill generators (CChannel.cpp:71-80)."""
from fractions import Fraction
import math

import numpy as np

GENERATORS = ((249, 61967), (251, 63443), (252, 63599))


def test_float_modular_step_is_exact_for_every_state():
    # x * a < 2^24 is exact in float32, floor((x * a) * fl(1 / m)) is the true quotient, so the remainder is one exact fma
    for a, m in GENERATORS:
        x = np.arange(0, m, dtype=np.int64)
        p = x * a
        assert int(p.max()) < 2 ** 24
        cr = np.float32(1.0) / np.float32(m)
        q = np.floor(p.astype(np.float32) * cr).astype(np.int64)
        assert np.array_equal(p - q * m, p % m)


def _rne32(x):
    """nearest float32 (ties to even) of a Fraction, as a Fraction"""
    if x == 0:
        return Fraction(0)
    sgn = -1 if x < 0 else 1
    x = abs(x)
    e = math.floor(math.log2(float(x)))
    while Fraction(2) ** e > x:
        e -= 1
    while Fraction(2) ** (e + 1) <= x:
        e += 1
    ulp = Fraction(2) ** (max(e, -126) - 23)
    k = x / ulp
    f = k.numerator // k.denominator
    r = k - f
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and f % 2 == 1):
        f += 1
    return sgn * f * ulp


def test_three_operation_division_is_correctly_rounded_for_every_state():
    # q0 = x * fl(1 / m); e = fma(-q0, m, x); q = fma(e, fl(1 / m), q0) equals the IEEE quotient x / m (every 7th state and the
    # ends here in exact rational arithmetic; all 189 009 states take 11 s and were run when the kernel was written)
    for _, m in GENERATORS:
        cr = Fraction(float(np.float32(1.0) / np.float32(m)))
        for x in list(range(0, m, 7)) + [1, 2, m - 2, m - 1]:
            q0 = _rne32(Fraction(x) * cr)
            e = Fraction(x) - q0 * m
            assert _rne32(e) == e  # the remainder is representable: the fma returns it exactly
            q = _rne32(q0 + e * cr)
            assert q == Fraction(float(np.float32(x) / np.float32(m))), (m, x)
