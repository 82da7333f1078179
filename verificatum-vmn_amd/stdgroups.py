"""stdgroups.py — the fixed safe-prime groups the benchmark configurations name (BASELINE.json: "RFC 3526 group 14 /
15"; the reference's demo uses ``vog -gen ModPGroup -fixed 2048``, demo/mixnet/.conf:191).

The RFC 2409 / RFC 3526 MODP primes are defined by a formula, p = 2^n - 2^(n-64) - 1 + 2^64 (floor(2^(n-130) pi) + c),
so they are computed here (pi by the Gauss-Legendre iteration on integers) instead of being stored; the generator
of the order-q subgroup, q = (p-1)/2, is 4.
"""
from __future__ import annotations

from math import isqrt
from typing import Tuple

_C = {1536: 741804, 2048: 124476, 3072: 1690314, 4096: 240904}


def _pi_times_2_to(bits: int) -> int:
    """floor(pi * 2^bits), Gauss-Legendre (quadratic convergence) in fixed point with guard bits."""
    guard = 96
    w = bits + guard
    one = 1 << w
    a, b, t, pw = one, isqrt(one * one // 2), one // 4, 1
    for _ in range(max(4, w.bit_length())):
        an = (a + b) // 2
        b = isqrt(a * b)
        t -= pw * (a - an) * (a - an) // one
        a, pw = an, 2 * pw
    return ((a + b) * (a + b) // (4 * t)) >> guard


def rfc3526_prime(bits: int) -> int:
    if bits not in _C:
        raise ValueError(f"no RFC 3526 group of {bits} bits (have {sorted(_C)})")
    return (1 << bits) - (1 << (bits - 64)) - 1 + (1 << 64) * (_pi_times_2_to(bits - 130) + _C[bits])


def modp_group(bits: int) -> Tuple[int, int, int]:
    """(p, q, g): RFC 3526 safe prime of that size, q = (p - 1) / 2, g = 4."""
    p = rfc3526_prime(bits)
    return p, (p - 1) // 2, 4
