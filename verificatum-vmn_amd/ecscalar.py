"""ecscalar.py — host-side arithmetic on *single* curve points (affine, Python integers).

Only the O(1) scalars of a proof live here (A', C', D', the final equality checks): the same part that
stays in VCR's scalar classes in the reference.  Arrays of points are handled on the GPU
(``csrc/ec_kernels.h``).  Points are ``(x, y)`` tuples, the point at infinity is ``None``.
"""
from __future__ import annotations

from typing import Optional, Tuple

Point = Optional[Tuple[int, int]]

CURVES = {
    "P-256": dict(
        p=0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF,
        n=0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551,
        b=0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B,
        gx=0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296,
        gy=0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5),
    "P-384": dict(
        p=2**384 - 2**128 - 2**96 + 2**32 - 1,
        n=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFC7634D81F4372DDF581A0DB248B0A77AECEC196ACCC52973,
        b=0xB3312FA7E23EE7E4988E056BE3F82D19181D9C6EFE8141120314088F5013875AC656398D8A2ED19D2A85C8EDD3EC2AEF,
        gx=0xAA87CA22BE8B05378EB1C71EF320AD746E1D3B628BA79B9859F741E082542A385502F25DBF55296C3A545E3872760AB7,
        gy=0x3617DE4A96262C6F5D9E98BF9292DC29F8F41DBD289A147CE9DA3113B5F0B8C00A60B1CE1D7E819D7A431D7C90EA0E5F),
}


def add(P: Point, Q: Point, p: int) -> Point:
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = (3 * x1 * x1 - 3) * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    return x3, (lam * (x1 - x3) - y1) % p


def neg(P: Point, p: int) -> Point:
    return None if P is None else (P[0], (-P[1]) % p)


def mul(k: int, P: Point, p: int, n: int) -> Point:
    k %= n
    acc: Point = None
    for bit in bin(k)[2:] if k else "":
        acc = add(acc, acc, p)
        if bit == "1":
            acc = add(acc, P, p)
    return acc
