"""pyref_ec.py — Python-integer reference for the elliptic-curve groups (ECqPGroup P-224 / P-256 / P-384 / P-521).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference tree only names the groups (default group P-256: demo/mixnet/.conf:153; P-224 in
demo/mixnet/.checkbaseconf:59); the arithmetic is VCR/VECJ (not in the tree).  Group elements are
affine points on y^2 = x^3 - 3x + b over F_p, the group operation ("mul" in VCR's multiplicative
notation) is point addition, "exp" is scalar multiplication.  Textbook affine formulas with modular
inverses (pow(., -1, p)) — deliberately the slow, obviously-correct form.  Infinity is None.

A second, OpenSSL-backed check lives in tests (ctypes on libcrypto EC_POINT_mul) so that this file
is not the only source of truth for the curve constants.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

Point = Optional[Tuple[int, int]]

CURVES = {
    "P-224": dict(
        p=2**224 - 2**96 + 1,
        n=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFF16A2E0B8F03E13DD29455C5C2A3D,
        b=0xB4050A850C04B3ABF54132565044B0B7D7BFD8BA270B39432355FFB4,
        gx=0xB70E0CBD6BB4BF7F321390B94A03C1D356C21122343280D6115C1D21,
        gy=0xBD376388B5F723FB4C22DFE6CD4375A05A07476444D5819985007E34),
    "P-521": dict(
        p=2**521 - 1,
        n=0x01FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFA51868783BF2F966B7FCC0148F709A5D03BB5C9B8899C47AEBB6FB71E91386409,
        b=0x0051953EB9618E1C9A1F929A21A0B68540EEA2DA725B99B315F3B8B489918EF109E156193951EC7E937B1652C0BD3BB1BF073573DF883D2C34F1EF451FD46B503F00,
        gx=0x00C6858E06B70404E9CD9E3ECB662395B4429C648139053FB521F828AF606B4D3DBAA14B5E77EFE75928FE1DC127A2FFA8DE3348B3C1856A429BF97E7E31C2E5BD66,
        gy=0x011839296A789A3BC0045C8A5FB42C7D1BD998F54449579B446817AFBD17273E662C97EE72995EF42640C550B9013FAD0761353C7086A272C24088BE94769FD16650),
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


class Curve:
    def __init__(self, name: str):
        c = CURVES[name]
        self.name, self.p, self.n, self.b = name, c["p"], c["n"], c["b"]
        self.a = self.p - 3
        self.g: Point = (c["gx"], c["gy"])
        self.nbytes = (self.p.bit_length() + 7) // 8
        assert self.on_curve(self.g)

    def on_curve(self, P: Point) -> bool:
        if P is None:
            return True
        x, y = P
        return 0 <= x < self.p and 0 <= y < self.p and (y * y - (x * x * x + self.a * x + self.b)) % self.p == 0

    def neg(self, P: Point) -> Point:
        return None if P is None else (P[0], (-P[1]) % self.p)

    def add(self, P: Point, Q: Point) -> Point:
        if P is None:
            return Q
        if Q is None:
            return P
        p = self.p
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = (3 * x1 * x1 + self.a) * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    def mul(self, k: int, P: Point) -> Point:
        k %= self.n
        acc: Point = None
        for bit in bin(k)[2:] if k else "":
            acc = self.add(acc, acc)
            if bit == "1":
                acc = self.add(acc, P)
        return acc

    # ---- array operations, VCR names (multiplicative notation) -----------------------------------
    def exp_array(self, X: Sequence[Point], E: Sequence[int]) -> List[Point]:
        return [self.mul(e, P) for P, e in zip(X, E)]

    def exp_fixed(self, B: Point, E: Sequence[int]) -> List[Point]:
        return [self.mul(e, B) for e in E]

    def mul_arrays(self, X: Sequence[Point], Y: Sequence[Point]) -> List[Point]:
        return [self.add(P, Q) for P, Q in zip(X, Y)]

    def prod(self, X: Sequence[Point]) -> Point:
        acc: Point = None
        for P in X:
            acc = self.add(acc, P)
        return acc

    def exp_prod(self, X: Sequence[Point], E: Sequence[int]) -> Point:
        acc: Point = None
        for P, e in zip(X, E):
            acc = self.add(acc, self.mul(e, P))
        return acc

    # ---- wire format used at the C ABI: x || y, fixed width, infinity = all 0xff ------------------
    def enc(self, P: Point) -> bytes:
        nb = self.nbytes
        if P is None:
            return b"\xff" * (2 * nb)
        return P[0].to_bytes(nb, "big") + P[1].to_bytes(nb, "big")

    def dec(self, buf: bytes) -> Point:
        nb = self.nbytes
        if buf == b"\xff" * (2 * nb):
            return None
        return int.from_bytes(buf[:nb], "big"), int.from_bytes(buf[nb:], "big")


# ------------------------------------------------------------------------------------------------------
# Model of the device formulas (Jacobian, lazy field values) used to validate them before they were
# written in HIP: the same operation sequence, every value reduced mod p here.  X, Y, Z, inf.
# ------------------------------------------------------------------------------------------------------
def jac_dbl(c: Curve, P):
    X, Y, Z, inf = P
    p = c.p
    delta = Z * Z % p
    gamma = Y * Y % p
    beta = X * gamma % p
    alpha = 3 * (X - delta) * (X + delta) % p
    X3 = (alpha * alpha - 8 * beta) % p
    Z3 = ((Y + Z) * (Y + Z) - gamma - delta) % p
    Y3 = (alpha * (4 * beta - X3) - 8 * gamma * gamma) % p
    return X3, Y3, Z3, inf


def jac_add(c: Curve, P, Q):
    X1, Y1, Z1, inf1 = P
    X2, Y2, Z2, inf2 = Q
    if inf1:
        return Q
    if inf2:
        return P
    p = c.p
    Z1Z1 = Z1 * Z1 % p
    Z2Z2 = Z2 * Z2 % p
    U1 = X1 * Z2Z2 % p
    U2 = X2 * Z1Z1 % p
    S1 = Y1 * Z2 % p * Z2Z2 % p
    S2 = Y2 * Z1 % p * Z1Z1 % p
    H = (U2 - U1) % p
    rr = (S2 - S1) % p
    if H == 0:
        if rr == 0:
            return jac_dbl(c, P)
        return 1, 1, 0, True
    I = 4 * H * H % p
    J = H * I % p
    r = 2 * rr % p
    V = U1 * I % p
    X3 = (r * r - J - 2 * V) % p
    Y3 = (r * (V - X3) - 2 * S1 * J) % p
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) % p * H % p
    return X3, Y3, Z3, False


def jac_to_affine(c: Curve, P) -> Point:
    X, Y, Z, inf = P
    if inf:
        return None
    zi = pow(Z, -1, c.p)
    return X * zi * zi % c.p, Y * zi * zi * zi % c.p
