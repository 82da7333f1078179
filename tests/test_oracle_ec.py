"""CPU suite: the Python curve reference against OpenSSL (libcrypto, an independent implementation), and the
Jacobian formula model that was used to validate the device formulas, including its exceptional cases."""
import ctypes
import ctypes.util
import random

import pytest

from oracle.pyref_ec import Curve, jac_add, jac_dbl, jac_to_affine

NIDS = {"P-256": 415, "P-384": 715}


def openssl_mul(name, k):
    lib = ctypes.CDLL(ctypes.util.find_library("crypto") or "libcrypto.so.3")
    for f in ("EC_GROUP_new_by_curve_name", "EC_POINT_new", "BN_new", "BN_CTX_new", "BN_bin2bn"):
        getattr(lib, f).restype = ctypes.c_void_p
    grp = ctypes.c_void_p(lib.EC_GROUP_new_by_curve_name(NIDS[name]))
    pt = ctypes.c_void_p(lib.EC_POINT_new(grp))
    ctx = ctypes.c_void_p(lib.BN_CTX_new())
    kb = k.to_bytes(64, "big")
    bn = ctypes.c_void_p(lib.BN_bin2bn(kb, len(kb), None))
    assert lib.EC_POINT_mul(grp, pt, bn, None, None, ctx) == 1
    x, y = ctypes.c_void_p(lib.BN_new()), ctypes.c_void_p(lib.BN_new())
    assert lib.EC_POINT_get_affine_coordinates(grp, pt, x, y, ctx) == 1
    out = []
    for v in (x, y):
        buf = ctypes.create_string_buffer(64)
        n = lib.BN_bn2bin(v, buf)
        out.append(int.from_bytes(buf.raw[:n], "big"))
    return tuple(out)


@pytest.mark.parametrize("name", ["P-256", "P-384"])
def test_reference_agrees_with_openssl(name):
    c = Curve(name)
    random.seed(7)
    for k in [1, 2, 3, c.n - 1, random.randrange(c.n), random.randrange(c.n)]:
        assert c.mul(k, c.g) == openssl_mul(name, k), k
    assert c.mul(c.n, c.g) is None


@pytest.mark.parametrize("name", ["P-256", "P-384"])
def test_jacobian_model_with_exceptional_cases(name):
    c = Curve(name)
    random.seed(3)
    J = lambda P, z: (1, 1, 0, True) if P is None else (P[0] * z * z % c.p, P[1] * z * z * z % c.p, z % c.p, False)
    for t in range(60):
        k1, k2 = random.randrange(c.n), random.randrange(c.n)
        k2 = [k2, k1, c.n - k1, 0, k2, k2][t % 6]
        k1 = 0 if t % 6 == 4 else k1
        P, Q = c.mul(k1, c.g), c.mul(k2, c.g)
        z1, z2 = random.randrange(1, c.p), random.randrange(1, c.p)
        assert jac_to_affine(c, jac_add(c, J(P, z1), J(Q, z2))) == c.add(P, Q)
        assert jac_to_affine(c, jac_dbl(c, J(P, z1))) == c.add(P, P)
