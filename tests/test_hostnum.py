"""CPU suite: the host-side big-number code of the C++ proof drivers (csrc/hostnum64.h: Montgomery arithmetic on
64-bit limbs for the O(1) scalars of a proof) against Python integers."""
import os
import subprocess

import pytest

from conftest import ROOT
from oracle import pyref


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hostnum") / "hostnum_harness")
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tests", "hostnum_harness.cpp")], check=True)
    return exe


def run(exe, n, a, b, e):
    out = subprocess.run([exe] + ["%x" % v for v in (n, a, b, e)], check=True, capture_output=True, text=True).stdout.split()
    return [int(x, 16) for x in out]


@pytest.mark.parametrize("bits", [2048, 3072])
def test_modp_scalars(bits, harness):
    p, q, g = pyref.modp_group(bits)
    vals = pyref.stream_ints(b"hostnum%d" % bits, 6, p)
    for k in range(3):
        a, b = 1 + vals[2 * k] % (p - 1), vals[2 * k + 1]
        e = pyref.stream_ints(b"hostnum-e%d" % k, 1, q)[0] if k else (1 << 612) + 12345      # also a wide exponent mod small n
        got = run(harness, p, a, b, e)
        assert got == [a * b % p, pow(a, e, p), pow(a, -1, p), e % p, (a + b) % p, (-a) % p]


def test_order_of_p256_and_edge_values(harness):
    n = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551
    for a, b, e in [(1, 0, 0), (n - 1, n - 1, n - 1), (2, 3, (1 << 612) - 1), (12345, n - 2, 1 << 300)]:
        got = run(harness, n, a, b, e)
        assert got == [a * b % n, pow(a, e, n), pow(a, -1, n), e % n, (a + b) % n, (-a) % n]


def _jacobi(a, n):
    a %= n
    t = 1
    while a:
        while a % 2 == 0:
            a //= 2
            if n % 8 in (3, 5):
                t = -t
        a, n = n, a
        if a % 4 == 3 and n % 4 == 3:
            t = -t
        a %= n
    return t if n == 1 else 0


@pytest.mark.parametrize("bits", [2048, 3072])
def test_jacobi_symbol_is_subgroup_membership_for_safe_primes(bits, harness):
    """The membership test of the single elements of a commitment (HostGroup::check_elements): symbol 1 <=> x^q = 1."""
    p, q, g = pyref.modp_group(bits)
    vals = [1, p - 1, 2, 1 << 64, (1 << 640) + (1 << 64), 0] + pyref.stream_ints(b"jac%d" % bits, 14, p)
    out = subprocess.run([harness, "jac", "%x" % p] + ["%x" % v for v in vals], check=True, capture_output=True, text=True).stdout.split()
    got = [int(x) for x in out]
    assert got == [_jacobi(v, p) for v in vals]
    assert got[:5] == [1 if pow(v, q, p) == 1 else -1 for v in vals[:5]] and got[5] == 0
    # a composite modulus: 0 when not coprime
    n = 3 * 5 * 7 * 11 * 13 * 17
    vals = list(range(0, 60))
    out = subprocess.run([harness, "jac", "%x" % n] + ["%x" % v for v in vals], check=True, capture_output=True, text=True).stdout.split()
    assert [int(x) for x in out] == [_jacobi(v, n) for v in vals]


@pytest.mark.parametrize("name", ["P-256", "P-384"])
def test_host_curve_points_against_the_affine_reference(name, harness):
    """csrc/hostcurve.h (single points of the C++ proof drivers) against oracle/pyref_ec.py, with the exceptional
    cases: infinity operands, P + P, P + (-P), exponents 0, n, above n."""
    from oracle.pyref_ec import Curve
    c = Curve(name)
    nb = (c.p.bit_length() + 7) // 8
    enc = lambda P: "ff" * (2 * nb) if P is None else "%0*x%0*x" % (2 * nb, P[0], 2 * nb, P[1])

    def run_ec(A, B, e):
        out = subprocess.run([harness, "ec", "%x" % c.p, enc(A), enc(B), "%x" % e], check=True, capture_output=True, text=True).stdout.split()
        return out

    ks = pyref.stream_ints(b"hostcurve" + name.encode(), 4, c.n)
    P, Q = c.mul(ks[0], c.g), c.mul(ks[1], c.g)
    cases = [(P, Q, ks[2]), (P, P, 0), (P, c.neg(P), c.n), (None, Q, ks[3]), (P, None, c.n + 5), (c.g, c.g, (1 << 612) + 77),
             (P, Q, (c.n + 1) // 2), (P, Q, c.n - 1)]
    for A, B, e in cases:
        got = run_ec(A, B, e)
        assert got == [enc(c.mul(e % c.n, A) if A is not None else None), enc(c.add(A, B)), enc(c.neg(A) if A is not None else None)], (A, B, e)
