"""pyref.py — pure-Python restatement (built-in ``pow`` / ``int``) of the array operations of the
Verificatum Mix-Net hot path.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Independent of GMP: it is the third leg that pins the oracle (GMP in ``vmn_oracle.c``, Python
integers here, golden vectors in ``tests/golden``).  The semantics of each VCR call are inferred
from the reference's call sites and comments; citations are to
``/root/reference/src/java/com/verificatum/protocol`` (``P/``).
"""
from __future__ import annotations

import hashlib
from typing import List, Sequence, Tuple

# RFC 2409 / RFC 3526 MODP groups: safe primes p = 2q + 1 defined by the digits of pi,
#     p = 2^n - 2^(n-64) - 1 + 2^64 * (floor(2^(n-130) * pi) + c).
# 4 = 2^2 generates the order-q subgroup.  BASELINE.json's configs name the 2048-bit ("group 14")
# and 3072-bit ("group 15") moduli.
_RFC_MODP_C = {768: 149686, 1024: 129093, 1536: 741804, 2048: 124476, 3072: 1690314, 4096: 240904, 6144: 929484, 8192: 4743158}

RFC3526_14_HEX = (
    "FFFFFFFFFFFFFFFFC90FDAA22168C234C4C6628B80DC1CD129024E088A67CC74020BBEA63B139B22514A08798E3404DD"
    "EF9519B3CD3A431B302B0A6DF25F14374FE1356D6D51C245E485B576625E7EC6F44C42E9A637ED6B0BFF5CB6F406B7ED"
    "EE386BFB5A899FA5AE9F24117C4B1FE649286651ECE45B3DC2007CB8A163BF0598DA48361C55D39A69163FA8FD24CF5F"
    "83655D23DCA3AD961C62F356208552BB9ED529077096966D670C354E4ABC9804F1746C08CA18217C32905E462E36CE3B"
    "E39E772C180E86039B2783A2EC07A28FB5C55DF06F4C52C9DE2BCBF6955817183995497CEA956AE515D2261898FA0510"
    "15728E5A8AACAA68FFFFFFFFFFFFFFFF")
RFC3526_14_P = int(RFC3526_14_HEX, 16)


def _pi_scaled(bits: int) -> int:
    """floor(pi * 2^bits) by Machin's formula with integer arithmetic."""
    guard = 64
    one = 1 << (bits + guard)

    def arctan_inv(x: int) -> int:
        total = term = one // x
        x2 = x * x
        k = 1
        while term:
            term //= x2
            k += 2
            total += (-1 if (k // 2) % 2 else 1) * (term // k)
        return total

    return (4 * (4 * arctan_inv(5) - arctan_inv(239))) >> guard


def rfc_modp_prime(bits: int) -> int:
    """The RFC 2409 / RFC 3526 safe prime of the given size (768 ... 8192)."""
    c = _RFC_MODP_C[bits]
    return (1 << bits) - (1 << (bits - 64)) - 1 + (1 << 64) * ((_pi_scaled(bits - 130)) + c)


def sha_stream(seed: bytes, nbytes: int) -> bytes:
    """SHA-256 counter-mode byte stream (the synthetic-input generator of tests and bench)."""
    out = bytearray()
    ctr = 0
    while len(out) < nbytes:
        out += hashlib.sha256(seed + ctr.to_bytes(8, "big")).digest()
        ctr += 1
    return bytes(out[:nbytes])


def stream_ints(seed: bytes, n: int, modulus: int) -> List[int]:
    """n integers in [0, modulus) from the SHA-256 counter stream (64 extra bits: negligible bias)."""
    nb = (modulus.bit_length() + 7) // 8 + 8
    buf = sha_stream(seed, n * nb)
    return [int.from_bytes(buf[i * nb:(i + 1) * nb], "big") % modulus for i in range(n)]


def is_probable_prime(n: int, rounds: int = 24) -> bool:
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for sp in small:
        if n % sp == 0:
            return n == sp
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in small[:rounds]:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def find_safe_prime(bits: int, seed: bytes) -> int:
    """Deterministic safe prime p = 2q + 1 with p = 7 (mod 8), searched upwards from a
    SHA-256-derived start (used once by tests/golden/gen_golden.py for the 512-bit test group that
    mirrors ModPGroup(512) of TestPoSCBasicTW.java:69-140)."""
    start = int.from_bytes(sha_stream(seed, bits // 8), "big") | (1 << (bits - 1))
    q = (start >> 1) | 3                                  # q = 3 mod 4  =>  p = 2q+1 = 7 mod 8
    small = [3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113]
    while True:
        p = 2 * q + 1
        if all(q % s and p % s for s in small) and pow(2, q - 1, q) == 1 and pow(2, p - 1, p) == 1:
            if is_probable_prime(q) and is_probable_prime(p):
                return p
        q += 4


def modp_group(bits: int) -> Tuple[int, int, int]:
    """(p, q, g) of the RFC safe-prime group of the requested size, g = 4."""
    p = RFC3526_14_P if bits == 2048 else rfc_modp_prime(bits)
    return p, (p - 1) // 2, 4


# --------------------------------------------------------------------------------------------
# array operations (lists of Python ints)
# --------------------------------------------------------------------------------------------
def exp_array(xs: Sequence[int], es: Sequence[int], p: int) -> List[int]:
    """K1a ``X.exp(E)``.  ref: P/hvzk/PoSBasicTW.java:1032; P/hvzk/PoSCBasicTW.java:694."""
    return [pow(x, e, p) for x, e in zip(xs, es)]


def exp_scalar(xs: Sequence[int], e: int, p: int) -> List[int]:
    """K1b ``X.exp(e)``.  ref: P/mixnet/ShufflerElGamalSession.java:506; P/mixnet/PermutationCommitment.java:357."""
    return [pow(x, e, p) for x in xs]


def exp_fixed(base: int, es: Sequence[int], p: int) -> List[int]:
    """K2 ``g.exp(E)``.  ref: P/mixnet/ShufflerElGamalSession.java:407; P/hvzk/PoSBasicTW.java:447, 606."""
    return [pow(base, e, p) for e in es]


def exp_prod(xs: Sequence[int], es: Sequence[int], p: int) -> int:
    """K3 ``X.expProd(E)``.  ref: P/hvzk/PoSBasicTW.java:408-409, 481, 690, 1021, 1063."""
    acc = 1
    for x, e in zip(xs, es):
        acc = acc * pow(x, e, p) % p
    return acc


def mul(xs: Sequence[int], ys: Sequence[int], p: int) -> List[int]:
    """K4 ``X.mul(Y)``.  ref: P/mixnet/ShufflerElGamalSession.java:273 (re-encryption)."""
    return [x * y % p for x, y in zip(xs, ys)]


def prod(xs: Sequence[int], p: int) -> int:
    """K5 ``X.prod()``.  ref: P/hvzk/PoSBasicTW.java:1013."""
    acc = 1
    for x in xs:
        acc = acc * x % p
    return acc


def permute(xs: Sequence[int], perm: Sequence[int]) -> List[int]:
    """K7 ``X.permute(pi)`` as a gather: result[i] = X[pi(i)] (comment at P/hvzk/PoSBasicTW.java:444;
    see SURVEY.md App. B on the convention)."""
    return [xs[j] for j in perm]


def shift_push(xs: Sequence[int], el: int) -> List[int]:
    """K7 ``X.shiftPush(x)`` = (x, X0, ..., X(N-2)).  ref: P/hvzk/PoSBasicTW.java:625-638."""
    return [el] + list(xs[:-1])


def rec_lin(bs: Sequence[int], es: Sequence[int], q: int) -> Tuple[List[int], int]:
    """K8 ``b.recLin(e)``: x0 = b0, xi = x(i-1)*ei + bi.  The reference's commented loop:
    P/hvzk/PoSBasicTW.java:583-598."""
    xs: List[int] = []
    x = 0
    for i, (b, e) in enumerate(zip(bs, es)):
        x = b % q if i == 0 else (x * e + b) % q
        xs.append(x)
    return xs, (xs[-1] if xs else 0)


def prods(es: Sequence[int], q: int) -> List[int]:
    """K8 ``e.prods()`` prefix products.  ref: P/hvzk/PoSBasicTW.java:600-604."""
    out, acc = [], 1
    for e in es:
        acc = acc * e % q
        out.append(acc)
    return out


def mul_add(xs: Sequence[int], v: int, ys: Sequence[int], q: int) -> List[int]:
    """K8 ``x.mulAdd(v, y)``.  ref: P/hvzk/PoSBasicTW.java:865-878."""
    return [(x * v + y) % q for x, y in zip(xs, ys)]


def inner_product(xs: Sequence[int], ys: Sequence[int], q: int) -> int:
    """K8 ``r.innerProduct(e)``.  ref: P/hvzk/PoSBasicTW.java:861."""
    return sum(x * y for x, y in zip(xs, ys)) % q
