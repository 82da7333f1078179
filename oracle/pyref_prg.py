"""pyref_prg.py — TEST INFRASTRUCTURE (oracle): VCR's PRGHeuristic / RandomOracle and the two derivations the
reference makes from them, restated from their published definition (the Verificatum verifier specification
describes both constructions and gives known-answer vectors; the classes themselves live in verificatum-vcr, which
is not in the reference tree):

  PRG(seed):          block_i = H(seed || uint32_be(i)), i = 0, 1, ...            (PRGHeuristic)
  RO_nout(data):      first ceil(nout/8) bytes of PRG(H(uint32_be(nout) || data)), superfluous leading bits cleared
  random vector:      e_i = the i-th ceil(n_e/8) bytes of PRG(seed), leading bits cleared   (PoSBasicTW.java:533-538)
  generators (ModP):  t_i = the i-th ceil((n_p + n_r)/8) bytes, leading bits cleared; h_i = t_i^((p-1)/q) mod p
                      (IndependentGeneratorsRO.java:117-130 -> pGroup.randomElementArray(n, prg, rbitlen))

The PRG / RO constructions are pinned by the published vectors in tests/test_prg.py; the two derivations follow the
specification's text and are not pinned by vectors (the header of DESIGN.md §2 says so).
"""
import hashlib
from typing import List


def prg_bytes(seed: bytes, nbytes: int, hashname: str = "sha256") -> bytes:
    out = bytearray()
    ctr = 0
    while len(out) < nbytes:
        out += hashlib.new(hashname, seed + ctr.to_bytes(4, "big")).digest()
        ctr += 1
    return bytes(out[:nbytes])


def random_oracle(data: bytes, nout: int, hashname: str = "sha256") -> bytes:
    seed = hashlib.new(hashname, nout.to_bytes(4, "big") + data).digest()
    nb = (nout + 7) // 8
    out = bytearray(prg_bytes(seed, nb, hashname))
    if nout % 8:
        out[0] &= (1 << (nout % 8)) - 1
    return bytes(out)


def random_integers(seed: bytes, n: int, bits: int) -> List[int]:
    vb = (bits + 7) // 8
    stream = prg_bytes(seed, n * vb)
    mask = (1 << bits) - 1
    return [int.from_bytes(stream[i * vb:(i + 1) * vb], "big") & mask for i in range(n)]


def modp_generators(seed: bytes, n: int, p: int, q: int, rbitlen: int) -> List[int]:
    cof = (p - 1) // q
    return [pow(t % p, cof, p) for t in random_integers(seed, n, p.bit_length() + rbitlen)]


def sqrt_mod(a: int, p: int):
    """A square root of a modulo the odd prime p, or None: a^((p+1)/4) when p = 3 mod 4, Tonelli-Shanks otherwise
    (P-224: p - 1 = 2^96 (2^128 - 1)).  Which of the two roots comes out is irrelevant to the callers (they take the smaller)."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, x = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, tt = 0, t
        while tt != 1:
            tt = tt * tt % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c, t, x = i, b * b % p, t * b * b % p, x * b % p
    return x


def ec_generators(seed: bytes, n: int, curve, rbitlen: int, hashname: str = "sha256"):
    """ECqPGroup.randomElementArray(n, prg, rbitlen) as the product restates it from the verifier specification
    [NOT-IN-REF: VCR's procedure; unpinned]: candidate j = the j-th (bits(p) + rbitlen)-bit integer of the PRG stream,
    x = t mod p; kept when x^3 + ax + b is a square mod p (sqrt_mod), the point being (x, min(z, p - z)); the array holds the
    first n kept candidates in order.  `curve`: oracle/pyref_ec.Curve."""
    p, a, b = curve.p, curve.a, curve.b
    bits = p.bit_length() + rbitlen
    vb = (bits + 7) // 8
    out, j = [], 0
    chunk = max(64, 3 * n)
    stream = b""
    while len(out) < n:
        if len(stream) < (j + 1) * vb:
            stream = prg_bytes(seed, (j + chunk) * vb, hashname)
        t = int.from_bytes(stream[j * vb:(j + 1) * vb], "big") & ((1 << bits) - 1)
        j += 1
        x = t % p
        rhs = (x * x * x + a * x + b) % p
        z = sqrt_mod(rhs, p)
        if z is not None:
            out.append((x, min(z, p - z)))
    return out
