"""mixnet.py — the arithmetic lines of the shuffler and of the permutation commitment.

Mirrors
  * ``ShufflerElGamalSession`` re-encryption + permutation (row A0),
    ref: src/java/com/verificatum/protocol/mixnet/ShufflerElGamalSession.java:400-409
    (exponents, ``widePublicKey.exp``), :273-278 (``input.mul(reencFactors)``, ``permute(inverse)``),
    raised generators :498-507 (row A5);
  * ``PermutationCommitment.precompute`` / raised commitment (row A4),
    ref: src/java/com/verificatum/protocol/mixnet/PermutationCommitment.java:189-215, :357.

Ciphertext arrays are lists of 2ω component arrays (see hvzk.py).
"""
from __future__ import annotations

import hashlib
from typing import List, Sequence

RAISED_BITLENGTH = 50        # ShufflerElGamalSession.java:75


def inv_perm(pi: Sequence[int]) -> List[int]:
    inv = [0] * len(pi)
    for i, j in enumerate(pi):
        inv[j] = i
    return inv


def reencFactors(group, widePublicKey: Sequence[int], reencExponents):
    """``widePublicKey.exp(reencExponents)`` (:407): component c of the key to the exponents of its column."""
    width = len(widePublicKey) // 2
    return [group.exp(pk, reencExponents[c % width]) for c, pk in enumerate(widePublicKey)]


def reencrypt(ciphertexts, factors, permutation: Sequence[int]):
    """``input.mul(reencFactors)`` then ``reenc.permute(permutation.inv())`` (:273-278)."""
    inverse = inv_perm(permutation)
    out = []
    for c, f in zip(ciphertexts, factors):
        reenc = c.mul(f)
        out.append(reenc.permute(inverse))
        reenc.free()
    return out


def raisedGenerators(generators, raisedExponent: int):
    """``generators.exp(raisedExponent)`` (:506)."""
    return generators.exp(raisedExponent)


class PermutationCommitment:
    """ref: mixnet/PermutationCommitment.java — precompute :148-219, raised commitment :357."""

    def __init__(self, group, generators):
        self.G, self.generators = group, generators

    def precompute(self, exponents_ints: Sequence[int], permutation: Sequence[int]):
        G = self.G
        self.exponents = G.ringArray(exponents_ints)
        tmp = G.exp(G.g, self.exponents)                     # pGroup.getg().exp(exponents)   :200
        self.identityCommitment = self.generators.mul(tmp)   # generators.mul(tmp)            :201
        tmp.free()
        self.permutation = list(permutation)
        self.commitment = self.identityCommitment.permute(self.permutation)   # :215
        return self.commitment

    def raise_(self, raisedExponent: int):
        self.raisedCommitment = self.commitment.exp(raisedExponent)           # :357
        return self.raisedCommitment


class ShaRandomSource:
    """Deterministic random source for the benchmark and the tests (SHA-256 counter stream).  VCR's
    own ``randomElementArray`` sampling is not part of the reference tree; any source works for the
    arithmetic, and parity tests feed the same tape to the oracle."""

    def __init__(self, seed: bytes, q: int):
        self.seed, self.q, self.ctr = seed, q, 0

    def _bytes(self, n: int) -> bytes:
        out = bytearray()
        while len(out) < n:
            out += hashlib.sha256(self.seed + self.ctr.to_bytes(8, "big")).digest()
            self.ctr += 1
        return bytes(out[:n])

    def int_array(self, n: int, bits: int) -> List[int]:
        nb = (bits + 7) // 8
        buf = self._bytes(n * nb)
        mask = (1 << bits) - 1
        return [int.from_bytes(buf[i * nb:(i + 1) * nb], "big") & mask for i in range(n)]

    def ring_array(self, n: int) -> List[int]:
        nb = (self.q.bit_length() + 7) // 8 + 8
        buf = self._bytes(n * nb)
        return [int.from_bytes(buf[i * nb:(i + 1) * nb], "big") % self.q for i in range(n)]

    def ring_element(self) -> int:
        return self.ring_array(1)[0]

    def permutation(self, n: int) -> List[int]:
        keys = self.int_array(n, 64)
        return sorted(range(n), key=lambda i: (keys[i], i))
