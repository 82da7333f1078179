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


def inv_perm(pi):
    try:
        import numpy as np
        if isinstance(pi, np.ndarray) or len(pi) > 4096:
            a = np.asarray(pi, dtype=np.int64)
            inv = np.empty(len(a), dtype=np.uint32)
            inv[a] = np.arange(len(a), dtype=np.uint32)
            return inv
    except ImportError:      # pragma: no cover
        pass
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
        self.permutation = permutation
        self.commitment = self.identityCommitment.permute(self.permutation)   # :215
        return self.commitment

    def raise_(self, raisedExponent: int):
        self.raisedCommitment = self.commitment.exp(raisedExponent)           # :357
        return self.raisedCommitment

    def shrink(self, noCiphertexts: int, keepList=None):
        """:390-471.  Prover (``keepList is None``): the positions of the commitment that commit to the first
        ``noCiphertexts`` generators are kept (:398-405; with this package's gather convention u[i] = X[pi[i]] that is
        ``pi[i] < n``), exponents are cut to [0, n) (:415), the permutation is compressed (:419).  Verifier: the keep
        list read from the prover, replaced by the trivial one unless it has exactly n flags set (:424-447).  Both:
        ``commitment.extract(keepList)`` (:462-463), the raised commitment likewise on the verifier's side (:466-469)."""
        n = noCiphertexts
        if keepList is None:
            keepList = [src < n for src in self.permutation]
            old = self.exponents
            self.exponents = old.copyOfRange(0, n)
            old.free()
            self.permutation = [src for src in self.permutation if src < n]
        else:
            keepList = list(keepList)
            if len(keepList) != self.commitment.size() or sum(1 for k in keepList if k) != n:
                keepList = [i < n for i in range(self.commitment.size())]
        old = self.commitment
        self.commitment = old.extract(keepList)
        old.free()
        if getattr(self, "raisedCommitment", None) is not None:
            old = self.raisedCommitment
            self.raisedCommitment = old.extract(keepList)
            old.free()
        return keepList


class SecureRandomSource:
    """The random source a prover uses by default: the operating system's CSPRNG (``os.urandom``), the counterpart of
    the reference's ``RandomDevice /dev/urandom`` (demo/mixnet/.checkbaseconf:124).  Single ring elements are drawn
    with ``rbitlen`` bits of slack and reduced (statistical distance 2^-rbitlen from uniform); the N-sized draws of a
    proof are expanded on the GPU from 32 fresh bytes each (``array_seed``, see ``vmn_random_source`` in
    include/vmnproofs.h), so no N-sized host buffer exists.  ``ring_array`` / ``int_array`` remain for drivers that
    want host rows (the Python mirror)."""

    def __init__(self, q: int, rbitlen: int = 100):
        import os
        self._urandom = os.urandom
        self.q, self.rbitlen = q, rbitlen

    def array_seed(self) -> bytes:
        return self._urandom(32)

    def ring_element(self) -> int:
        nb = (self.q.bit_length() + self.rbitlen + 7) // 8
        return int.from_bytes(self._urandom(nb), "big") % self.q

    def ring_array(self, n: int) -> List[int]:
        return [self.ring_element() for _ in range(n)]

    def int_array(self, n: int, bits: int) -> List[int]:
        nb = (bits + 7) // 8
        mask = (1 << bits) - 1
        return [int.from_bytes(self._urandom(nb), "big") & mask for _ in range(n)]

    def permutation(self, n: int) -> List[int]:
        import random
        pi = list(range(n))
        random.SystemRandom().shuffle(pi)
        return pi


class InsecureShaRandomSource:
    """NOT FOR PRODUCTION: a fixed-seed, reproducible stream (tests, benchmark).
    Deterministic random source for the benchmark and the tests (SHA-256 counter stream).  VCR's
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


class InsecureBulkRandomSource:
    """NOT FOR PRODUCTION: numpy's PCG64 is not a cryptographic generator (benchmark inputs only).
    numpy-backed random source for large N (benchmark): hands out big-endian byte blocks of the
    group's wire width instead of Python integers, so that no per-element Python work happens.
    Ring elements are uniform below 2^(bits(q)-1) <= q (top bit cleared): statistically as good for
    a throughput run, and always in range."""

    def __init__(self, seed: int, q: int, nbytes: int):
        import numpy as np
        self.np = np
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.q, self.nbytes = q, nbytes
        self.qbits = q.bit_length()

    def _block(self, n: int, bits: int) -> bytes:
        np = self.np
        a = np.zeros((n, self.nbytes), dtype=np.uint8)
        nb = (bits + 7) // 8
        a[:, self.nbytes - nb:] = self.rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
        if bits % 8:
            a[:, self.nbytes - nb] &= (1 << (bits % 8)) - 1
        return a.tobytes()

    def int_array(self, n: int, bits: int) -> bytes:
        return self._block(n, min(bits, self.qbits - 1))

    def ring_array(self, n: int) -> bytes:
        return self._block(n, self.qbits - 1)

    def ring_element(self) -> int:
        return int.from_bytes(self._block(1, self.qbits - 1), "big")

    def permutation(self, n: int):
        return self.rng.permutation(n).astype(self.np.uint32)

    def array_seed(self) -> bytes:
        """32 bytes for a device-expanded array draw (vmn_random_source.array_seed)."""
        return self.rng.integers(0, 256, size=32, dtype=self.np.uint8).tobytes()
