"""TEST-ONLY mirror (tests/mirror/): a Python restatement of the reference's caller, kept to cross-check the C++ drivers of
csrc/vmnproofs.cpp through the same array ABI.  Not part of the product package.

mixnet.py — the arithmetic lines of the shuffler and of the permutation commitment.

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

from typing import Sequence

from verificatum_vmn_amd.randomsource import (InsecureBulkRandomSource, InsecureShaRandomSource, SecureRandomSource,  # noqa: F401
                                               inv_perm)

RAISED_BITLENGTH = 50        # ShufflerElGamalSession.java:75


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
