"""elgamal.py — the arithmetic of verifiable threshold decryption.

Mirrors
  * ``DistrElGamalSession.decrypt`` arithmetic lines,
    ref: src/java/com/verificatum/protocol/elgamal/DistrElGamalSession.java:365-385 (decryption factors
    f_j = u^(-x_j / c)), :536-538 (plaintexts = v * combined factors);
  * ``DistrElGamalSessionBasic``, ref: elgamal/DistrElGamalSessionBasic.java — prodFactor :318-344,
    modifiedLagrangeCoefficient(s) :358-452, combineDecryptionFactors :465-503, setBatchVector :513-518,
    batchInput :524-526, commit :534-540, reply :595-598, combine :642-678, batchCombined :683-685,
    verifyCombined :693-700, batch :707-709, verify :718-727.

Array work (per-element exponentiations with the secret share, the per-element simultaneous
exponentiation with the Lagrange integers, the batching multi-exponentiations, the final products)
runs on the GPU; the O(k) scalars of the Chaum–Pedersen proof stay on the host like in hvzk.py.
"""
from __future__ import annotations

from typing import List, Sequence

ODD_PRIMES = [3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97]


def primeLog(number: int, prime: int) -> int:
    """Largest power of ``prime`` not exceeding ``number`` (:294-303)."""
    resA = resB = 1
    while resB <= number:
        resA = resB
        resB *= prime
    return resA


def prodFactor(q: int, k: int) -> int:
    """c = (prod over primes p <= k of p^floor(log_p k))^2 mod q (:318-344)."""
    res, prime, i = 1, 2, 0
    while prime <= k:
        res *= primeLog(k, prime)
        prime = ODD_PRIMES[i]
        i += 1
    return res * res % q


def modifiedLagrangeCoefficients(q: int, correct: Sequence[bool], k: int, threshold: int) -> List[int]:
    """Integers of smallest absolute value (possibly negative), :358-452.  ``correct`` is indexed 1..k."""
    pf = prodFactor(q, k)
    out = []
    i = 1
    while len(out) < threshold and i <= k:
        if correct[i]:
            res, t, l = pf, 0, 1
            while t < threshold and l <= k:
                if correct[l]:
                    if l != i:
                        res = res * l % q * pow((l - i) % q, -1, q) % q
                    t += 1
                l += 1
            alt = res - q
            out.append(alt if abs(alt) < res else res)
        i += 1
    if len(out) < threshold:
        raise ValueError("ProtocolError: attempting to combine too few decryption factors")
    return out


def decryptionFactors(u, secretKey: int, q: int, k: int):
    """``firstComponents.exp(secretKey.neg().mul(inverseFactor))`` (DistrElGamalSession.java:384-385)."""
    inverseFactor = pow(prodFactor(q, k), -1, q)
    return u.exp((-secretKey) * inverseFactor % q)


def combineDecryptionFactors(decryptionFactors, correct: Sequence[bool], k: int, threshold: int, q: int):
    """``pGroup.expProd(bases, integers, bitLength)``: out[i] = prod_j bases[j][i]^(integers[j]) (:465-503).
    Negative integers go through one batch inversion of the product of the negative part."""
    bases = [decryptionFactors[i] for i in range(1, k + 1) if correct[i]][:threshold]
    integers = modifiedLagrangeCoefficients(q, correct, k, threshold)
    pos = neg = None
    for base, c in zip(bases, integers):
        if c == 0:
            continue
        t = base.exp(abs(c))
        if c > 0:
            pos, old = (t, None) if pos is None else (pos.mul(t), pos)
        else:
            neg, old = (t, None) if neg is None else (neg.mul(t), neg)
        if old is not None:
            old.free()
            t.free()
    if neg is not None:
        ninv = neg.inv()
        neg.free()
        if pos is None:
            return ninv
        out = pos.mul(ninv)
        pos.free()
        ninv.free()
        return out
    return pos


def plaintexts(v, combinedFactors):
    """``v.mul(combinedFactors)`` (DistrElGamalSession.java:536-538)."""
    return v.mul(combinedFactors)


class DistrElGamalSessionBasic:
    """Batched Chaum–Pedersen proof of correct decryption factors (one instance per party j)."""

    def __init__(self, group, j: int, k: int, threshold: int, ebitlen: int, rand=None):
        self.G, self.j, self.k, self.threshold, self.ebitlen, self.rand = group, j, k, threshold, ebitlen, rand
        self.p, self.q, self.g = group.p, group.q, group.g
        # single elements through the group object (ModPGroup integers or ECqPGroup points)
        self._mul, self._exp, self._inv = group.k_mul, group.k_exp, group.k_inv
        self.inverseFactor = pow(prodFactor(self.q, k), -1, self.q)
        self.yp, self.Bp, self.k_x, self.B = {}, {}, {}, {}

    def setInstance(self, u, y: Sequence[int], f):
        """u: first components; y[l]: public key shares g^(x_l) (1-indexed list); f[l]: decryption factor arrays."""
        self.u, self.y, self.f = u, y, f

    def setBatchVector(self, e_ints):
        self.e = self.G.ringArray(e_ints if isinstance(e_ints, (bytes, bytearray)) else list(e_ints))
        self.e_bits = min(self.ebitlen, self.q.bit_length())

    def batchInput(self):
        self.A = self.u.expProd(self.e, self.e_bits)           # :524-526

    def commit(self, x: int):
        """:534-540 (prover j).  x = this party's secret share."""
        self.x = x
        self.r = self.rand.ring_element()
        self.yp[self.j] = self._exp(self.g, self.r)
        self.Bp[self.j] = self._exp(self.A, self.r)
        return self.yp[self.j], self.Bp[self.j]

    def reply(self, v: int) -> int:
        """:595-598: k_x = -x * inverseFactor * v + r."""
        q = self.q
        self.k_x[self.j] = ((-self.x) * self.inverseFactor % q * (v % q) + self.r) % q
        return self.k_x[self.j]

    def setCommitment(self, l: int, yp: int, Bp: int):
        self.yp[l], self.Bp[l] = yp, Bp

    def setReply(self, l: int, k_x: int):
        self.k_x[l] = k_x

    def batch(self, l: int):
        self.B[l] = self.f[l].expProd(self.e, self.e_bits)      # :707-709

    def verify(self, l: int, v: int) -> bool:
        """:718-727."""
        q = self.q
        lhs1 = self._mul(self._exp(self._inv(self.y[l]), self.inverseFactor * (v % q) % q), self.yp[l])
        ok1 = lhs1 == self._exp(self.g, self.k_x[l])
        ok2 = self._mul(self._exp(self.B[l], v % q), self.Bp[l]) == self._exp(self.A, self.k_x[l])
        return ok1 and ok2

    def combine(self, correct: Sequence[bool], combinedy: int, combinedf):
        """:642-678 plus the inputs of verifyCombined (combined public key and combined factors)."""
        q = self.q
        ints = modifiedLagrangeCoefficients(q, correct, self.k, self.threshold)
        exps = [c % q for c in ints]
        self.combinedyp = self.combinedBp = self.G.ONE
        self.combinedk_x = 0
        t = 0
        for l in range(1, self.k + 1):
            if t >= self.threshold:
                break
            if correct[l]:
                self.combinedyp = self._mul(self.combinedyp, self._exp(self.yp[l], exps[t]))
                self.combinedBp = self._mul(self.combinedBp, self._exp(self.Bp[l], exps[t]))
                self.combinedk_x = (self.combinedk_x + self.k_x[l] * exps[t]) % q
                t += 1
        self.combinedy, self.combinedf = combinedy, combinedf

    def batchCombined(self):
        self.combinedB = self.combinedf.expProd(self.e, self.e_bits)     # :683-685

    def verifyCombined(self, v: int) -> bool:
        """:693-700."""
        q = self.q
        ok1 = self._mul(self._exp(self._inv(self.combinedy), v % q), self.combinedyp) == self._exp(self.g, self.combinedk_x)
        ok2 = self._mul(self._exp(self.combinedB, v % q), self.combinedBp) == self._exp(self.A, self.combinedk_x)
        return ok1 and ok2
