"""hvzk.py — the sigma-protocol cores of the reference, issuing the same sequence of array
operations against the HIP library.

Mirrors (same method names, same operation order, same message layout):
  * ``PoSBasicTW``   — proof of a shuffle (Terelius–Wikström),
                       ref: src/java/com/verificatum/protocol/hvzk/PoSBasicTW.java
                       (precompute :436-482, setBatchVector :533-538, commit :546-700,
                       reply :856-888, computeAF :407-410, verify :1000-1066)
  * ``PoSCBasicTW``  — proof of a shuffle of commitments, ref: hvzk/PoSCBasicTW.java
                       (commit :363-529, reply :607-636, verify :646-727, short-circuiting)
  * ``CCPoSBasicW``  — commitment-consistent proof of a shuffle, ref: hvzk/CCPoSBasicW.java
                       (commit :344-396, reply :462-485, computeAB :493-506, verify :519-584)

Every per-element operation runs on the GPU through the C ABI (``PGroupElementArray`` /
``PRingElementArray`` of this package).  What stays on the host are the O(1) scalars of a proof
(A', C', D', k_A … and the final equality checks on single group elements), exactly the part
that stays in Java/VCR scalar classes in the reference.

Differences from the reference, all at the edges and none in the arithmetic:
  * the batching vector ``e`` and the prover's random values are *inputs* (``setBatchVector(e)``,
    a ``rand`` source object): VCR's PRG / ``randomElementArray`` sampling is not part of the
    reference tree, so the random tape is explicit;
  * messages are Python dicts of arrays / ints instead of byte trees (byte-tree framing is a
    "next" row).

A ciphertext array of width ω is a list of 2ω component arrays ``[u_1..u_ω, v_1..v_ω]``
(struct of arrays, the way ``PPGroupElementArray.project`` exposes it,
ref: src/java/com/verificatum/protocol/elgamal/DistrElGamalSession.java:377-378), the wide public
key the matching list of scalars ``[g..g, y..y]`` (ProtocolElGamal.java:785-800).
"""
from __future__ import annotations

import sys
from typing import List, Optional, Sequence


def _inv_perm(pi):
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


def _is_bytes(x) -> bool:
    """A block of big-endian rows (bytes, or a host buffer object such as a pinned tensor) rather than integers."""
    return isinstance(x, (bytes, bytearray)) or hasattr(x, "data_ptr") or hasattr(x, "ctypes")


class _Base:
    def __init__(self, group, vbitlen: int, ebitlen: int, rbitlen: int, rand=None):
        self.G = group
        self.p, self.q = group.p, group.q
        self.vbitlen, self.ebitlen, self.rbitlen = vbitlen, ebitlen, rbitlen
        self.rand = rand
        qbits = self.q.bit_length()
        self.e_bits = min(ebitlen, qbits)
        self.eps_bits = min(ebitlen + vbitlen + rbitlen, qbits)

    @staticmethod
    def _received_bits(k_E) -> int:
        """Bit length to use for an exponent array that came in a message: every bit of it counts (the reference parses
        full field elements, PoSBasicTW.java:985-989), so it is measured on the GPU rather than assumed."""
        return max(1, k_E.maxBits())

    # scalar helpers (single group elements on the host, as VCR's scalar classes): through the group object, so
    # the same driver serves ModPGroup (integers) and ECqPGroup (affine points)
    def _gexp(self, base, e: int):
        return self.G.k_exp(base, e)

    def _div(self, a, b):
        return self.G.k_mul(a, self.G.k_inv(b))

    def _expmul(self, a, v: int, b):
        """``a.expMul(v, b)`` = a^v * b (PoSBasicTW.java:1016-1021)."""
        return self.G.k_mul(self.G.k_exp(a, v), b)

    def setBatchVectorSeed(self, seed: bytes):
        """``setBatchVector(byte[] prgSeed)`` (PoSBasicTW.java:533-538): prg.setSeed(seed); e = N integers of ebitlen
        bits from the PRG, derived on the GPU."""
        self.e = self.G.ringArrayFromPRG(seed, self.size, self.ebitlen)

    def _eps_array(self):
        """epsilon: N integers of ebitlen+vbitlen+rbitlen bits, as field elements (PoSBasicTW.java:470-475).
        A random source may hand out big-endian bytes of the group's wire width directly (bulk path)."""
        eps = self.rand.int_array(self.size, self.ebitlen + self.vbitlen + self.rbitlen)
        return self.G.ringArray(eps if _is_bytes(eps) else [x % self.q for x in eps])

    def _ciph_expprod(self, w, E, ebits) -> list:
        """``w.expProd(E)`` of a ciphertext array = one multi-exponentiation per component, exponents sorted once."""
        multi = getattr(sys.modules.get("verificatum_vmn_amd"), "expProdMulti", None)
        if multi is not None and hasattr(w[0], "_h"):
            return multi(list(w), E, ebits)
        return [c.expProd(E, ebits) for c in w]


class PoSBasicTW(_Base):
    """ref: hvzk/PoSBasicTW.java"""

    # ---- both ---------------------------------------------------------------------------------
    def precompute(self, g: int, h, pi: Optional[Sequence[int]] = None):
        """VERIFIER (pi is None) :394-402 / PROVER :436-482."""
        self.size = h.size()
        self.g, self.h = g, h
        if pi is None:
            return
        G = self.G
        self.pi = pi
        # u_i = g^{r_pi(i)} * h_pi(i)
        self.r = G.ringArray(self.rand.ring_array(self.size))
        tmp1 = G.exp(g, self.r)
        tmp2 = h.mul(tmp1)
        tmp1.free()
        self.u = tmp2.permute(self.pi)
        tmp2.free()
        self.alpha = self.rand.ring_element()
        self.epsilon = self._eps_array()
        # A' = g^alpha * prod h_i^eps_i
        self.Ap = self.G.k_mul(self._gexp(g, self.alpha), h.expProd(self.epsilon, self.eps_bits))

    def setInstance(self, pkey: Sequence[int], w, wp, s=None):
        """:421-433 (verifier) / prover variant with the re-encryption exponents s (list of ω ring arrays)."""
        self.pkey, self.w, self.wp, self.s = list(pkey), w, wp, s

    def setPermutationCommitment(self, u):
        self.u = u

    def setBatchVector(self, e_ints: Sequence[int]):
        """:533-538 — e = first N values of ebitlen bits from the PRG (the values are the input here)."""
        self.e = self.G.ringArray(e_ints if _is_bytes(e_ints) else list(e_ints))

    # ---- prover -------------------------------------------------------------------------------
    def commit(self):
        """:546-700.  Returns (B, A', B', C', D', F')."""
        G, g, h, q, p = self.G, self.g, self.h, self.q, self.p
        piinv = _inv_perm(self.pi)
        self.ipe = self.e.permute(piinv)
        h0 = h.get(0)
        self.b = G.ringArray(self.rand.ring_array(self.size))
        x, self.d = self.b.recLin(self.ipe)
        y = self.ipe.prods()
        g_exp_x = G.exp(g, x)
        h0_exp_y = G.exp(h0, y)
        self.B = g_exp_x.mul(h0_exp_y)
        g_exp_x.free()
        h0_exp_y.free()
        self.beta = G.ringArray(self.rand.ring_array(self.size))
        xp = x.shiftPush(0)
        yp = y.shiftPush(1)
        y.free()
        x.free()
        xp_mul_epsilon = xp.mul(self.epsilon)
        beta_add_prod = self.beta.add(xp_mul_epsilon)
        g_exp_beta_add_prod = G.exp(g, beta_add_prod)
        yp_mul_epsilon = yp.mul(self.epsilon)
        h0_exp_yp_mul_epsilon = G.exp(h0, yp_mul_epsilon)
        self.Bp = g_exp_beta_add_prod.mul(h0_exp_yp_mul_epsilon)
        for t in (h0_exp_yp_mul_epsilon, yp_mul_epsilon, g_exp_beta_add_prod, beta_add_prod, xp_mul_epsilon, yp, xp):
            t.free()
        self.gamma = self.rand.ring_element()
        self.Cp = self._gexp(g, self.gamma)
        self.delta = self.rand.ring_element()
        self.Dp = self._gexp(g, self.delta)
        width = len(self.pkey) // 2
        self.phi = [self.rand.ring_element() for _ in range(width)]      # ciphPRing element: one value per column
        prods = self._ciph_expprod(self.wp, self.epsilon, self.eps_bits)
        self.Fp = [G.k_mul(self._gexp(pk, -self.phi[c % width]), t) for c, (pk, t) in enumerate(zip(self.pkey, prods))]
        return {"B": self.B, "Ap": self.Ap, "Bp": self.Bp, "Cp": self.Cp, "Dp": self.Dp, "Fp": self.Fp}

    def setChallenge(self, v: int):
        self.v = int(v)

    def reply(self, v: int):
        """:856-888.  Returns (k_A, k_B, k_C, k_D, k_E, k_F)."""
        self.setChallenge(v)
        q = self.q
        a = self.r.innerProduct(self.ipe)
        c = self.r.sum()
        f = [si.innerProduct(self.e) for si in self.s]                   # product-ring inner product: per column
        self.k_A = (a * v + self.alpha) % q
        self.k_B = self.b.mulAdd(v % q, self.beta)
        self.k_C = (c * v + self.gamma) % q
        self.k_D = (self.d * v + self.delta) % q
        self.k_E = self.ipe.mulAdd(v % q, self.epsilon)
        self.k_F = [(fc * v + ph) % q for fc, ph in zip(f, self.phi)]
        return {"k_A": self.k_A, "k_B": self.k_B, "k_C": self.k_C, "k_D": self.k_D, "k_E": self.k_E, "k_F": self.k_F}

    # ---- verifier -----------------------------------------------------------------------------
    def computeAF(self):
        """:407-410."""
        res = self._ciph_expprod([self.u] + list(self.w), self.e, self.e_bits)      # one sort of e for u and w
        self.A, self.F = res[0], res[1:]

    def setCommitment(self, msg):
        """:780-823 (parsing is out of scope; the parsed objects are handed over)."""
        self.B, self.Ap, self.Bp = msg["B"], msg["Ap"], msg["Bp"]
        self.Cp, self.Dp, self.Fp = msg["Cp"], msg["Dp"], msg["Fp"]

    def verify(self, reply) -> bool:
        """:1000-1066 — all five checks are evaluated (no short-circuit)."""
        G, g, h, p, q, v = self.G, self.g, self.h, self.p, self.q, self.v
        k_A, k_B, k_C, k_D, k_E, k_F = (reply[k] for k in ("k_A", "k_B", "k_C", "k_D", "k_E", "k_F"))
        h0 = h.get(0)
        C = self._div(self.u.prod(), h.prod())
        D = self._div(self.B.get(self.size - 1), self._gexp(h0, self.e.prod()))
        kE_bits = self._received_bits(k_E)
        kE_prods = self._ciph_expprod([h] + list(self.wp), k_E, kE_bits)              # one sort of k_E for h and w'
        verdictA = self._expmul(self.A, v, self.Ap) == G.k_mul(self._gexp(g, k_A), kE_prods[0])
        B_exp_v = self.B.exp(v)
        leftSide = B_exp_v.mul(self.Bp)
        g_exp_k_B = G.exp(g, k_B)
        B_shift = self.B.shiftPush(h0)
        B_shift_exp_k_E = B_shift.exp(k_E, kE_bits)
        rightSide = g_exp_k_B.mul(B_shift_exp_k_E)
        verdictB = leftSide.equals(rightSide)
        for t in (B_exp_v, leftSide, g_exp_k_B, B_shift, B_shift_exp_k_E, rightSide):
            t.free()
        verdictC = self._expmul(C, v, self.Cp) == self._gexp(g, k_C)
        verdictD = self._expmul(D, v, self.Dp) == self._gexp(g, k_D)
        prods = kE_prods[1:]
        width = len(self.pkey) // 2
        verdictF = all(self._expmul(Fc, v, Fpc) == G.k_mul(self._gexp(pk, -k_F[c % width]), t)
                       for c, (Fc, Fpc, pk, t) in enumerate(zip(self.F, self.Fp, self.pkey, prods)))
        self.verdicts = (verdictA, verdictB, verdictC, verdictD, verdictF)
        return all(self.verdicts)


class PoSCBasicTW(_Base):
    """ref: hvzk/PoSCBasicTW.java"""

    def setInstance(self, g: int, h, u, r=None, pi: Optional[Sequence[int]] = None):
        """:306-340."""
        self.g, self.h, self.u, self.r = g, h, u, r
        self.pi = pi
        self.size = h.size()

    def setBatchVector(self, e_ints: Sequence[int]):
        self.e = self.G.ringArray(e_ints if _is_bytes(e_ints) else list(e_ints))

    def commit(self):
        """:363-529.  Returns (B, A', B', C', D')."""
        G, g, h = self.G, self.g, self.h
        self.ipe = self.e.permute(_inv_perm(self.pi))
        h0 = h.get(0)
        self.b = G.ringArray(self.rand.ring_array(self.size))
        x, self.d = self.b.recLin(self.ipe)
        y = self.ipe.prods()
        g_exp_x = G.exp(g, x)
        h0_exp_y = G.exp(h0, y)
        self.B = g_exp_x.mul(h0_exp_y)
        g_exp_x.free()
        h0_exp_y.free()
        self.alpha = self.rand.ring_element()
        self.epsilon = self._eps_array()
        self.Ap = self.G.k_mul(self._gexp(g, self.alpha), h.expProd(self.epsilon, self.eps_bits))
        self.beta = G.ringArray(self.rand.ring_array(self.size))
        xp = x.shiftPush(0)
        yp = y.shiftPush(1)
        xp_mul_epsilon = xp.mul(self.epsilon)
        beta_add_prod = self.beta.add(xp_mul_epsilon)
        g_exp_beta_add_prod = G.exp(g, beta_add_prod)
        yp_mul_epsilon = yp.mul(self.epsilon)
        h0_exp_yp_mul_epsilon = G.exp(h0, yp_mul_epsilon)
        self.Bp = g_exp_beta_add_prod.mul(h0_exp_yp_mul_epsilon)
        for t in (x, y, xp, yp, xp_mul_epsilon, beta_add_prod, g_exp_beta_add_prod, yp_mul_epsilon, h0_exp_yp_mul_epsilon):
            t.free()
        self.gamma = self.rand.ring_element()
        self.Cp = self._gexp(g, self.gamma)
        self.delta = self.rand.ring_element()
        self.Dp = self._gexp(g, self.delta)
        return {"B": self.B, "Ap": self.Ap, "Bp": self.Bp, "Cp": self.Cp, "Dp": self.Dp}

    def setChallenge(self, v: int):
        self.v = int(v)

    def reply(self, v: int):
        """:607-636."""
        self.setChallenge(v)
        q = self.q
        a = self.r.innerProduct(self.ipe)
        c = self.r.sum()
        self.k_A = (a * v + self.alpha) % q
        self.k_B = self.b.mulAdd(v % q, self.beta)
        self.k_C = (c * v + self.gamma) % q
        self.k_D = (self.d * v + self.delta) % q
        self.k_E = self.ipe.mulAdd(v % q, self.epsilon)
        return {"k_A": self.k_A, "k_B": self.k_B, "k_C": self.k_C, "k_D": self.k_D, "k_E": self.k_E}

    def setCommitment(self, msg):
        self.B, self.Ap, self.Bp, self.Cp, self.Dp = msg["B"], msg["Ap"], msg["Bp"], msg["Cp"], msg["Dp"]

    def verify(self, reply) -> bool:
        """:646-727 — short-circuits after the first failing check."""
        G, g, h, p, v = self.G, self.g, self.h, self.p, self.v
        k_A, k_B, k_C, k_D, k_E = (reply[k] for k in ("k_A", "k_B", "k_C", "k_D", "k_E"))
        h0 = h.get(0)
        A = self.u.expProd(self.e, self.e_bits)
        C = self._div(self.u.prod(), h.prod())
        D = self._div(self.B.get(self.size - 1), self._gexp(h0, self.e.prod()))
        kE_bits = self._received_bits(k_E)
        if self._expmul(A, v, self.Ap) != G.k_mul(self._gexp(g, k_A), h.expProd(k_E, kE_bits)):
            return False
        B_exp_v = self.B.exp(v)
        leftSide = B_exp_v.mul(self.Bp)
        g_exp_k_B = G.exp(g, k_B)
        B_shift = self.B.shiftPush(h0)
        B_shift_exp_k_E = B_shift.exp(k_E, kE_bits)
        rightSide = g_exp_k_B.mul(B_shift_exp_k_E)
        B_res = leftSide.equals(rightSide)
        for t in (B_exp_v, leftSide, g_exp_k_B, B_shift, B_shift_exp_k_E, rightSide):
            t.free()
        if not B_res:
            return False
        if self._expmul(C, v, self.Cp) != self._gexp(g, k_C):
            return False
        if self._expmul(D, v, self.Dp) != self._gexp(g, k_D):
            return False
        return True


class CCPoSBasicW(_Base):
    """ref: hvzk/CCPoSBasicW.java"""

    def setInstance(self, g: int, h, u, pkey: Sequence[int], w, wp, r=None, pi=None, s=None):
        self.g, self.h, self.u, self.pkey, self.w, self.wp = g, h, u, list(pkey), w, wp
        self.r, self.s = r, s
        self.pi = pi
        self.size = h.size()

    def setBatchVector(self, e_ints: Sequence[int]):
        self.e = self.G.ringArray(e_ints if _is_bytes(e_ints) else list(e_ints))

    def commit(self):
        """:344-396.  Returns (A', B')."""
        G, g, h, p = self.G, self.g, self.h, self.p
        self.ipe = self.e.permute(_inv_perm(self.pi))
        self.alpha = self.rand.ring_element()
        self.epsilon = self._eps_array()
        eps_prods = self._ciph_expprod([h] + list(self.wp), self.epsilon, self.eps_bits)   # one sort of epsilon
        self.Ap = G.k_mul(self._gexp(g, self.alpha), eps_prods[0])
        width = len(self.pkey) // 2
        self.beta = [self.rand.ring_element() for _ in range(width)]
        prods = eps_prods[1:]
        self.Bp = [G.k_mul(self._gexp(pk, -self.beta[c % width]), t) for c, (pk, t) in enumerate(zip(self.pkey, prods))]
        return {"Ap": self.Ap, "Bp": self.Bp}

    def setChallenge(self, v: int):
        self.v = int(v)

    def reply(self, v: int):
        """:462-485."""
        self.setChallenge(v)
        q = self.q
        a = self.r.innerProduct(self.ipe)
        b = [si.innerProduct(self.e) for si in self.s]
        self.k_A = (a * v + self.alpha) % q
        self.k_B = [(bc * v + bt) % q for bc, bt in zip(b, self.beta)]
        self.k_E = self.ipe.mulAdd(v % q, self.epsilon)
        return {"k_A": self.k_A, "k_B": self.k_B, "k_E": self.k_E}

    def setCommitment(self, msg):
        self.Ap, self.Bp = msg["Ap"], msg["Bp"]

    def computeAB(self, raisedu=None):
        """:493-506.  raisedu = u^rho selects the single-equation ("raised") form."""
        if raisedu is None:
            res = self._ciph_expprod([self.u] + list(self.w), self.e, self.e_bits)
            self.A, self.B = res[0], res[1:]
        else:
            self.AB = []
            for c in self.w:                       # w.mul(raisedu): the base-group array multiplies every component
                tmp = c.mul(raisedu)
                self.AB.append(tmp.expProd(self.e, self.e_bits))
                tmp.free()

    def verify(self, reply, raisedh=None, raisedExponent: Optional[int] = None) -> bool:
        """:519-584."""
        g, h, p, v = self.g, self.h, self.p, self.v
        k_A, k_B, k_E = reply["k_A"], reply["k_B"], reply["k_E"]
        if raisedExponent is None:
            kE_prods = self._ciph_expprod([h] + list(self.wp), k_E, self._received_bits(k_E))
            if self._expmul(self.A, v, self.Ap) != self.G.k_mul(self._gexp(g, k_A), kE_prods[0]):
                return False
            prods = kE_prods[1:]
            width = len(self.pkey) // 2
            return all(self._expmul(Bc, v, Bpc) == self.G.k_mul(self._gexp(pk, -k_B[c % width]), t)
                       for c, (Bc, Bpc, pk, t) in enumerate(zip(self.B, self.Bp, self.pkey, prods)))
        rho = raisedExponent
        KG = self.G
        Ap_rho = KG.k_exp(self.Ap, rho)
        g_term = self._gexp(g, k_A * rho)
        ok = True
        width = len(self.pkey) // 2
        for c, (ABc, Bpc, pk, col) in enumerate(zip(self.AB, self.Bp, self.pkey, self.wp)):
            wp_mul_raisedh = col.mul(raisedh)
            t = wp_mul_raisedh.expProd(k_E, self._received_bits(k_E))
            wp_mul_raisedh.free()
            lhs = self._expmul(ABc, v, KG.k_mul(Bpc, Ap_rho))
            rhs = KG.k_mul(KG.k_mul(self._gexp(pk, -k_B[c % width]), t), g_term)
            ok = ok and lhs == rhs
        return ok


class IndependentGeneratorsBasicI:
    """ref: protocol/distr/IndependentGeneratorsBasicI.java (interactive derivation of independent generators,
    SURVEY.md §8a row A7): setInstance :166-175, setBatchVector :186-193, commit :201-208, reply :245-248,
    verify() :275-289, verify(l) :297-299.  Parties are numbered 1..threshold."""

    def __init__(self, group, j: int, threshold: int, ebitlen: int, rand=None):
        self.G, self.j, self.threshold, self.ebitlen, self.rand = group, j, threshold, ebitlen, rand
        self.q = group.q
        self.e_bits = min(ebitlen, self.q.bit_length())
        self.Ap, self.k_a = {}, {}

    def setInstance(self, g, h, s, combinedh):
        self.g, self.h, self.s, self.combinedh = g, h, s, combinedh

    def setBatchVector(self, e_ints):
        self.e = self.G.ringArray(e_ints if _is_bytes(e_ints) else list(e_ints))

    def setBatchVectorSeed(self, seed: bytes):
        self.e = self.G.ringArrayFromPRG(seed, self.combinedh.size(), self.ebitlen)

    def commit(self):
        self.a = self.s.innerProduct(self.e)                  # :202
        self.r = self.rand.ring_element()
        self.Ap[self.j] = self.G.k_exp(self.g, self.r)        # :205
        return self.Ap[self.j]

    def setCommitment(self, l: int, Ap):
        self.Ap[l] = Ap

    def setChallenge(self, v: int):
        self.v = int(v) % self.q

    def reply(self) -> int:
        self.k_a[self.j] = (self.a * self.v + self.r) % self.q    # :246
        return self.k_a[self.j]

    def setReply(self, l: int, k_a: int):
        self.k_a[l] = k_a if 0 <= k_a < self.q else 0

    def verify(self, l: Optional[int] = None) -> bool:
        G = self.G
        if l is not None:                                         # :297-299
            A = self.h[l].expProd(self.e, self.e_bits)
            return G.k_mul(G.k_exp(A, self.v), self.Ap[l]) == G.k_exp(self.g, self.k_a[l])
        k, Ap = 0, G.ONE                                          # :275-289
        for p in range(1, self.threshold + 1):
            k = (k + self.k_a[p]) % self.q
            Ap = G.k_mul(Ap, self.Ap[p])
        A = self.combinedh.expProd(self.e, self.e_bits)
        return G.k_mul(G.k_exp(A, self.v), Ap) == G.k_exp(self.g, k)
