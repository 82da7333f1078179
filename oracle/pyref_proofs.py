"""pyref_proofs.py — Python-integer restatement of the reference's sigma-protocol cores and of the
shuffler arithmetic.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows, statement by statement (P/ = /root/reference/src/java/com/verificatum/protocol):
  PoS    P/hvzk/PoSBasicTW.java:436-482 (precompute), 546-700 (commit), 856-888 (reply),
         407-410 (computeAF), 1000-1066 (verify, all five checks evaluated)
  PoSC   P/hvzk/PoSCBasicTW.java:363-529, 607-636, 646-727 (short-circuit)
  CCPoS  P/hvzk/CCPoSBasicW.java:344-396, 462-485, 493-506, 519-584 (plain and "raised" form)
  shuffle P/mixnet/ShufflerElGamalSession.java:400-409, 273-278, 498-507
  permutation commitment P/mixnet/PermutationCommitment.java:189-215, 357

Everything is a list of Python ints; exponentiation is the built-in pow.  The random values are
drawn from the ``rand`` object in the same order as the reference draws them from its
RandomSource (r, alpha, epsilon | b, beta, gamma, delta, phi), so that a product run fed with the
same tape must produce identical messages.

Parity status: the reference's own unit test for this layer asserts only accept / reject
(P/hvzk/TestPoSCBasicTW.java:147-163); it holds no transcript vectors.  "parity unpinned" by
reference fixtures; pinned by: honest transcript verifies, tampered witness is rejected
(the reference's own negative case r <- r + r, TestPoSCBasicTW.java:109-111), and message-for-message
equality between this restatement and the HIP path on the same tape.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

from . import pyref


def inv_perm(pi: Sequence[int]) -> List[int]:
    inv = [0] * len(pi)
    for i, j in enumerate(pi):
        inv[j] = i
    return inv


class ModPAdapter:
    """Group operations on Python ints mod p (multiplicative group)."""

    def __init__(self, p: int, q: int):
        self.p, self.q, self.one = p, q, 1

    def mul(self, a, b):
        return a * b % self.p

    def exp(self, a, e):
        return pow(a, e % self.q, self.p)

    def inv(self, a):
        return pow(a, -1, self.p)

    def exp_fixed(self, base, es):
        return pyref.exp_fixed(base, es, self.p)

    def exp_array(self, xs, es):
        return pyref.exp_array(xs, es, self.p)

    def exp_scalar(self, xs, e):
        return pyref.exp_scalar(xs, e, self.p)

    def exp_prod(self, xs, es):
        return pyref.exp_prod(xs, es, self.p)

    def mul_arrays(self, xs, ys):
        return pyref.mul(xs, ys, self.p)

    def prod(self, xs):
        return pyref.prod(xs, self.p)


class ECAdapter:
    """The same interface over an elliptic curve (oracle/pyref_ec.Curve): mul = point addition, exp = scalar
    multiplication."""

    def __init__(self, curve):
        self.c, self.q, self.one = curve, curve.n, None

    def mul(self, a, b):
        return self.c.add(a, b)

    def exp(self, a, e):
        return self.c.mul(e % self.q, a)

    def inv(self, a):
        return self.c.neg(a)

    def exp_fixed(self, base, es):
        return self.c.exp_fixed(base, es)

    def exp_array(self, xs, es):
        return self.c.exp_array(xs, es)

    def exp_scalar(self, xs, e):
        return [self.c.mul(e, P) for P in xs]

    def exp_prod(self, xs, es):
        return self.c.exp_prod(xs, es)

    def mul_arrays(self, xs, ys):
        return self.c.mul_arrays(xs, ys)

    def prod(self, xs):
        return self.c.prod(xs)


def reenc_factors(pkey: Sequence[int], s_cols: Sequence[Sequence[int]], p: int):
    width = len(pkey) // 2
    return [pyref.exp_fixed(pk, s_cols[c % width], p) for c, pk in enumerate(pkey)]


def reencrypt(w: Sequence[Sequence[int]], factors, permutation: Sequence[int], p: int):
    inverse = inv_perm(permutation)
    return [pyref.permute(pyref.mul(c, f, p), inverse) for c, f in zip(w, factors)]


def permutation_commitment(g: int, generators: Sequence[int], exponents: Sequence[int], permutation, p: int):
    ident = pyref.mul(generators, pyref.exp_fixed(g, exponents, p), p)
    return pyref.permute(ident, permutation)


def shrink_permutation(permutation: Sequence[int], n: int):
    """PermutationCommitment.shrink, P/mixnet/PermutationCommitment.java:390-471: (keep list, compressed permutation).
    The kept positions of the commitment are those that commit to the first n generators -- the exponents and the
    generators are cut to [0, n) (:415, ShufflerElGamalSession.java:684-703) -- which in the gather convention of
    pyref.permute (out[i] = X[perm[i]]) is perm[i] < n; the reference writes keepList[permutation.map(i)] = true for
    i < n (:398-405), the same list when perm is the table of its permutation's inverse."""
    keep = [src < n for src in permutation]
    return keep, [src for src in permutation if src < n]


def extract(xs: Sequence, keep: Sequence[bool]) -> list:
    """``array.extract(boolean[])``: the elements whose flag is set, order preserved (:462-469)."""
    return [x for x, k in zip(xs, keep) if k]


def sanitize_keep_list(keep: Sequence[bool], n_max: int, n: int):
    """:424-447: a received keep list is used only if it has n_max flags of which exactly n are set."""
    keep = list(keep)
    if len(keep) != n_max or sum(1 for k in keep if k) != n:
        return [i < n for i in range(n_max)]
    return keep


class _Base:
    def __init__(self, p, q, vbitlen, ebitlen, rbitlen, rand=None):
        self.p, self.q = p, q
        self.vbitlen, self.ebitlen, self.rbitlen = vbitlen, ebitlen, rbitlen
        self.rand = rand

    def _ciph_expprod(self, w, E):
        return [pyref.exp_prod(c, E, self.p) for c in w]

    def _div(self, a, b):
        return a * pow(b, -1, self.p) % self.p


class PoS(_Base):
    def precompute(self, g, h, pi=None):
        self.size, self.g, self.h = len(h), g, list(h)
        if pi is None:
            return
        p, q = self.p, self.q
        self.pi = list(pi)
        self.r = self.rand.ring_array(self.size)
        self.u = pyref.permute(pyref.mul(h, pyref.exp_fixed(g, self.r, p), p), self.pi)
        self.alpha = self.rand.ring_element()
        self.epsilon = [x % q for x in self.rand.int_array(self.size, self.ebitlen + self.vbitlen + self.rbitlen)]
        self.Ap = pow(g, self.alpha, p) * pyref.exp_prod(h, self.epsilon, p) % p

    def setInstance(self, pkey, w, wp, s=None):
        self.pkey, self.w, self.wp, self.s = list(pkey), w, wp, s

    def setBatchVector(self, e):
        self.e = list(e)

    def commit(self):
        p, q, g, h = self.p, self.q, self.g, self.h
        self.ipe = pyref.permute(self.e, inv_perm(self.pi))
        h0 = h[0]
        self.b = self.rand.ring_array(self.size)
        x, self.d = pyref.rec_lin(self.b, self.ipe, q)
        y = pyref.prods(self.ipe, q)
        self.B = pyref.mul(pyref.exp_fixed(g, x, p), pyref.exp_fixed(h0, y, p), p)
        self.beta = self.rand.ring_array(self.size)
        xp = pyref.shift_push(x, 0)
        yp = pyref.shift_push(y, 1)
        beta_add_prod = [(bt + a * ep) % q for bt, a, ep in zip(self.beta, xp, self.epsilon)]
        yp_mul_epsilon = [a * ep % q for a, ep in zip(yp, self.epsilon)]
        self.Bp = pyref.mul(pyref.exp_fixed(g, beta_add_prod, p), pyref.exp_fixed(h0, yp_mul_epsilon, p), p)
        self.gamma = self.rand.ring_element()
        self.Cp = pow(g, self.gamma, p)
        self.delta = self.rand.ring_element()
        self.Dp = pow(g, self.delta, p)
        width = len(self.pkey) // 2
        self.phi = [self.rand.ring_element() for _ in range(width)]      # element of the product ring R^width
        self.Fp = [pow(pk, (-self.phi[c % width]) % q, p) * t % p
                   for c, (pk, t) in enumerate(zip(self.pkey, self._ciph_expprod(self.wp, self.epsilon)))]
        return {"B": self.B, "Ap": self.Ap, "Bp": self.Bp, "Cp": self.Cp, "Dp": self.Dp, "Fp": self.Fp}

    def reply(self, v):
        q = self.q
        self.v = v
        a = pyref.inner_product(self.r, self.ipe, q)
        c = sum(self.r) % q
        f = [pyref.inner_product(si, self.e, q) for si in self.s]       # one value per column
        return {"k_A": (a * v + self.alpha) % q,
                "k_B": pyref.mul_add(self.b, v % q, self.beta, q),
                "k_C": (c * v + self.gamma) % q,
                "k_D": (self.d * v + self.delta) % q,
                "k_E": pyref.mul_add(self.ipe, v % q, self.epsilon, q),
                "k_F": [(fc * v + ph) % q for fc, ph in zip(f, self.phi)]}

    def computeAF(self):
        self.A = pyref.exp_prod(self.u, self.e, self.p)
        self.F = self._ciph_expprod(self.w, self.e)

    def setCommitment(self, msg):
        self.B, self.Ap, self.Bp = msg["B"], msg["Ap"], msg["Bp"]
        self.Cp, self.Dp, self.Fp = msg["Cp"], msg["Dp"], msg["Fp"]

    def verify(self, reply, v):
        p, q, g, h = self.p, self.q, self.g, self.h
        k_A, k_B, k_C, k_D, k_E, k_F = (reply[k] for k in ("k_A", "k_B", "k_C", "k_D", "k_E", "k_F"))
        h0 = h[0]
        C = self._div(pyref.prod(self.u, p), pyref.prod(h, p))
        eprod = 1
        for t in self.e:
            eprod = eprod * t % q
        D = self._div(self.B[self.size - 1], pow(h0, eprod, p))
        self.C, self.D = C, D            # the verifier's intermediates (getC / getD, PoSBasicTW.java:949, 958)
        verdictA = (pow(self.A, v, p) * self.Ap % p) == (pow(g, k_A, p) * pyref.exp_prod(h, k_E, p) % p)
        left = pyref.mul(pyref.exp_scalar(self.B, v, p), self.Bp, p)
        right = pyref.mul(pyref.exp_fixed(g, k_B, p), pyref.exp_array(pyref.shift_push(self.B, h0), k_E, p), p)
        verdictB = left == right
        verdictC = (pow(C, v, p) * self.Cp % p) == pow(g, k_C, p)
        verdictD = (pow(D, v, p) * self.Dp % p) == pow(g, k_D, p)
        prods = self._ciph_expprod(self.wp, k_E)
        width = len(self.pkey) // 2
        verdictF = all((pow(Fc, v, p) * Fpc % p) == (pow(pk, (-k_F[c % width]) % q, p) * t % p)
                       for c, (Fc, Fpc, pk, t) in enumerate(zip(self.F, self.Fp, self.pkey, prods)))
        self.verdicts = (verdictA, verdictB, verdictC, verdictD, verdictF)
        return all(self.verdicts)


class PoSC(_Base):
    def setInstance(self, g, h, u, r=None, pi=None):
        self.g, self.h, self.u, self.r = g, list(h), list(u), r
        self.pi = list(pi) if pi is not None else None
        self.size = len(h)

    def setBatchVector(self, e):
        self.e = list(e)

    def commit(self):
        p, q, g, h = self.p, self.q, self.g, self.h
        self.ipe = pyref.permute(self.e, inv_perm(self.pi))
        h0 = h[0]
        self.b = self.rand.ring_array(self.size)
        x, self.d = pyref.rec_lin(self.b, self.ipe, q)
        y = pyref.prods(self.ipe, q)
        self.B = pyref.mul(pyref.exp_fixed(g, x, p), pyref.exp_fixed(h0, y, p), p)
        self.alpha = self.rand.ring_element()
        self.epsilon = [t % q for t in self.rand.int_array(self.size, self.ebitlen + self.vbitlen + self.rbitlen)]
        self.Ap = pow(g, self.alpha, p) * pyref.exp_prod(h, self.epsilon, p) % p
        self.beta = self.rand.ring_array(self.size)
        xp = pyref.shift_push(x, 0)
        yp = pyref.shift_push(y, 1)
        beta_add_prod = [(bt + a * ep) % q for bt, a, ep in zip(self.beta, xp, self.epsilon)]
        yp_mul_epsilon = [a * ep % q for a, ep in zip(yp, self.epsilon)]
        self.Bp = pyref.mul(pyref.exp_fixed(g, beta_add_prod, p), pyref.exp_fixed(h0, yp_mul_epsilon, p), p)
        self.gamma = self.rand.ring_element()
        self.Cp = pow(g, self.gamma, p)
        self.delta = self.rand.ring_element()
        self.Dp = pow(g, self.delta, p)
        return {"B": self.B, "Ap": self.Ap, "Bp": self.Bp, "Cp": self.Cp, "Dp": self.Dp}

    def reply(self, v):
        q = self.q
        a = pyref.inner_product(self.r, self.ipe, q)
        c = sum(self.r) % q
        return {"k_A": (a * v + self.alpha) % q,
                "k_B": pyref.mul_add(self.b, v % q, self.beta, q),
                "k_C": (c * v + self.gamma) % q,
                "k_D": (self.d * v + self.delta) % q,
                "k_E": pyref.mul_add(self.ipe, v % q, self.epsilon, q)}

    def setCommitment(self, msg):
        self.B, self.Ap, self.Bp, self.Cp, self.Dp = msg["B"], msg["Ap"], msg["Bp"], msg["Cp"], msg["Dp"]

    def verify(self, reply, v):
        p, q, g, h = self.p, self.q, self.g, self.h
        k_A, k_B, k_C, k_D, k_E = (reply[k] for k in ("k_A", "k_B", "k_C", "k_D", "k_E"))
        h0 = h[0]
        A = self.A = pyref.exp_prod(self.u, self.e, p)
        C = self._div(pyref.prod(self.u, p), pyref.prod(h, p))
        eprod = 1
        for t in self.e:
            eprod = eprod * t % q
        D = self._div(self.B[self.size - 1], pow(h0, eprod, p))
        self.C, self.D = C, D            # the verifier's intermediates (getC / getD, PoSBasicTW.java:949, 958)
        if (pow(A, v, p) * self.Ap % p) != (pow(g, k_A, p) * pyref.exp_prod(h, k_E, p) % p):
            return False
        left = pyref.mul(pyref.exp_scalar(self.B, v, p), self.Bp, p)
        right = pyref.mul(pyref.exp_fixed(g, k_B, p), pyref.exp_array(pyref.shift_push(self.B, h0), k_E, p), p)
        if left != right:
            return False
        if (pow(C, v, p) * self.Cp % p) != pow(g, k_C, p):
            return False
        return (pow(D, v, p) * self.Dp % p) == pow(g, k_D, p)


class CCPoS(_Base):
    def setInstance(self, g, h, u, pkey, w, wp, r=None, pi=None, s=None):
        self.g, self.h, self.u, self.pkey, self.w, self.wp = g, list(h), list(u), list(pkey), w, wp
        self.r, self.s = r, s
        self.pi = list(pi) if pi is not None else None
        self.size = len(h)

    def setBatchVector(self, e):
        self.e = list(e)

    def commit(self):
        p, q, g, h = self.p, self.q, self.g, self.h
        self.ipe = pyref.permute(self.e, inv_perm(self.pi))
        self.alpha = self.rand.ring_element()
        self.epsilon = [t % q for t in self.rand.int_array(self.size, self.ebitlen + self.vbitlen + self.rbitlen)]
        self.Ap = pow(g, self.alpha, p) * pyref.exp_prod(h, self.epsilon, p) % p
        width = len(self.pkey) // 2
        self.beta = [self.rand.ring_element() for _ in range(width)]
        self.Bp = [pow(pk, (-self.beta[c % width]) % q, p) * t % p
                   for c, (pk, t) in enumerate(zip(self.pkey, self._ciph_expprod(self.wp, self.epsilon)))]
        return {"Ap": self.Ap, "Bp": self.Bp}

    def reply(self, v):
        q = self.q
        a = pyref.inner_product(self.r, self.ipe, q)
        b = [pyref.inner_product(si, self.e, q) for si in self.s]
        return {"k_A": (a * v + self.alpha) % q, "k_B": [(bc * v + bt) % q for bc, bt in zip(b, self.beta)],
                "k_E": pyref.mul_add(self.ipe, v % q, self.epsilon, q)}

    def setCommitment(self, msg):
        self.Ap, self.Bp = msg["Ap"], msg["Bp"]

    def computeAB(self, raisedu=None):
        p = self.p
        if raisedu is None:
            self.A = pyref.exp_prod(self.u, self.e, p)
            self.B = self._ciph_expprod(self.w, self.e)
        else:
            self.AB = [pyref.exp_prod(pyref.mul(c, raisedu, p), self.e, p) for c in self.w]

    def verify(self, reply, v, raisedh=None, raisedExponent: Optional[int] = None):
        p, q, g, h = self.p, self.q, self.g, self.h
        k_A, k_B, k_E = reply["k_A"], reply["k_B"], reply["k_E"]
        if raisedExponent is None:
            if (pow(self.A, v, p) * self.Ap % p) != (pow(g, k_A, p) * pyref.exp_prod(h, k_E, p) % p):
                return False
            prods = self._ciph_expprod(self.wp, k_E)
            width = len(self.pkey) // 2
            return all((pow(Bc, v, p) * Bpc % p) == (pow(pk, (-k_B[c % width]) % q, p) * t % p)
                       for c, (Bc, Bpc, pk, t) in enumerate(zip(self.B, self.Bp, self.pkey, prods)))
        rho = raisedExponent
        Ap_rho = pow(self.Ap, rho, p)
        g_term = pow(g, k_A * rho % q, p)
        ok = True
        width = len(self.pkey) // 2
        for c, (ABc, Bpc, pk, col) in enumerate(zip(self.AB, self.Bp, self.pkey, self.wp)):
            t = pyref.exp_prod(pyref.mul(col, raisedh, p), k_E, p)
            ok = ok and (pow(ABc, v, p) * (Bpc * Ap_rho % p) % p) == (pow(pk, (-k_B[c % width]) % q, p) * t % p * g_term % p)
        return ok


# ------------------------------------------------------------------------------------------------
# Verifiable threshold decryption (row A6).  P/elgamal/DistrElGamalSessionBasic.java:318-344 (prodFactor),
# :358-452 (modified Lagrange integers), :465-503 (combineDecryptionFactors), :524-526, :683-685, :707-709
# (batching), :693-700, :718-727 (checks); P/elgamal/DistrElGamalSession.java:365-385, :536-538.
# ------------------------------------------------------------------------------------------------
_ODD_PRIMES = [3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97]


def prod_factor(q: int, k: int) -> int:
    res, prime, i = 1, 2, 0
    while prime <= k:
        a = b = 1
        while b <= k:
            a, b = b, b * prime
        res *= a
        prime = _ODD_PRIMES[i]
        i += 1
    return res * res % q


def lagrange_integers(q: int, correct, k: int, threshold: int):
    pf = prod_factor(q, k)
    out = []
    for i in range(1, k + 1):
        if len(out) >= threshold:
            break
        if not correct[i]:
            continue
        res, t = pf, 0
        for l in range(1, k + 1):
            if t >= threshold:
                break
            if correct[l]:
                if l != i:
                    res = res * l * pow(l - i, -1, q) % q
                t += 1
        out.append(res - q if q - res < res else res)
    return out


def decryption_factors(u, x_j: int, p: int, q: int, k: int):
    return pyref.exp_scalar(u, (-x_j) * pow(prod_factor(q, k), -1, q) % q, p)


def combine_decryption_factors(factors, correct, k: int, threshold: int, p: int, q: int):
    bases = [factors[i] for i in range(1, k + 1) if correct[i]][:threshold]
    ints = lagrange_integers(q, correct, k, threshold)
    n = len(bases[0])
    out = [1] * n
    for base, c in zip(bases, ints):
        for i in range(n):
            out[i] = out[i] * pow(base[i], c, p) % p          # Python's pow handles negative exponents (inverse)
    return out


# ------------------------------------------------------------------------------------------------------
# Group-generic restatements (adapter K = ModPAdapter or ECAdapter): the same statements as PoS / PoSC / CCPoS
# above, written against the group interface so that they also cover ECqPGroup (the reference's code is
# group-agnostic: SURVEY.md §2.3 K11).  tests/test_proofs_oracle.py checks that over a ModPGroup they produce the
# same transcripts as the integer-only classes above.
# ------------------------------------------------------------------------------------------------------
def g_reenc_factors(K, pkey, s_cols):
    width = len(pkey) // 2
    return [K.exp_fixed(pk, s_cols[c % width]) for c, pk in enumerate(pkey)]


def g_reencrypt(K, w, factors, permutation):
    inverse = inv_perm(permutation)
    return [pyref.permute(K.mul_arrays(c, f), inverse) for c, f in zip(w, factors)]


def g_permutation_commitment(K, g, generators, exponents, permutation):
    return pyref.permute(K.mul_arrays(generators, K.exp_fixed(g, exponents)), permutation)


class GPoS:
    def __init__(self, K, vbitlen, ebitlen, rbitlen, rand=None):
        self.K, self.q = K, K.q
        self.vbitlen, self.ebitlen, self.rbitlen, self.rand = vbitlen, ebitlen, rbitlen, rand

    def _cx(self, w, E):
        return [self.K.exp_prod(c, E) for c in w]

    def precompute(self, g, h, pi=None):
        K, q = self.K, self.q
        self.size, self.g, self.h = len(h), g, list(h)
        if pi is None:
            return
        self.pi = list(pi)
        self.r = self.rand.ring_array(self.size)
        self.u = pyref.permute(K.mul_arrays(h, K.exp_fixed(g, self.r)), self.pi)
        self.alpha = self.rand.ring_element()
        self.epsilon = [x % q for x in self.rand.int_array(self.size, self.ebitlen + self.vbitlen + self.rbitlen)]
        self.Ap = K.mul(K.exp(g, self.alpha), K.exp_prod(h, self.epsilon))

    def setInstance(self, pkey, w, wp, s=None):
        self.pkey, self.w, self.wp, self.s = list(pkey), w, wp, s

    def setBatchVector(self, e):
        self.e = list(e)

    def commit(self):
        K, q, g, h = self.K, self.q, self.g, self.h
        self.ipe = pyref.permute(self.e, inv_perm(self.pi))
        h0 = h[0]
        self.b = self.rand.ring_array(self.size)
        x, self.d = pyref.rec_lin(self.b, self.ipe, q)
        y = pyref.prods(self.ipe, q)
        self.B = K.mul_arrays(K.exp_fixed(g, x), K.exp_fixed(h0, y))
        self.beta = self.rand.ring_array(self.size)
        xp = pyref.shift_push(x, 0)
        yp = pyref.shift_push(y, 1)
        beta_add_prod = [(bt + a * ep) % q for bt, a, ep in zip(self.beta, xp, self.epsilon)]
        yp_mul_epsilon = [a * ep % q for a, ep in zip(yp, self.epsilon)]
        self.Bp = K.mul_arrays(K.exp_fixed(g, beta_add_prod), K.exp_fixed(h0, yp_mul_epsilon))
        self.gamma = self.rand.ring_element()
        self.Cp = K.exp(g, self.gamma)
        self.delta = self.rand.ring_element()
        self.Dp = K.exp(g, self.delta)
        width = len(self.pkey) // 2
        self.phi = [self.rand.ring_element() for _ in range(width)]
        self.Fp = [K.mul(K.exp(pk, -self.phi[c % width]), t) for c, (pk, t) in enumerate(zip(self.pkey, self._cx(self.wp, self.epsilon)))]
        return {"B": self.B, "Ap": self.Ap, "Bp": self.Bp, "Cp": self.Cp, "Dp": self.Dp, "Fp": self.Fp}

    def reply(self, v):
        q = self.q
        self.v = v
        a = pyref.inner_product(self.r, self.ipe, q)
        c = sum(self.r) % q
        f = [pyref.inner_product(si, self.e, q) for si in self.s]
        return {"k_A": (a * v + self.alpha) % q, "k_B": pyref.mul_add(self.b, v % q, self.beta, q),
                "k_C": (c * v + self.gamma) % q, "k_D": (self.d * v + self.delta) % q,
                "k_E": pyref.mul_add(self.ipe, v % q, self.epsilon, q),
                "k_F": [(fc * v + ph) % q for fc, ph in zip(f, self.phi)]}

    def computeAF(self):
        self.A = self.K.exp_prod(self.u, self.e)
        self.F = self._cx(self.w, self.e)

    def setCommitment(self, msg):
        self.B, self.Ap, self.Bp = msg["B"], msg["Ap"], msg["Bp"]
        self.Cp, self.Dp, self.Fp = msg["Cp"], msg["Dp"], msg["Fp"]

    def verify(self, reply, v):
        K, q, g, h = self.K, self.q, self.g, self.h
        k_A, k_B, k_C, k_D, k_E, k_F = (reply[k] for k in ("k_A", "k_B", "k_C", "k_D", "k_E", "k_F"))
        h0 = h[0]
        C = K.mul(K.prod(self.u), K.inv(K.prod(h)))
        eprod = 1
        for t in self.e:
            eprod = eprod * t % q
        D = K.mul(self.B[self.size - 1], K.inv(K.exp(h0, eprod)))
        self.C, self.D = C, D            # the verifier's intermediates (getC / getD, PoSBasicTW.java:949, 958)
        verdictA = K.mul(K.exp(self.A, v), self.Ap) == K.mul(K.exp(g, k_A), K.exp_prod(h, k_E))
        left = K.mul_arrays(K.exp_scalar(self.B, v), self.Bp)
        right = K.mul_arrays(K.exp_fixed(g, k_B), K.exp_array(pyref.shift_push(self.B, h0), k_E))
        verdictB = left == right
        verdictC = K.mul(K.exp(C, v), self.Cp) == K.exp(g, k_C)
        verdictD = K.mul(K.exp(D, v), self.Dp) == K.exp(g, k_D)
        prods = self._cx(self.wp, k_E)
        width = len(self.pkey) // 2
        verdictF = all(K.mul(K.exp(Fc, v), Fpc) == K.mul(K.exp(pk, -k_F[c % width]), t)
                       for c, (Fc, Fpc, pk, t) in enumerate(zip(self.F, self.Fp, self.pkey, prods)))
        self.verdicts = (verdictA, verdictB, verdictC, verdictD, verdictF)
        return all(self.verdicts)


class GCCPoS:
    def __init__(self, K, vbitlen, ebitlen, rbitlen, rand=None):
        self.K, self.q = K, K.q
        self.vbitlen, self.ebitlen, self.rbitlen, self.rand = vbitlen, ebitlen, rbitlen, rand

    def _cx(self, w, E):
        return [self.K.exp_prod(c, E) for c in w]

    def setInstance(self, g, h, u, pkey, w, wp, r=None, pi=None, s=None):
        self.g, self.h, self.u, self.pkey, self.w, self.wp = g, list(h), list(u), list(pkey), w, wp
        self.r, self.s = r, s
        self.pi = list(pi) if pi is not None else None
        self.size = len(h)

    def setBatchVector(self, e):
        self.e = list(e)

    def commit(self):
        K, q, g, h = self.K, self.q, self.g, self.h
        self.ipe = pyref.permute(self.e, inv_perm(self.pi))
        self.alpha = self.rand.ring_element()
        self.epsilon = [t % q for t in self.rand.int_array(self.size, self.ebitlen + self.vbitlen + self.rbitlen)]
        self.Ap = K.mul(K.exp(g, self.alpha), K.exp_prod(h, self.epsilon))
        width = len(self.pkey) // 2
        self.beta = [self.rand.ring_element() for _ in range(width)]
        self.Bp = [K.mul(K.exp(pk, -self.beta[c % width]), t) for c, (pk, t) in enumerate(zip(self.pkey, self._cx(self.wp, self.epsilon)))]
        return {"Ap": self.Ap, "Bp": self.Bp}

    def reply(self, v):
        q = self.q
        a = pyref.inner_product(self.r, self.ipe, q)
        b = [pyref.inner_product(si, self.e, q) for si in self.s]
        return {"k_A": (a * v + self.alpha) % q, "k_B": [(bc * v + bt) % q for bc, bt in zip(b, self.beta)],
                "k_E": pyref.mul_add(self.ipe, v % q, self.epsilon, q)}

    def setCommitment(self, msg):
        self.Ap, self.Bp = msg["Ap"], msg["Bp"]

    def computeAB(self, raisedu=None):
        """CCPoSBasicW.java:493-506: plain, or AB = (w * raisedu).expProd(e) -- the base-group array multiplies every
        component of the ciphertext array."""
        K = self.K
        if raisedu is None:
            self.A = K.exp_prod(self.u, self.e)
            self.B = self._cx(self.w, self.e)
        else:
            self.AB = [K.exp_prod(K.mul_arrays(c, raisedu), self.e) for c in self.w]

    def verify(self, reply, v, raisedh=None, raisedExponent=None):
        """:519-584; with raisedExponent the single equation :571-580:
        AB^v (B' A'^rho) = pk^(-k_B) prod (w'_i h_i^rho)^(k_E,i) g^(k_A rho)."""
        K, g, h = self.K, self.g, self.h
        k_A, k_B, k_E = reply["k_A"], reply["k_B"], reply["k_E"]
        width = len(self.pkey) // 2
        if raisedExponent is None:
            if K.mul(K.exp(self.A, v), self.Ap) != K.mul(K.exp(g, k_A), K.exp_prod(h, k_E)):
                return False
            prods = self._cx(self.wp, k_E)
            return all(K.mul(K.exp(Bc, v), Bpc) == K.mul(K.exp(pk, -k_B[c % width]), t)
                       for c, (Bc, Bpc, pk, t) in enumerate(zip(self.B, self.Bp, self.pkey, prods)))
        rho = raisedExponent
        Ap_rho = K.exp(self.Ap, rho)
        g_term = K.exp(g, k_A * rho % self.q)
        ok = True
        for c, (ABc, Bpc, pk, col) in enumerate(zip(self.AB, self.Bp, self.pkey, self.wp)):
            t = K.exp_prod(K.mul_arrays(col, raisedh), k_E)
            ok = ok and K.mul(K.exp(ABc, v), K.mul(Bpc, Ap_rho)) == K.mul(K.mul(K.exp(pk, -k_B[c % width]), t), g_term)
        return ok


class IndependentGeneratorsI:
    """IndependentGeneratorsBasicI over Python integers (distr/IndependentGeneratorsBasicI.java:166-299): party j
    proves knowledge of the exponents s of its generator parts h_j = g^s.  Parties 1..threshold."""

    def __init__(self, p, q, j, threshold, rand=None):
        self.p, self.q, self.j, self.threshold, self.rand = p, q, j, threshold, rand
        self.Ap, self.k_a = {}, {}

    def setInstance(self, g, h, s, combinedh):
        self.g, self.h, self.s, self.combinedh = g, h, s, combinedh

    def setBatchVector(self, e):
        self.e = list(e)

    def commit(self):
        self.a = pyref.inner_product(self.s, self.e, self.q)
        self.r = self.rand.ring_element()
        self.Ap[self.j] = pow(self.g, self.r, self.p)
        return self.Ap[self.j]

    def reply(self, v):
        self.k_a[self.j] = (self.a * v + self.r) % self.q
        return self.k_a[self.j]

    def verify(self, l, v):
        lhs = pow(pyref.exp_prod(self.h[l], self.e, self.p), v % self.q, self.p) * self.Ap[l] % self.p
        return lhs == pow(self.g, self.k_a[l], self.p)

    def verify_combined(self, v):
        k, A = 0, 1
        for l in range(1, self.threshold + 1):
            k = (k + self.k_a[l]) % self.q
            A = A * self.Ap[l] % self.p
        lhs = pow(pyref.exp_prod(self.combinedh, self.e, self.p), v % self.q, self.p) * A % self.p
        return lhs == pow(self.g, k, self.p)
