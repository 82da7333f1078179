"""TEST-ONLY mirror (tests/mirror/): the sharded proof of a shuffle restated in Python against the array *interface* only, so
that the CPU suite can run it on gloo ranks with an integer-backed stand-in for the arrays (tests/fake_backend.py) and
compare with the single-process oracle transcript.  The product's sharded drivers are the C++ ones
(``vmn_pos_set_comm``, csrc/vmnproofs.cpp); this file cross-checks their exchange pattern.  Not part of the package.

The classes mirror ``hvzk.PoSBasicTW`` / ``CCPoSBasicW`` (same method names and message layout; the messages hold the
local shards of the array-valued parts).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

from verificatum_vmn_amd.parallel import Comm, shard_bounds  # noqa: F401


def _take(arr, idx):
    """Rows ``idx`` (an index list / numpy index array / range) of a host-side random or public array,
    which is either a list of ints or fixed-width big-endian bytes (bulk path, numpy)."""
    if isinstance(arr, (bytes, bytearray)):
        raise TypeError("byte blocks need the row width: use _take_bytes")
    if isinstance(idx, range):
        return arr[idx.start:idx.stop]
    return [arr[int(i)] for i in idx]


def _as_row_matrix(arr, nbytes: int):
    """A block of big-endian rows (bytes, a pinned torch tensor, a uint8 array) as an (n, nbytes) uint8 view;
    None when ``arr`` is a list of integers."""
    import numpy as np
    if isinstance(arr, (bytes, bytearray)):
        return np.frombuffer(arr, dtype=np.uint8).reshape(-1, nbytes)
    if hasattr(arr, "data_ptr") and hasattr(arr, "numpy"):
        return arr.numpy().reshape(-1, nbytes)               # host tensor: shares memory
    if isinstance(arr, np.ndarray) and arr.dtype == np.uint8:
        return arr.reshape(-1, nbytes)
    return None


def _take_rows(arr, idx, nbytes: int):
    a = _as_row_matrix(arr, nbytes)
    if a is not None:
        import numpy as np
        if isinstance(idx, range):
            return a[idx.start:idx.stop].tobytes()
        return a[np.asarray(idx, dtype=np.int64)].tobytes()
    return _take(arr, idx)


class _ShardedBase:
    """What the sharded mirrors share: the exchanges of single elements / ring scalars and the slicing of host tapes."""

    def __init__(self, group, vbitlen: int, ebitlen: int, rbitlen: int, comm: Comm, rand=None):
        self.G, self.comm, self.rand = group, comm, rand
        self.p, self.q, self.nb = group.p, group.q, max(group.nbytes, group.exp_bytes)
        self.xb = group.exp_bytes                     # width of the tape's ring rows
        self.vbitlen, self.ebitlen, self.rbitlen = vbitlen, ebitlen, rbitlen
        qbits = self.q.bit_length()
        self.e_bits = min(ebitlen, qbits)
        self.eps_bits = min(ebitlen + vbitlen + rbitlen, qbits)

    # ---- helpers (single elements through the group object: ModPGroup integers or ECqPGroup points) --------
    def _gexp(self, base, e: int):
        return self.G.k_exp(base, e)

    def _div(self, a, b):
        return self.G.k_mul(a, self.G.k_inv(b))

    def _expmul(self, a, v: int, b):
        return self.G.k_mul(self.G.k_exp(a, v), b)

    def _gather_elems(self, local):
        """All-gather of a few group elements per rank (fixed-width wire encoding); returns per-rank lists."""
        eb = self.G.elem_bytes
        as_ints = [int.from_bytes(self.G.enc_el(x), "big") for x in local]
        parts = self.comm.all_gather_ints(as_ints, eb)
        return [[self.G.dec_el(v.to_bytes(eb, "big")) for v in pr] for pr in parts]

    def _prod_all(self, local) -> list:
        """Component-wise group product of every rank's partial elements (all-gather + multiply: modular
        multiplication / point addition is not an RCCL reduction operator)."""
        parts = self._gather_elems(local)
        return [self.G.mulPartials([pr[c] for pr in parts]) for c in range(len(local))]

    def _sum_all(self, local: Sequence[int]) -> List[int]:
        parts = self.comm.all_gather_ints(local, self.nb)
        return [sum(pr[c] for pr in parts) % self.q for c in range(len(local))]

    def _prodq_all(self, local: int) -> int:
        acc = 1
        for pr in self.comm.all_gather_ints([local], self.nb):
            acc = acc * pr[0] % self.q
        return acc

    def _ring_rows(self, arr, idx):
        rows = _take_rows(arr, idx, self.xb)
        if not isinstance(rows, (bytes, bytearray)):
            rows = [x % self.q for x in rows]        # integers longer than q (epsilon over a 256-bit curve order) act mod q
        return self.G.ringArray(rows)

    def _set_size(self, size: int):
        self.size = size
        self.lo, self.hi = shard_bounds(size, self.comm.world, self.comm.rank)
        self.local = range(self.lo, self.hi)

    def _local(self, arr):
        """An array argument is the whole array (replicated) or already this rank's shard -- told apart by size."""
        if arr.size() == self.size and (self.hi - self.lo) != self.size:
            return arr.copyOfRange(self.lo, self.hi)
        return arr


class ShardedPoSBasicTW(_ShardedBase):
    """Proof of a shuffle with every array sharded by position; see the module docstring.

    ``group`` is a ``ModPGroup`` (or any object with the same interface).  Replicated public inputs are
    handed over as full arrays on this rank's GPU; secrets and the batching vector come from host
    tapes shared by all ranks.  (Python mirror of the sharded C++ driver, ``vmn_pos_set_comm``: the CPU suite runs it
    on gloo ranks over tests/fake_backend.py; the product path is ``native.PoSBasicTW.setComm``.)
    """

    # ---- setup --------------------------------------------------------------------------------
    def precompute(self, g: int, h_full, pi=None):
        """h_full: the N independent generators, replicated.  Prover (pi given): draws r, alpha, epsilon
        from the shared tape and computes its shard of u and the global A'."""
        G = self.G
        self._set_size(h_full.size())
        self.g, self.h_full = g, h_full
        self.h = h_full.copyOfRange(self.lo, self.hi)
        self.h0 = h_full.get(0)
        if pi is None:
            return
        self.pi = pi
        self.piinv = _inv(pi)
        r_full = self.rand.ring_array(self.size)
        self.alpha = self.rand.ring_element()
        eps_full = self.rand.int_array(self.size, self.ebitlen + self.vbitlen + self.rbitlen)
        self.r = self._ring_rows(r_full, self.local)                     # r indexed like h
        pi_loc = _slice_idx(pi, self.lo, self.hi)
        r_perm = self._ring_rows(r_full, pi_loc)                          # r_{pi(i)}, i in the shard
        tmp1 = G.exp(g, r_perm)
        tmp2 = h_full.permute(pi_loc)                                     # h_{pi(i)}
        self.u = tmp2.mul(tmp1)                                           # u_i = h_{pi(i)} g^{r_{pi(i)}}
        for t in (tmp1, tmp2, r_perm):
            t.free()
        self.epsilon = self._ring_rows(eps_full, self.local)
        part = self.h.expProd(self.epsilon, self.eps_bits)
        self.Ap = self.G.k_mul(self._gexp(g, self.alpha), self._prod_all([part])[0])

    def reencrypt(self, pkey: Sequence[int], w_full, s_full):
        """This rank's shard of w' = permute(w pk^s, pi^-1): local gathers of the replicated input.
        s_full: one host array (list / bytes) per column."""
        width = len(pkey) // 2
        piinv_loc = _slice_idx(self.piinv, self.lo, self.hi)
        wp = []
        for c, pk in enumerate(pkey):
            s_perm = self._ring_rows(s_full[c % width], piinv_loc)
            f = self.G.exp(pk, s_perm)
            wsel = w_full[c].permute(piinv_loc)
            wp.append(wsel.mul(f))
            for t in (s_perm, f, wsel):
                t.free()
        self.s = [self._ring_rows(col, self.local) for col in s_full]     # s indexed like w
        return wp

    def setInstance(self, pkey: Sequence[int], w_full, wp, s=None):
        self.pkey = list(pkey)
        self.w = [c.copyOfRange(self.lo, self.hi) for c in w_full]
        self.wp = wp
        if s is not None:
            self.s = s

    def setPermutationCommitment(self, u):
        self.u = u

    def setBatchVector(self, e_full):
        self.e_full = e_full
        self.e = self._ring_rows(e_full, self.local)

    # ---- sharded scans --------------------------------------------------------------------------
    def _scans(self, b, ipe):
        """x = b.recLin(ipe), y = ipe.prods() over the *global* index space, from local scans plus one
        all-gather of the carries.  Returns (x, y, d, x_in, y_in)."""
        q = self.q
        x_loc, d_loc = b.recLin(ipe)
        P = ipe.prods()
        n_loc = self.hi - self.lo
        e_tot = P.get(n_loc - 1) if n_loc else 1
        carr = self.comm.all_gather_ints([e_tot, d_loc if n_loc else 0], self.nb)
        x_in, y_in = 0, 1
        for k in range(self.comm.rank):
            x_in = (x_in * carr[k][0] + carr[k][1]) % q
            y_in = y_in * carr[k][0] % q
        d = 0
        for k in range(self.comm.world):
            d = (d * carr[k][0] + carr[k][1]) % q
        if self.comm.rank == 0:
            return x_loc, P, d, x_in, y_in
        x = P.mulAdd(x_in, x_loc)
        y = P.mulAdd(y_in, None)
        x_loc.free()
        P.free()
        return x, y, d, x_in, y_in

    # ---- prover ---------------------------------------------------------------------------------
    def commit(self):
        G, g, p = self.G, self.g, self.p
        piinv_loc = _slice_idx(self.piinv, self.lo, self.hi)
        self.ipe = self._ring_rows(self.e_full, piinv_loc)               # e'_i = e_{pi^-1(i)}
        b_full = self.rand.ring_array(self.size)
        self.b = self._ring_rows(b_full, self.local)
        x, y, self.d, x_in, y_in = self._scans(self.b, self.ipe)
        g_exp_x = G.exp(g, x)
        h0_exp_y = G.exp(self.h0, y)
        self.B = g_exp_x.mul(h0_exp_y)
        g_exp_x.free()
        h0_exp_y.free()
        beta_full = self.rand.ring_array(self.size)
        self.beta = self._ring_rows(beta_full, self.local)
        xp = x.shiftPush(x_in)                                             # x_{lo-1}: the carry into this shard
        yp = y.shiftPush(y_in)
        xp_mul_epsilon = xp.mul(self.epsilon)
        beta_add_prod = self.beta.add(xp_mul_epsilon)
        g_exp_beta_add_prod = G.exp(g, beta_add_prod)
        yp_mul_epsilon = yp.mul(self.epsilon)
        h0_exp_yp_mul_epsilon = G.exp(self.h0, yp_mul_epsilon)
        self.Bp = g_exp_beta_add_prod.mul(h0_exp_yp_mul_epsilon)
        for t in (x, y, xp, yp, xp_mul_epsilon, beta_add_prod, g_exp_beta_add_prod, yp_mul_epsilon, h0_exp_yp_mul_epsilon):
            t.free()
        self.gamma = self.rand.ring_element()
        self.Cp = self._gexp(g, self.gamma)
        self.delta = self.rand.ring_element()
        self.Dp = self._gexp(g, self.delta)
        width = len(self.pkey) // 2
        self.phi = [self.rand.ring_element() for _ in range(width)]
        parts = [c.expProd(self.epsilon, self.eps_bits) for c in self.wp]
        prods = self._prod_all(parts)
        self.Fp = [self.G.k_mul(self._gexp(pk, -self.phi[c % width]), t) for c, (pk, t) in enumerate(zip(self.pkey, prods))]
        return {"B": self.B, "Ap": self.Ap, "Bp": self.Bp, "Cp": self.Cp, "Dp": self.Dp, "Fp": self.Fp}

    def setChallenge(self, v: int):
        self.v = int(v)

    def reply(self, v: int):
        self.setChallenge(v)
        q = self.q
        local = [self.r.innerProduct(self.ipe), self.r.sum()] + [si.innerProduct(self.e) for si in self.s]
        tot = self._sum_all(local)
        a, c, f = tot[0], tot[1], tot[2:]
        self.k_A = (a * v + self.alpha) % q
        self.k_B = self.b.mulAdd(v % q, self.beta)
        self.k_C = (c * v + self.gamma) % q
        self.k_D = (self.d * v + self.delta) % q
        self.k_E = self.ipe.mulAdd(v % q, self.epsilon)
        self.k_F = [(fc * v + ph) % q for fc, ph in zip(f, self.phi)]
        return {"k_A": self.k_A, "k_B": self.k_B, "k_C": self.k_C, "k_D": self.k_D, "k_E": self.k_E, "k_F": self.k_F}

    # ---- verifier -------------------------------------------------------------------------------
    def computeAF(self):
        parts = [self.u.expProd(self.e, self.e_bits)] + [c.expProd(self.e, self.e_bits) for c in self.w]
        tot = self._prod_all(parts)
        self.A, self.F = tot[0], tot[1:]

    def setCommitment(self, msg):
        self.B, self.Ap, self.Bp = msg["B"], msg["Ap"], msg["Bp"]
        self.Cp, self.Dp, self.Fp = msg["Cp"], msg["Dp"], msg["Fp"]

    def verify(self, reply) -> bool:
        G, g, p, v = self.G, self.g, self.p, self.v
        k_A, k_B, k_C, k_D, k_E, k_F = (reply[k] for k in ("k_A", "k_B", "k_C", "k_D", "k_E", "k_F"))
        n_loc = self.hi - self.lo
        # one exchange for all partial products of this phase + each shard's last B element
        b_last = self.B.get(n_loc - 1) if n_loc else self.G.ONE
        kE_bits = max(1, k_E.maxBits())            # every bit of a received exponent counts (this shard's maximum)
        parts = [self.u.prod(), self.h.prod(), self.h.expProd(k_E, kE_bits), b_last] + \
                [c.expProd(k_E, kE_bits) for c in self.wp]
        gathered = self._gather_elems(parts)
        mulp = lambda idx: self.G.mulPartials([pr[idx] for pr in gathered])
        u_prod, h_prod, h_kE = mulp(0), mulp(1), mulp(2)
        wp_kE = [mulp(4 + c) for c in range(len(self.wp))]
        b_lasts = [pr[3] for pr in gathered]
        e_prod = self._prodq_all(self.e.prod())
        C = self._div(u_prod, h_prod)
        B_final = next(b_lasts[k] for k in range(self.comm.world - 1, -1, -1)
                       if shard_bounds(self.size, self.comm.world, k)[1] > shard_bounds(self.size, self.comm.world, k)[0])
        D = self._div(B_final, self._gexp(self.h0, e_prod))
        verdictA = self._expmul(self.A, v, self.Ap) == self.G.k_mul(self._gexp(g, k_A), h_kE)
        # B check on the shard; the element shifted in is the previous non-empty shard's last B (or h0)
        prev = self.h0
        for k in range(self.comm.rank):
            klo, khi = shard_bounds(self.size, self.comm.world, k)
            if khi > klo:
                prev = b_lasts[k]
        B_exp_v = self.B.exp(v)
        leftSide = B_exp_v.mul(self.Bp)
        g_exp_k_B = G.exp(g, k_B)
        B_shift = self.B.shiftPush(prev)
        B_shift_exp_k_E = B_shift.exp(k_E, kE_bits)
        rightSide = g_exp_k_B.mul(B_shift_exp_k_E)
        verdictB = self.comm.all_true(leftSide.equals(rightSide))
        for t in (B_exp_v, leftSide, g_exp_k_B, B_shift, B_shift_exp_k_E, rightSide):
            t.free()
        verdictC = self._expmul(C, v, self.Cp) == self._gexp(g, k_C)
        verdictD = self._expmul(D, v, self.Dp) == self._gexp(g, k_D)
        width = len(self.pkey) // 2
        verdictF = all(self._expmul(Fc, v, Fpc) == self.G.k_mul(self._gexp(pk, -k_F[c % width]), t)
                       for c, (Fc, Fpc, pk, t) in enumerate(zip(self.F, self.Fp, self.pkey, wp_kE)))
        self.verdicts = (verdictA, verdictB, verdictC, verdictD, verdictF)
        return all(self.verdicts)


class ShardedCCPoSBasicW(_ShardedBase):
    """Commitment-consistent proof of a shuffle, sharded by position (mirror of ``vmn_ccpos_set_comm``;
    ref: hvzk/CCPoSBasicW.java:344-396, 462-506, 519-584).  h is the whole array; u, w, w', r, s may be whole arrays or
    shards; the batching vector and epsilon come from tapes shared by all ranks."""

    def setInstance(self, g, h_full, u, pkey, w, wp, r=None, pi=None, s=None):
        self._set_size(h_full.size())
        self.g, self.pkey = g, list(pkey)
        self.h = self._local(h_full)
        self.u = self._local(u)
        self.w = [self._local(c) for c in w]
        self.wp = [self._local(c) for c in wp]
        self.r = self._local(r) if r is not None else None
        self.s = [self._local(c) for c in s] if s is not None else None
        self.piinv = _inv(pi) if pi is not None else None

    def setBatchVector(self, e_full):
        self.e_full = e_full
        self.e = self._ring_rows(e_full, self.local)

    def commit(self):
        G, g = self.G, self.g
        self.ipe = self._ring_rows(self.e_full, _slice_idx(self.piinv, self.lo, self.hi))
        self.alpha = self.rand.ring_element()
        eps_full = self.rand.int_array(self.size, self.ebitlen + self.vbitlen + self.rbitlen)
        self.epsilon = self._ring_rows(eps_full, self.local)
        width = len(self.pkey) // 2
        self.beta = [self.rand.ring_element() for _ in range(width)]
        parts = [self.h.expProd(self.epsilon, self.eps_bits)] + [c.expProd(self.epsilon, self.eps_bits) for c in self.wp]
        tot = self._prod_all(parts)
        self.Ap = G.k_mul(self._gexp(g, self.alpha), tot[0])
        self.Bp = [G.k_mul(self._gexp(pk, -self.beta[c % width]), t) for c, (pk, t) in enumerate(zip(self.pkey, tot[1:]))]
        return {"Ap": self.Ap, "Bp": self.Bp}

    def reply(self, v: int):
        q = self.q
        tot = self._sum_all([self.r.innerProduct(self.ipe)] + [si.innerProduct(self.e) for si in self.s])
        self.k_E = self.ipe.mulAdd(v % q, self.epsilon)
        return {"k_A": (tot[0] * v + self.alpha) % q, "k_B": [(b * v + bt) % q for b, bt in zip(tot[1:], self.beta)], "k_E": self.k_E}

    def setCommitment(self, msg):
        self.Ap, self.Bp = msg["Ap"], msg["Bp"]

    def setChallenge(self, v: int):
        self.v = int(v)

    def computeAB(self, raisedu=None):
        if raisedu is None:
            tot = self._prod_all([self.u.expProd(self.e, self.e_bits)] + [c.expProd(self.e, self.e_bits) for c in self.w])
            self.A, self.B = tot[0], tot[1:]
        else:
            ru = self._local(raisedu)
            self.AB = self._prod_all([c.mul(ru).expProd(self.e, self.e_bits) for c in self.w])

    def verify(self, reply, raisedh=None, raisedExponent=None) -> bool:
        G, g, v = self.G, self.g, self.v
        k_A, k_B, k_E = reply["k_A"], reply["k_B"], reply["k_E"]
        width = len(self.pkey) // 2
        kE_bits = max(1, k_E.maxBits())
        if raisedExponent is None:
            tot = self._prod_all([self.h.expProd(k_E, kE_bits)] + [c.expProd(k_E, kE_bits) for c in self.wp])
            if self._expmul(self.A, v, self.Ap) != G.k_mul(self._gexp(g, k_A), tot[0]):
                return False
            return all(self._expmul(Bc, v, Bpc) == G.k_mul(self._gexp(pk, -k_B[c % width]), t)
                       for c, (Bc, Bpc, pk, t) in enumerate(zip(self.B, self.Bp, self.pkey, tot[1:])))
        rho = raisedExponent
        rh = self._local(raisedh)
        tot = self._prod_all([c.mul(rh).expProd(k_E, kE_bits) for c in self.wp])
        Ap_rho = self._gexp(self.Ap, rho)
        g_term = self._gexp(g, k_A * rho % self.q)
        return all(self._expmul(ABc, v, G.k_mul(Bpc, Ap_rho)) == G.k_mul(G.k_mul(self._gexp(pk, -k_B[c % width]), t), g_term)
                   for c, (ABc, Bpc, pk, t) in enumerate(zip(self.AB, self.Bp, self.pkey, tot)))


def _inv(pi):
    try:
        import numpy as np
        a = np.asarray(pi, dtype=np.int64)
        inv = np.empty(len(a), dtype=np.uint32)
        inv[a] = np.arange(len(a), dtype=np.uint32)
        return inv
    except ImportError:      # pragma: no cover
        inv = [0] * len(pi)
        for i, j in enumerate(pi):
            inv[j] = i
        return inv


def _slice_idx(idx, lo, hi):
    return idx[lo:hi]
