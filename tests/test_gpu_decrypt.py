"""GPU suite, row A6: threshold decryption factors, their combination with (negative) modified Lagrange
integers, plaintext recovery, the batched Chaum-Pedersen checks, and batch inversion."""
import importlib.util
import os
import sys

import pytest

from conftest import load_golden
from oracle import pyref, pyref_proofs as P
from tape import Tape

pytestmark = pytest.mark.gpu


class _NativeFacade:
    """The C++ decryption drivers (native.py) behind the call signatures of elgamal.py."""

    def __init__(self, nat, group_of):
        self.nat, self.group_of = nat, group_of
        for name in ("decryptionFactors", "combineDecryptionFactors", "plaintexts", "DistrElGamalSessionBasic"):
            setattr(self, name, getattr(nat, name))

    def prodFactor(self, q, k):
        return self.nat.prodFactor(q, k, group=self.group_of(q))

    def modifiedLagrangeCoefficients(self, q, correct, k, threshold):
        return self.nat.modifiedLagrangeCoefficients(q, correct, k, threshold, group=self.group_of(q))


@pytest.fixture(scope="module", params=["python", "native"])
def eg(request, entry, vmn, gpu_ctx):
    import mirror
    name = "elgamal" if request.param == "python" else "native"
    m = mirror.load(entry, (name,))[name]
    if request.param == "python":
        return m
    groups = {}

    def group_of(q):
        if q not in groups:
            for bits in (512, 2048):
                grp, _ = load_golden(bits)
                if grp["q"] == q:
                    groups[q] = vmn.ModPGroup(gpu_ctx, grp["p"], grp["q"], grp["g"])
        return groups[q]
    return _NativeFacade(m, group_of)


def test_batch_inversion(vmn, gpu_ctx):
    grp, _ = load_golden(2048)
    p, q, g = grp["p"], grp["q"], grp["g"]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    for n in (1, 2, 17, 1000):
        xs = [1 + v % (p - 1) for v in pyref.stream_ints(b"inv%d" % n, n, p)]
        got = G.toElementArray(xs).inv().toInts()
        assert got == [pow(x, -1, p) for x in xs], n
    assert G.toElementArray([]).inv().toInts() == []


def test_lagrange_integers_match_oracle_and_can_be_negative(eg):
    grp, _ = load_golden(512)
    q = grp["q"]
    seen_negative = False
    for k, thr, bad in ((3, 2, ()), (5, 3, (2,)), (7, 4, (1, 5)), (4, 4, ())):
        correct = [False] + [i not in bad for i in range(1, k + 1)]
        a = eg.modifiedLagrangeCoefficients(q, correct, k, thr)
        assert a == P.lagrange_integers(q, correct, k, thr)
        seen_negative |= any(c < 0 for c in a)
        assert eg.prodFactor(q, k) == P.prod_factor(q, k)
    assert seen_negative


@pytest.mark.parametrize("bits,n,k,thr,bad", [(512, 40, 3, 2, ()), (2048, 130, 5, 3, (2,))])
def test_threshold_decryption_recovers_plaintexts_and_proofs_verify(bits, n, k, thr, bad, vmn, gpu_ctx, eg):
    grp, _ = load_golden(bits)
    p, q, g = grp["p"], grp["q"], grp["g"]
    t = Tape(b"dec%d" % bits, q)
    # Shamir sharing of the key x over Z_q: x_j = poly(j), degree thr-1
    coeffs = t.ring_array(thr)
    x = coeffs[0]
    share = lambda j: sum(c * pow(j, d, q) for d, c in enumerate(coeffs)) % q
    xs = [None] + [share(j) for j in range(1, k + 1)]
    ys = [None] + [pow(g, xj, p) for xj in xs[1:]]
    y = pow(g, x, p)
    msgs = pyref.exp_fixed(g, t.ring_array(n), p)
    rs = t.ring_array(n)
    u = pyref.exp_fixed(g, rs, p)
    v = pyref.mul(msgs, pyref.exp_fixed(y, rs, p), p)
    correct = [False] + [j not in bad for j in range(1, k + 1)]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    U, V = G.toElementArray(u), G.toElementArray(v)
    # decryption factors of every party, GPU vs oracle
    F = [None]
    f_o = [None]
    for j in range(1, k + 1):
        F.append(eg.decryptionFactors(U, xs[j], q, k))
        f_o.append(P.decryption_factors(u, xs[j], p, q, k))
        assert F[j].toInts() == f_o[j]
    comb = eg.combineDecryptionFactors(F, correct, k, thr, q)
    comb_o = P.combine_decryption_factors(f_o, correct, k, thr, p, q)
    assert comb.toInts() == comb_o
    assert eg.plaintexts(V, comb).toInts() == msgs                      # decryption is correct
    # batched proofs: every party proves, a verifier checks each and the combination
    NE, NV = 100, 100
    e = t.int_array(n, NE)
    chal = t.int_array(1, NV)[0]
    ver = eg.DistrElGamalSessionBasic(G, 1, k, thr, NE)
    ver.setInstance(U, ys, F)
    ver.setBatchVector(e)
    ver.batchInput()
    assert getattr(ver, "A", None) in (None, pyref.exp_prod(u, e, p))
    for j in range(1, k + 1):
        pr = eg.DistrElGamalSessionBasic(G, j, k, thr, NE, rand=Tape(b"party%d" % j, q))
        pr.setInstance(U, ys, F)
        pr.setBatchVector(e)
        pr.batchInput()
        yp, Bp = pr.commit(xs[j])
        ver.setCommitment(j, yp, Bp)
        ver.setReply(j, pr.reply(chal))
    for j in range(1, k + 1):
        ver.batch(j)
        assert not hasattr(ver, "B") or ver.B[j] == pyref.exp_prod(f_o[j], e, p)
        assert ver.verify(j, chal)
    ver.setReply(1, (ver.k_x[1] + 1) % q)
    assert not ver.verify(1, chal)
    ver.setReply(1, (ver.k_x[1] - 1) % q)
    # combined check: combined public key = prod y_l^(lambda_l) = y^c
    ints = eg.modifiedLagrangeCoefficients(q, correct, k, thr)
    idx = [l for l in range(1, k + 1) if correct[l]][:thr]
    combinedy = 1
    for l, c in zip(idx, ints):
        combinedy = combinedy * pow(ys[l], c % q, p) % p
    # the factors are u^(-x_l/c), so the matching "public key" of the combined factors is y itself
    ver.combine(correct, y, comb)
    ver.batchCombined()
    assert ver.verifyCombined(chal)
    assert combinedy == pow(y, eg.prodFactor(q, k), p)
