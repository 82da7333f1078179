"""Shared drivers of the proof-level parity tests: one function runs a whole flow (re-encryption, proof, verification)
through the product's drivers AND through the oracle's group-generic restatement on the same random tape, and
compares every message.  Test infrastructure only."""
import importlib.util
import os
import sys

from oracle import pyref_proofs as P
from tape import Tape


def load_driver_modules(entry):
    """The C++ drivers' bindings (native) and the test-only Python mirror (tests/mirror: hvzk, mixnet, elgamal)."""
    import mirror
    out = mirror.load(entry, ("hvzk", "mixnet", "native", "elgamal"))
    return out


def ints_of(x):
    return x.toInts() if hasattr(x, "toInts") else x


def same_msg(a, b):
    assert set(a) == set(b)
    for k in a:
        assert ints_of(a[k]) == ints_of(b[k]), k


def make_instance(K, g, n, width, seed):
    """Public instance over the oracle adapter K: independent generators, key, `width`-wide ciphertexts (2*width
    component lists [u_1..u_w, v_1..v_w]) of random group elements, as ProtocolElGamalInterfaceRaw.demoCiphertexts
    (P/elgamal/ProtocolElGamalInterfaceRaw.java:99-130) makes them."""
    t = Tape(seed, K.q)
    h = K.exp_fixed(g, t.ring_array(n))
    y = K.exp(g, t.ring_element())
    pkey = [g] * width + [y] * width
    msgs = [K.exp_fixed(g, t.ring_array(n)) for _ in range(width)]
    enc_r = [t.ring_array(n) for _ in range(width)]
    w = [K.exp_fixed(g, enc_r[c]) for c in range(width)] + \
        [K.mul_arrays(msgs[c], K.exp_fixed(y, enc_r[c])) for c in range(width)]
    return h, pkey, w, t


def reencrypt_product(impl, mods, G, pkey, W, S, pi):
    if impl == "native":
        return mods["native"].reencrypt_native(G, pkey, W, S, pi)
    mx = mods["mixnet"]
    return mx.reencrypt(W, mx.reencFactors(G, pkey, S), pi)


def check_pos(impl, mods, G, K, g, h, pkey, w, t, bits3, tamper=True):
    """A0 + A1: re-encrypt, PoSBasicTW prove and verify; every message equals the oracle's.  Returns the arrays for
    callers that go on (H, W, WP, wp_o, s, pi)."""
    NV, NE, NR = bits3
    hv = mods["hvzk" if impl == "python" else "native"]
    n, width, q = len(h), len(pkey) // 2, K.q
    pi = t.permutation(n)
    s = [t.ring_array(n) for _ in range(width)]
    e = t.int_array(n, NE)
    v = t.int_array(1, NV)[0]
    o = P.GPoS(K, NV, NE, NR, rand=Tape(b"prover", q))
    o.precompute(g, h, pi)
    wp_o = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi)
    o.setInstance(pkey, w, wp_o, s)
    o.setBatchVector(e)
    com_o, rep_o = o.commit(), o.reply(v)
    H = G.toElementArray(h)
    W = [G.toElementArray(c) for c in w]
    S = [G.ringArray(c) for c in s]
    pr = hv.PoSBasicTW(G, NV, NE, NR, rand=Tape(b"prover", q))
    pr.precompute(g, H, pi)
    assert pr.u.toInts() == o.u
    WP = reencrypt_product(impl, mods, G, pkey, W, S, pi)
    assert [c.toInts() for c in WP] == wp_o
    pr.setInstance(pkey, W, WP, S)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    same_msg(com, com_o)
    same_msg(rep, rep_o)
    ver = hv.PoSBasicTW(G, NV, NE, NR)
    ver.precompute(g, H)
    ver.setPermutationCommitment(pr.u)
    ver.setInstance(pkey, W, WP)
    ver.setBatchVector(e)
    ver.computeAF()
    ver.setCommitment(com)
    ver.setChallenge(v)
    assert ver.verify(rep)
    if tamper:
        bad = dict(rep)
        bad["k_F"] = [(x + (1 if c == width - 1 else 0)) % q for c, x in enumerate(rep["k_F"])]     # last column only
        assert not ver.verify(bad) and ver.verdicts == (True, True, True, True, False)
    # the oracle's verifier accepts the product's messages
    ov = P.GPoS(K, NV, NE, NR)
    ov.precompute(g, h)
    ov.u = pr.u.toInts()
    ov.setInstance(pkey, w, wp_o)
    ov.setBatchVector(e)
    ov.computeAF()
    ov.setCommitment({k: ints_of(x) for k, x in com.items()})
    assert ov.verify({k: ints_of(x) for k, x in rep.items()}, v)
    if impl == "native":
        # the verifier's intermediates equal the oracle's (getA :716, getF :761, getC :949, getD :958 of PoSBasicTW.java: what
        # `vmnv -t PoS.A,PoS.F,PoS.C,PoS.D` prints); read after the last verify() -- the tampered reply changes none of them
        assert ver.getA() == ov.A and list(ver.getF()) == list(ov.F), "PoS.A / PoS.F differ from the oracle"
        assert ver.getC() == ov.C and ver.getD() == ov.D, "PoS.C / PoS.D differ from the oracle"
    return H, W, WP, wp_o, s, S, pi


def check_ccpos(impl, mods, G, K, g, h, H, u_o, U, pkey, w, W, wp_o, WP, r, R, pi, s, S, t, bits3, rho=None):
    """A3 on a given permutation commitment (u_o / U with opening r, pi): prove, verify plain and -- with rho -- in
    the raised single-equation form; every message equals the oracle's."""
    NV, NE, NR = bits3
    hv = mods["hvzk" if impl == "python" else "native"]
    n, width, q = len(h), len(pkey) // 2, K.q
    e = t.int_array(n, NE)
    v = t.int_array(1, NV)[0]
    oc = P.GCCPoS(K, NV, NE, NR, rand=Tape(b"ccprover", q))
    oc.setInstance(g, h, u_o, pkey, w, wp_o, r, pi, s)
    oc.setBatchVector(e)
    cc_o, cr_o = oc.commit(), oc.reply(v)
    cp = hv.CCPoSBasicW(G, NV, NE, NR, rand=Tape(b"ccprover", q))
    cp.setInstance(g, H, U, pkey, W, WP, R, pi, S)
    cp.setBatchVector(e)
    cc, cr = cp.commit(), cp.reply(v)
    same_msg(cc, cc_o)
    same_msg(cr, cr_o)
    cv = hv.CCPoSBasicW(G, NV, NE, NR)
    cv.setInstance(g, H, U, pkey, W, WP)
    cv.setBatchVector(e)
    cv.setCommitment(cc)
    cv.setChallenge(v)
    cv.computeAB()
    assert cv.verify(cr)
    bad = dict(cr)
    bad["k_B"] = [(x + (1 if c == 0 else 0)) % q for c, x in enumerate(cr["k_B"])]
    assert not cv.verify(bad)
    # the oracle's verifier on the product's messages, plain form
    ov = P.GCCPoS(K, NV, NE, NR)
    ov.setInstance(g, h, u_o, pkey, w, wp_o)
    ov.setBatchVector(e)
    ov.setCommitment({k: ints_of(x) for k, x in cc.items()})
    ov.computeAB()
    assert ov.verify({k: ints_of(x) for k, x in cr.items()}, v)
    if impl == "native":
        A_gpu, B_gpu = cv.getAB()                      # computeAB's values (CCPoSBasicW.java:493-506), plain form
        assert A_gpu == ov.A and list(B_gpu) == list(ov.B), "CCPoS A / B differ from the oracle"
    if rho is not None:
        RU, RH = U.exp(rho), H.exp(rho)
        cv2 = hv.CCPoSBasicW(G, NV, NE, NR)
        cv2.setInstance(g, H, U, pkey, W, WP)
        cv2.setBatchVector(e)
        cv2.setCommitment(cc)
        cv2.setChallenge(v)
        cv2.computeAB(RU)
        assert cv2.verify(cr, RH, rho)
        bad = dict(cr)
        bad["k_A"] = (cr["k_A"] + 1) % q
        assert not cv2.verify(bad, RH, rho)
        ov2 = P.GCCPoS(K, NV, NE, NR)
        ov2.setInstance(g, h, u_o, pkey, w, wp_o)
        ov2.setBatchVector(e)
        ov2.setCommitment({k: ints_of(x) for k, x in cc.items()})
        ov2.computeAB(K.exp_scalar(u_o, rho))
        assert ov2.verify({k: ints_of(x) for k, x in cr.items()}, v, K.exp_scalar(h, rho), rho)
        if impl == "native":
            none, AB_gpu = cv2.getAB()                 # raised form: AB = (w u^rho).expProd(e)
            assert none is None and list(AB_gpu) == list(ov2.AB), "CCPoS AB (raised form) differs from the oracle"
    return cc, cr
