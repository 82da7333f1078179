"""CPU suite: the Python restatement of PoS / PoSC / CCPoS is self-consistent the way the reference's own
unit test checks it (TestPoSCBasicTW.java:147-163): an honest transcript verifies, a tampered witness
(r <- r + r, :109-111) is rejected."""
import pytest

from conftest import load_golden
from oracle import pyref, pyref_proofs as P
from tape import Tape

NE = NV = 100
NR = 50


def setup(bits, n, width=1, seed=b"proofs"):
    grp, _ = load_golden(bits)
    p, q, g = grp["p"], grp["q"], grp["g"]
    t = Tape(seed, q)
    h = [pow(g, x, p) for x in t.ring_array(n)]
    xkey = t.ring_element()
    y = pow(g, xkey, p)
    pkey = [g] * width + [y] * width
    msgs = [[pow(g, m, p) for m in t.ring_array(n)] for _ in range(width)]
    enc_r = [t.ring_array(n) for _ in range(width)]
    w = [pyref.exp_fixed(g, enc_r[c], p) for c in range(width)] + \
        [pyref.mul(msgs[c], pyref.exp_fixed(y, enc_r[c], p), p) for c in range(width)]
    return p, q, g, h, pkey, w, t


@pytest.mark.parametrize("width", [1, 2])
def test_pos_honest_accepts_tampered_rejects(width):
    n = 12
    p, q, g, h, pkey, w, t = setup(512, n, width)
    pi = t.permutation(n)
    prover = P.PoS(p, q, NV, NE, NR, rand=t)
    prover.precompute(g, h, pi)
    s = [t.ring_array(n) for _ in range(width)]
    wp = P.reencrypt(w, P.reenc_factors(pkey, s, p), pi, p)
    prover.setInstance(pkey, w, wp, s)
    e = t.int_array(n, NE)
    prover.setBatchVector(e)
    com = prover.commit()
    v = t.int_array(1, NV)[0]
    rep = prover.reply(v)
    ver = P.PoS(p, q, NV, NE, NR)
    ver.precompute(g, h)
    ver.u = prover.u
    ver.setInstance(pkey, w, wp)
    ver.setBatchVector(e)
    ver.computeAF()
    ver.setCommitment(com)
    assert ver.verify(rep, v)
    # wrong re-encryption exponents: only F may fail
    bad = dict(rep)
    bad["k_F"] = [(x + 1) % q for x in rep["k_F"]]
    assert not ver.verify(bad, v) and ver.verdicts == (True, True, True, True, False)
    # a different output list is rejected
    wp2 = [list(c) for c in wp]
    wp2[0][0], wp2[0][1] = wp2[0][1], wp2[0][0]
    ver.setInstance(pkey, w, wp2)
    assert not ver.verify(rep, v)


def test_posc_honest_accepts_doubled_r_rejects():
    n = 10
    p, q, g, h, _, _, t = setup(512, n)
    pi = t.permutation(n)
    r = t.ring_array(n)
    u = P.permutation_commitment(g, h, r, pi, p)

    def run(rr):
        prover = P.PoSC(p, q, NV, NE, NR, rand=Tape(b"posc", q))
        prover.setInstance(g, h, u, rr, pi)
        e = t.int_array(n, NE)
        prover.setBatchVector(e)
        com = prover.commit()
        v = t.int_array(1, NV)[0]
        rep = prover.reply(v)
        ver = P.PoSC(p, q, NV, NE, NR)
        ver.setInstance(g, h, u)
        ver.setBatchVector(e)
        ver.setCommitment(com)
        return ver.verify(rep, v)

    # note: the commitment u = permute(h * g^r, pi) opens with r indexed like h (before permuting)
    assert run(r)
    assert not run([(x + x) % q for x in r])


@pytest.mark.parametrize("raised", [False, True])
def test_ccpos_plain_and_raised(raised):
    n = 9
    p, q, g, h, pkey, w, t = setup(512, n)
    pi = t.permutation(n)
    r = t.ring_array(n)
    u = P.permutation_commitment(g, h, r, pi, p)
    s = [t.ring_array(n)]
    wp = P.reencrypt(w, P.reenc_factors(pkey, s, p), pi, p)
    prover = P.CCPoS(p, q, NV, NE, NR, rand=t)
    prover.setInstance(g, h, u, pkey, w, wp, r, pi, s)
    e = t.int_array(n, NE)
    prover.setBatchVector(e)
    com = prover.commit()
    v = t.int_array(1, NV)[0]
    rep = prover.reply(v)
    ver = P.CCPoS(p, q, NV, NE, NR)
    ver.setInstance(g, h, u, pkey, w, wp)
    ver.setBatchVector(e)
    ver.setCommitment(com)
    if raised:
        rho = t.int_array(1, 50)[0]
        ver.computeAB(pyref.exp_scalar(u, rho, p))
        assert ver.verify(rep, v, pyref.exp_scalar(h, rho, p), rho)
        bad = dict(rep)
        bad["k_A"] = (rep["k_A"] + 1) % q
        assert not ver.verify(bad, v, pyref.exp_scalar(h, rho, p), rho)
    else:
        ver.computeAB()
        assert ver.verify(rep, v)
        bad = dict(rep)
        bad["k_B"] = [(x + 1) % q for x in rep["k_B"]]
        assert not ver.verify(bad, v)


def test_group_generic_restatement_equals_integer_restatement():
    """GPoS / GCCPoS over a ModPGroup adapter produce exactly the transcripts of the integer-only classes."""
    n, width = 9, 2
    p, q, g, h, pkey, w, t = setup(512, n, width, seed=b"generic")
    pi = t.permutation(n)
    s = [t.ring_array(n) for _ in range(width)]
    e = t.int_array(n, NE)
    v = t.int_array(1, NV)[0]
    K = P.ModPAdapter(p, q)
    a = P.PoS(p, q, NV, NE, NR, rand=Tape(b"x", q))
    b = P.GPoS(K, NV, NE, NR, rand=Tape(b"x", q))
    for o in (a, b):
        o.precompute(g, h, pi)
    wp = P.reencrypt(w, P.reenc_factors(pkey, s, p), pi, p)
    assert wp == P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi)
    for o in (a, b):
        o.setInstance(pkey, w, wp, s)
        o.setBatchVector(e)
    assert a.commit() == b.commit() and a.reply(v) == b.reply(v)


@pytest.mark.parametrize("name", ["P-256"])
def test_proofs_over_an_elliptic_curve_accept_and_reject(name):
    from oracle.pyref_ec import Curve
    c = Curve(name)
    K = P.ECAdapter(c)
    n = 6
    t = Tape(b"ecproof", c.n)
    g = c.g
    h = [c.mul(x, g) for x in t.ring_array(n)]
    y = c.mul(t.ring_element(), g)
    pkey = [g, y]
    er = t.ring_array(n)
    w = [c.exp_fixed(g, er), c.mul_arrays([c.mul(m, g) for m in t.ring_array(n)], c.exp_fixed(y, er))]
    pi = t.permutation(n)
    s = [t.ring_array(n)]
    e = t.int_array(n, NE)
    v = t.int_array(1, NV)[0]
    pr = P.GPoS(K, NV, NE, NR, rand=t)
    pr.precompute(g, h, pi)
    wp = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi)
    pr.setInstance(pkey, w, wp, s)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    ver = P.GPoS(K, NV, NE, NR)
    ver.precompute(g, h)
    ver.u = pr.u
    ver.setInstance(pkey, w, wp)
    ver.setBatchVector(e)
    ver.computeAF()
    ver.setCommitment(com)
    assert ver.verify(rep, v)
    bad = dict(rep)
    bad["k_D"] = (rep["k_D"] + 1) % c.n
    assert not ver.verify(bad, v) and ver.verdicts == (True, True, True, False, True)


def test_interactive_independent_generators_oracle():
    """Row A7 restated (distr/IndependentGeneratorsBasicI.java): honest parties accept one by one and combined; a
    wrong reply of one party fails its own check and the combined one."""
    p = pyref.find_safe_prime(512, b"vmn-test-group-512")
    q, g = (p - 1) // 2, 4
    n, thr = 25, 3
    t = Tape(b"igen-oracle", q)
    s = [None] + [t.ring_array(n) for _ in range(thr)]
    h = [None] + [pyref.exp_fixed(g, s[l], p) for l in range(1, thr + 1)]
    combined = h[1]
    for l in range(2, thr + 1):
        combined = pyref.mul(combined, h[l], p)
    e, v = t.int_array(n, 100), t.int_array(1, 100)[0]
    ver = P.IndependentGeneratorsI(p, q, 1, thr)
    ver.setInstance(g, h, None, combined)
    ver.setBatchVector(e)
    for j in range(1, thr + 1):
        o = P.IndependentGeneratorsI(p, q, j, thr, rand=Tape(b"p%d" % j, q))
        o.setInstance(g, h, s[j], combined)
        o.setBatchVector(e)
        ver.Ap[j], ver.k_a[j] = o.commit(), o.reply(v)
    assert all(ver.verify(l, v) for l in range(1, thr + 1)) and ver.verify_combined(v)
    ver.k_a[3] = (ver.k_a[3] + 1) % q
    assert not ver.verify(3, v) and not ver.verify_combined(v) and ver.verify(1, v)
