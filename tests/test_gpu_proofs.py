"""GPU suite: the HIP-backed PoS / PoSC / CCPoS drivers against the Python-integer restatement on the
same random tape: every prover message must be identical, verdicts must agree (honest accept,
tampered reject), for widths 1 and 2, in a 512-bit group (the size of the reference's own unit
test, TestPoSCBasicTW.java:69-140) and in the 2048-bit group of the benchmark."""
import pytest

from conftest import load_golden
from oracle import pyref, pyref_proofs as P
from tape import Tape

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods(entry, vmn):
    import importlib.util, os, sys
    import mirror
    out = mirror.load(entry, ("hvzk", "mixnet", "native"))
    return out


@pytest.fixture(params=["python", "native"])
def hv(request, mods):
    """The proof drivers under test: the Python mirror (hvzk.py) or the C++ drivers behind include/vmnproofs.h
    (native.py); both expose the reference's method names."""
    return mods["hvzk"] if request.param == "python" else mods["native"]


def make_instance(bits, n, width, seed):
    grp, _ = load_golden(bits)
    p, q, g = grp["p"], grp["q"], grp["g"]
    t = Tape(seed, q)
    h = pyref.exp_fixed(g, t.ring_array(n), p)
    y = pow(g, t.ring_element(), p)
    pkey = [g] * width + [y] * width
    msgs = [pyref.exp_fixed(g, t.ring_array(n), p) for _ in range(width)]
    enc_r = [t.ring_array(n) for _ in range(width)]
    w = [pyref.exp_fixed(g, enc_r[c], p) for c in range(width)] + \
        [pyref.mul(msgs[c], pyref.exp_fixed(y, enc_r[c], p), p) for c in range(width)]
    return p, q, g, h, pkey, w, t


def ints_of(x):
    return x.toInts() if hasattr(x, "toInts") else x


def same_msg(a, b):
    assert set(a) == set(b)
    for k in a:
        va, vb = a[k], b[k]
        assert ints_of(va) == ints_of(vb), k


@pytest.mark.parametrize("bits,n,width,nbits", [(512, 70, 1, (100, 100, 50)), (512, 33, 2, (100, 100, 50)),
                                                (2048, 130, 1, (256, 256, 100)), (4096, 12, 1, (256, 256, 100))])
def test_pos_transcript_matches_oracle(bits, n, width, nbits, vmn, gpu_ctx, mods, hv):
    if hv is mods["hvzk"] and bits > 2048:
        pytest.skip("the Python mirror of the drivers runs the 512- and 2048-bit shapes; the C++ drivers (the product) run every size")
    NV, NE, NR = nbits
    p, q, g, h, pkey, w, t = make_instance(bits, n, width, b"pos%d" % bits)
    pi = t.permutation(n)
    s = [t.ring_array(n) for _ in range(width)]
    e = t.int_array(n, NE)
    v = t.int_array(1, NV)[0]
    # oracle run
    o = P.PoS(p, q, NV, NE, NR, rand=Tape(b"prover", q))
    o.precompute(g, h, pi)
    wp_o = P.reencrypt(w, P.reenc_factors(pkey, s, p), pi, p)
    o.setInstance(pkey, w, wp_o, s)
    o.setBatchVector(e)
    com_o = o.commit()
    rep_o = o.reply(v)
    # HIP run on the same tape
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    H = G.toElementArray(h)
    W = [G.toElementArray(c) for c in w]
    S = [G.ringArray(c) for c in s]
    mx = mods["mixnet"]
    pr = hv.PoSBasicTW(G, NV, NE, NR, rand=Tape(b"prover", q))
    pr.precompute(g, H, pi)
    assert pr.u.toInts() == o.u and getattr(pr, "Ap", o.Ap) == o.Ap        # A' is part of the commitment (checked below)
    if hv is mods["native"]:
        WP = hv.reencrypt_native(G, pkey, W, S, pi)
    else:
        WP = mx.reencrypt(W, mx.reencFactors(G, pkey, S), pi)
    assert [c.toInts() for c in WP] == wp_o
    pr.setInstance(pkey, W, WP, S)
    pr.setBatchVector(e)
    com = pr.commit()
    same_msg(com, com_o)
    rep = pr.reply(v)
    same_msg(rep, rep_o)
    # verifier on the GPU
    ver = hv.PoSBasicTW(G, NV, NE, NR)
    ver.precompute(g, H)
    ver.setPermutationCommitment(pr.u)
    ver.setInstance(pkey, W, WP)
    ver.setBatchVector(e)
    ver.computeAF()
    ver.setCommitment(com)
    ver.setChallenge(v)
    assert ver.verify(rep)
    bad = dict(rep)
    bad["k_F"] = [(x + 1) % q for x in rep["k_F"]]
    assert not ver.verify(bad) and ver.verdicts == (True, True, True, True, False)
    kb = rep["k_B"].toInts()
    kb[n // 2] = (kb[n // 2] + 1) % q
    bad = dict(rep)
    bad["k_B"] = G.ringArray(kb)
    assert not ver.verify(bad) and ver.verdicts == (True, False, True, True, True)
    # oracle verifier accepts the GPU prover's messages
    ov = P.PoS(p, q, NV, NE, NR)
    ov.precompute(g, h)
    ov.u = pr.u.toInts()
    ov.setInstance(pkey, w, wp_o)
    ov.setBatchVector(e)
    ov.computeAF()
    ov.setCommitment({k: ints_of(x) for k, x in com.items()})
    assert ov.verify({k: ints_of(x) for k, x in rep.items()}, v)


def test_posc_and_permutation_commitment(vmn, gpu_ctx, mods, hv):
    NV, NE, NR = 100, 100, 50
    n = 100                                   # the reference's unit test size, TestPoSCBasicTW.java
    p, q, g, h, _, _, t = make_instance(512, n, 1, b"posc")
    pi = t.permutation(n)
    r = t.ring_array(n)
    e = t.int_array(n, NE)
    v = t.int_array(1, NV)[0]
    rho = t.int_array(1, 50)[0]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    H = G.toElementArray(h)
    mx = mods["mixnet"]
    pc = mx.PermutationCommitment(G, H)
    U = pc.precompute(r, pi)
    u_o = P.permutation_commitment(g, h, r, pi, p)
    assert U.toInts() == u_o
    assert mods["native"].permutation_commitment_native(G, g, H, G.ringArray(r), pi).toInts() == u_o
    assert pc.raise_(rho).toInts() == pyref.exp_scalar(u_o, rho, p)
    assert mx.raisedGenerators(H, rho).toInts() == pyref.exp_scalar(h, rho, p)

    def run(rr):
        o = P.PoSC(p, q, NV, NE, NR, rand=Tape(b"poscprover", q))
        o.setInstance(g, h, u_o, rr, pi)
        o.setBatchVector(e)
        com_o, rep_o = o.commit(), o.reply(v)
        pr = hv.PoSCBasicTW(G, NV, NE, NR, rand=Tape(b"poscprover", q))
        pr.setInstance(g, H, U, G.ringArray(rr), pi)
        pr.setBatchVector(e)
        com = pr.commit()
        rep = pr.reply(v)
        same_msg(com, com_o)
        same_msg(rep, rep_o)
        ver = hv.PoSCBasicTW(G, NV, NE, NR)
        ver.setInstance(g, H, U)
        ver.setBatchVector(e)
        ver.setCommitment(com)
        ver.setChallenge(v)
        ok = ver.verify(rep)
        if hv is mods["native"]:
            # the verifier's intermediates against the oracle's (private fields A, C, D of PoSCBasicTW.java:676, 718-727)
            ov = P.PoSC(p, q, NV, NE, NR)
            ov.setInstance(g, h, u_o)
            ov.setBatchVector(e)
            ov.setCommitment({k: (x.toInts() if hasattr(x, "toInts") else x) for k, x in com.items()})
            assert ov.verify({k: (x.toInts() if hasattr(x, "toInts") else x) for k, x in rep.items()}, v) == ok
            assert (ver.getA(), ver.getC(), ver.getD()) == (ov.A, ov.C, ov.D)
        return ok

    assert run(r)
    assert not run([(x + x) % q for x in r])          # TestPoSCBasicTW.java:109-111


@pytest.mark.parametrize("raised", [False, True])
def test_ccpos_transcript_matches_oracle(raised, vmn, gpu_ctx, mods, hv):
    NV, NE, NR = 256, 256, 100
    n, width = 130, 1
    p, q, g, h, pkey, w, t = make_instance(2048, n, width, b"ccpos")
    pi = t.permutation(n)
    r = t.ring_array(n)
    s = [t.ring_array(n)]
    e = t.int_array(n, NE)
    v = t.int_array(1, NV)[0]
    rho = t.int_array(1, 50)[0]
    u_o = P.permutation_commitment(g, h, r, pi, p)
    wp_o = P.reencrypt(w, P.reenc_factors(pkey, s, p), pi, p)
    o = P.CCPoS(p, q, NV, NE, NR, rand=Tape(b"ccprover", q))
    o.setInstance(g, h, u_o, pkey, w, wp_o, r, pi, s)
    o.setBatchVector(e)
    com_o, rep_o = o.commit(), o.reply(v)
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    H, U = G.toElementArray(h), G.toElementArray(u_o)
    W = [G.toElementArray(c) for c in w]
    WP = [G.toElementArray(c) for c in wp_o]
    pr = hv.CCPoSBasicW(G, NV, NE, NR, rand=Tape(b"ccprover", q))
    pr.setInstance(g, H, U, pkey, W, WP, G.ringArray(r), pi, [G.ringArray(s[0])])
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    same_msg(com, com_o)
    same_msg(rep, rep_o)
    ver = hv.CCPoSBasicW(G, NV, NE, NR)
    ver.setInstance(g, H, U, pkey, W, WP)
    ver.setBatchVector(e)
    ver.setCommitment(com)
    ver.setChallenge(v)
    if raised:
        ver.computeAB(U.exp(rho))
        RH = H.exp(rho)
        assert ver.verify(rep, RH, rho)
        bad = dict(rep)
        bad["k_A"] = (rep["k_A"] + 1) % q
        assert not ver.verify(bad, RH, rho)
    else:
        ver.computeAB()
        assert ver.verify(rep)
        bad = dict(rep)
        bad["k_B"] = [(x + 1) % q for x in rep["k_B"]]
        assert not ver.verify(bad)


def test_proof_level_abi_reports_misuse_and_malformed_messages(vmn, gpu_ctx, mods):
    """Error conventions of SURVEY.md §8b at the proof level: misuse comes back as a status (VmnError), a commitment
    holding a value that is not a group element as a format error the caller can catch (PoSBasicTW.java:794-815),
    a wrong reply as verdict False -- never a crash."""
    nat = mods["native"]
    NV, NE, NR, n = 100, 100, 50, 16
    p, q, g, h, pkey, w, t = make_instance(512, n, 1, b"misuse")
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    H, W = G.toElementArray(h), [G.toElementArray(c) for c in w]
    pi, s, e, v = t.permutation(n), [t.ring_array(n)], t.int_array(n, NE), t.int_array(1, NV)[0]
    S = [G.ringArray(s[0])]
    pr = nat.PoSBasicTW(G, NV, NE, NR, rand=Tape(b"m", q))
    with pytest.raises(vmn.VmnError):
        pr.commit()                                            # before precompute / instance / batching vector
    with pytest.raises(vmn.VmnError):
        pr.precompute(g, H, [0] * n)                           # not a permutation
    pr.precompute(g, H, pi)
    WP = nat.reencrypt_native(G, pkey, W, S, pi)
    short = G.toElementArray(w[0][:-1])
    with pytest.raises(vmn.VmnError):
        pr.setInstance(pkey, [short, W[1]], WP, S)             # component of the wrong size
    with pytest.raises(vmn.VmnError):
        nat.PoSBasicTW(G, NV, NE, NR).commit()                 # a verifier object has no random source
    pr.setInstance(pkey, W, WP, S)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    ver = nat.PoSBasicTW(G, NV, NE, NR)
    ver.precompute(g, H)
    ver.setPermutationCommitment(pr.u)
    ver.setInstance(pkey, W, WP)
    ver.setBatchVector(e)
    with pytest.raises(vmn.VmnError):
        ver.verify(rep)                                        # before computeAF / setCommitment / setChallenge
    ver.computeAF()
    bad = dict(com)
    bad["Cp"] = p                                              # not a group element (>= p)
    with pytest.raises(vmn.VmnError) as ei:
        ver.setCommitment(bad)
    assert ei.value.status == -4                               # VMN_ERR_FORMAT: the caller substitutes trivial values
    bad = dict(com)
    bad["Dp"] = p - 1                                          # in range but outside the subgroup of order q
    with pytest.raises(vmn.VmnError) as ei:
        ver.setCommitment(bad)
    assert ei.value.status == -4
    ver.setCommitment(com)
    ver.setChallenge(v)
    assert ver.verify(rep)
    wrong = dict(rep)
    wrong["k_A"] = (rep["k_A"] + 1) % q
    assert not ver.verify(wrong) and ver.verdicts == (False, True, True, True, True)
    truncated = dict(rep)
    truncated["k_F"] = []                                      # wrong shape: status, not a crash
    with pytest.raises(vmn.VmnError):
        ver.verify(truncated)


def test_two_party_threads_with_their_own_contexts(vmn, mods):
    """vdemo runs k parties as threads of one JVM (demo/Demo.java:282-291): two threads, each with its own context and
    group, prove and verify concurrently through the C++ drivers; both transcripts equal the oracle's."""
    import threading
    nat = mods["native"]
    NV, NE, NR, n = 100, 100, 50, 400
    p, q, g, h, _, _, t = make_instance(512, n, 1, b"threads")
    pi, r, e, v = t.permutation(n), t.ring_array(n), t.int_array(n, NE), t.int_array(1, NV)[0]
    u = P.permutation_commitment(g, h, r, pi, p)
    want = {}
    for party in (1, 2):
        o = P.PoSC(p, q, NV, NE, NR, rand=Tape(b"party%d" % party, q))
        o.setInstance(g, h, u, r, pi)
        o.setBatchVector(e)
        want[party] = (o.commit(), o.reply(v))
    got, errors = {}, []

    def party_thread(party):
        try:
            ctx = vmn.Context(0)
            G = vmn.ModPGroup(ctx, p, q, g)
            H, U, R = G.toElementArray(h), G.toElementArray(u), G.ringArray(r)
            for rep_no in range(3):
                pr = nat.PoSCBasicTW(G, NV, NE, NR, rand=Tape(b"party%d" % party, q))
                pr.setInstance(g, H, U, R, pi)
                pr.setBatchVector(e)
                com, rep = pr.commit(), pr.reply(v)
                ver = nat.PoSCBasicTW(G, NV, NE, NR)
                ver.setInstance(g, H, U)
                ver.setBatchVector(e)
                ver.setCommitment(com)
                ver.setChallenge(v)
                ok = ver.verify(rep)
                got[(party, rep_no)] = ({k: ints_of(x) for k, x in com.items()}, {k: ints_of(x) for k, x in rep.items()}, ok)
                ver.free()
                pr.free()
        except Exception as exc:      # pragma: no cover
            errors.append(exc)

    threads = [threading.Thread(target=party_thread, args=(party,)) for party in (1, 2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for party in (1, 2):
        for rep_no in range(3):
            com, rep, ok = got[(party, rep_no)]
            assert ok and (com, rep) == want[party]


def test_interactive_independent_generators(vmn, gpu_ctx, mods, hv):
    """SURVEY.md §8a row A7 (distr/IndependentGeneratorsBasicI.java): every party proves knowledge of the exponents
    of its generator parts; the per-party and the combined checks accept, a wrong reply is rejected; commitments and
    replies equal the oracle's."""
    from oracle import pyref_prg
    NE, NV, n, thr = 100, 100, 60, 3
    grp, _ = load_golden(512)
    p, q, g = grp["p"], grp["q"], grp["g"]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    t = Tape(b"igen", q)
    s = [None] + [t.ring_array(n) for _ in range(thr)]
    h = [None] + [pyref.exp_fixed(g, s[l], p) for l in range(1, thr + 1)]
    combined = h[1]
    for l in range(2, thr + 1):
        combined = pyref.mul(combined, h[l], p)
    seed = pyref_prg.random_oracle(b"igen-seed", 256)
    e = pyref_prg.random_integers(seed, n, NE)
    v = t.int_array(1, NV)[0]
    H = [None] + [G.toElementArray(h[l]) for l in range(1, thr + 1)]
    CH = G.toElementArray(combined)
    ver = hv.IndependentGeneratorsBasicI(G, 1, thr, NE)
    ver.setInstance(g, H, None, CH)
    ver.setBatchVectorSeed(seed)
    ver.setChallenge(v)
    over = P.IndependentGeneratorsI(p, q, 1, thr)
    over.setInstance(g, h, None, combined)
    over.setBatchVector(e)
    for j in range(1, thr + 1):
        o = P.IndependentGeneratorsI(p, q, j, thr, rand=Tape(b"igen-party%d" % j, q))
        o.setInstance(g, h, s[j], combined)
        o.setBatchVector(e)
        Ap_o, ka_o = o.commit(), o.reply(v)
        pr = hv.IndependentGeneratorsBasicI(G, j, thr, NE, rand=Tape(b"igen-party%d" % j, q))
        pr.setInstance(g, H, G.ringArray(s[j]), CH)
        pr.setBatchVectorSeed(seed)
        Ap = pr.commit()
        pr.setChallenge(v)
        ka = pr.reply()
        assert (Ap, ka) == (Ap_o, ka_o)
        ver.setCommitment(j, Ap)
        ver.setReply(j, ka)
        over.Ap[j], over.k_a[j] = Ap, ka
    assert all(ver.verify(l) for l in range(1, thr + 1)) and ver.verify()
    assert all(over.verify(l, v) for l in range(1, thr + 1)) and over.verify_combined(v)
    ver.setReply(2, (over.k_a[2] + 1) % q)
    assert not ver.verify(2) and not ver.verify() and ver.verify(1)


def test_verify_prepare_is_the_reply_side_of_verify(vmn, gpu_ctx, mods):
    """vmn_pos_verify_prepare: the part of verify() that needs no challenge, run ahead (while the challenge is hashed);
    the verdicts are those of verify() alone, and a verifier prepared for one reply recomputes for another."""
    hv = mods["native"]
    NV, NE, NR = 256, 256, 100
    bits, n, width = 2048, 40, 1
    p, q, g, h, pkey, w, t = make_instance(bits, n, width, b"prep")
    pi, s, e, v = t.permutation(n), [t.ring_array(n)], t.int_array(n, NE), t.int_array(1, NV)[0]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    H, W, S = G.toElementArray(h), [G.toElementArray(c) for c in w], [G.ringArray(c) for c in s]
    pr = hv.PoSBasicTW(G, NV, NE, NR, rand=Tape(b"prep-prover", q))
    pr.precompute(g, H, pi)
    WP = hv.reencrypt_native(G, pkey, W, S, pi)
    pr.setInstance(pkey, W, WP, S)
    pr.setBatchVector(e)
    com = pr.commit()
    rep = pr.reply(v)

    def verifier():
        ver = hv.PoSBasicTW(G, NV, NE, NR)
        ver.precompute(g, H)
        ver.setPermutationCommitment(pr.u)
        ver.setInstance(pkey, W, WP)
        ver.setBatchVector(e)
        ver.computeAF()
        ver.setCommitment(com.native)
        return ver
    plain = verifier()
    plain.setChallenge(v)
    assert plain.verify(rep.native) and plain.verdicts == (True,) * 5
    ver = verifier()
    ver.verifyPrepare(rep.native)                      # before the challenge is known
    ver.setChallenge(v)
    assert ver.verify(rep.native) and ver.verdicts == (True,) * 5
    ver.verifyPrepare(rep.native)
    bad = dict(rep)
    bad["k_F"] = [(x + 1) % q for x in rep["k_F"]]
    assert not ver.verify(bad) and ver.verdicts == (True, True, True, True, False)      # another reply: recomputed, not reused
    assert ver.verify(rep.native)
    ver.setChallenge((v + 1) % (1 << NV))              # the prepared part does not depend on the challenge; the verdict does
    ver.verifyPrepare(rep.native)
    assert not ver.verify(rep.native)


@pytest.mark.parametrize("raised", [False, True])
def test_verify_prepare_of_posc_and_ccpos(raised, vmn, gpu_ctx, mods):
    """vmn_posc_verify_prepare / vmn_ccpos_verify_prepare: the same verdicts as verify() alone, honest and tampered."""
    hv, mx = mods["native"], mods["mixnet"]
    NV, NE, NR = 256, 256, 100
    bits, n, width = 2048, 30, 2
    p, q, g, h, pkey, w, t = make_instance(bits, n, width, b"prep2")
    pi, r, s = t.permutation(n), t.ring_array(n), [t.ring_array(n) for _ in range(width)]
    e, v = t.int_array(n, NE), t.int_array(1, NV)[0]
    rho = t.int_array(1, mx.RAISED_BITLENGTH)[0]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    H, W, S, R = G.toElementArray(h), [G.toElementArray(c) for c in w], [G.ringArray(c) for c in s], G.ringArray(r)
    U = G.toElementArray(P.permutation_commitment(g, h, r, pi, p))
    WP = hv.reencrypt_native(G, pkey, W, S, pi)
    # PoSC
    pr = hv.PoSCBasicTW(G, NV, NE, NR, rand=Tape(b"prep2-posc", q))
    pr.setInstance(g, H, U, R, pi)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    for tamper in (False, True):
        ver = hv.PoSCBasicTW(G, NV, NE, NR)
        ver.setInstance(g, H, U)
        ver.setBatchVector(e)
        ver.setCommitment(com.native)
        ver.verifyPrepare(rep.native)
        ver.setChallenge(v if not tamper else v ^ 1)
        assert ver.verify(rep.native) is (not tamper)
    # CCPoS, plain or raised
    cp = hv.CCPoSBasicW(G, NV, NE, NR, rand=Tape(b"prep2-ccpos", q))
    cp.setInstance(g, H, U, pkey, W, WP, R, pi, S)
    cp.setBatchVector(e)
    com2, rep2 = cp.commit(), cp.reply(v)
    RU, RH = (U.exp(rho), H.exp(rho)) if raised else (None, None)
    for tamper in (False, True):
        cv = hv.CCPoSBasicW(G, NV, NE, NR)
        cv.setInstance(g, H, U, pkey, W, WP)
        cv.setBatchVector(e)
        cv.setCommitment(com2.native)
        cv.computeAB(RU)
        cv.verifyPrepare(rep2.native, RH, rho if raised else None)
        cv.setChallenge(v if not tamper else v ^ 1)
        assert cv.verify(rep2.native, RH, rho if raised else None) is (not tamper)


def test_check_b_as_one_simultaneous_power_gives_the_same_verdicts(vmn, gpu_ctx, mods, monkeypatch):
    """Large arrays verify check (B) in the form (B_i^v (B_{i-1}^{-1})^{k_E,i}) B'_i = g^{k_B,i} (vmn_garray_exp2 +
    vmn_garray_inv); here at a small size (VMN_COMBINED_MIN=1) beside the separate form: honest and tampered replies and
    commitments, PoS and PoSC, and a commitment with the residue 0 in B (no inverse: the separate form takes over)."""
    hv = mods["native"]
    NV, NE, NR = 256, 256, 100
    bits, n, width = 2048, 45, 1
    p, q, g, h, pkey, w, t = make_instance(bits, n, width, b"comb")
    pi, r, s = t.permutation(n), t.ring_array(n), [t.ring_array(n)]
    e, v = t.int_array(n, NE), t.int_array(1, NV)[0]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    H, W, S, R = G.toElementArray(h), [G.toElementArray(c) for c in w], [G.ringArray(c) for c in s], G.ringArray(r)
    pr = hv.PoSBasicTW(G, NV, NE, NR, rand=Tape(b"comb-prover", q))
    pr.precompute(g, H, pi)
    WP = hv.reencrypt_native(G, pkey, W, S, pi)
    pr.setInstance(pkey, W, WP, S)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    U = G.toElementArray(P.permutation_commitment(g, h, r, pi, p))
    pc = hv.PoSCBasicTW(G, NV, NE, NR, rand=Tape(b"comb-posc", q))
    pc.setInstance(g, H, U, R, pi)
    pc.setBatchVector(e)
    com_c, rep_c = pc.commit(), pc.reply(v)

    def tampered(msg, key, pos, delta, mod):
        out = dict(msg)
        vals = msg[key].toInts()
        vals[pos] = (vals[pos] + delta) % mod if delta else 0
        out[key] = G.ringArray(vals) if mod == q else G.toElementArray(vals, checked=False)
        return out
    cases = [(com, rep), (com, tampered(rep, "k_B", 7, 1, q)), (com, tampered(rep, "k_E", n - 1, 1, q)),
             (tampered(com, "B", 3, 1, p), rep), (tampered(com, "Bp", 0, 5, p), rep),
             (tampered(com, "B", 11, 0, p), rep), (tampered(com, "B", n - 1, 0, p), rep)]       # the last two: a zero in B
    verdicts = {}
    for mode in ("separate", "paired", "combined"):                    # paired: both powers of the separate form in one launch
        monkeypatch.setenv("VMN_COMBINED_MIN", "1" if mode == "combined" else "1000000000")
        monkeypatch.setenv("VMN_PAIR_MAX", "0" if mode == "separate" else "131072")
        out = []
        for c, rp in cases:
            ver = hv.PoSBasicTW(G, NV, NE, NR)
            ver.precompute(g, H)
            ver.setPermutationCommitment(pr.u)
            ver.setInstance(pkey, W, WP)
            ver.setBatchVector(e)
            ver.computeAF()
            ver.setCommitment(c)
            ver.setChallenge(v)
            ok = ver.verify(rp)
            out.append((ok, ver.verdicts))
        for c, rp in [(com_c, rep_c), (com_c, tampered(rep_c, "k_B", 2, 1, q)), (tampered(com_c, "B", 5, 0, p), rep_c)]:
            ver = hv.PoSCBasicTW(G, NV, NE, NR)
            ver.setInstance(g, H, U)
            ver.setBatchVector(e)
            ver.setCommitment(c)
            ver.setChallenge(v)
            out.append((ver.verify(rp), None))
        verdicts[mode] = out
    assert verdicts["separate"] == verdicts["combined"] == verdicts["paired"]
    assert verdicts["combined"][0] == (True, (True,) * 5) and verdicts["combined"][1] == (False, (True, False, True, True, True))
    assert [ok for ok, _ in verdicts["combined"][2:7]] == [False] * 5
    assert [ok for ok, _ in verdicts["combined"][7:]] == [True, False, False]
