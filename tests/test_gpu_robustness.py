"""GPU suite: behaviour at the edges of the proof-level ABI that a transcript comparison does not reach -- a cached
reply side never survives a change of the verifier's inputs, ring scalars of a reply must be field elements
(PoSBasicTW.java:985-989: pRing.toElement), a block freed by the other lane's thread returns to the lane that owns it."""
import threading

import pytest

from oracle import pyref_proofs as P
from tape import Tape
from test_gpu_proofs import make_instance, mods  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
NV, NE, NR = 100, 100, 50


def _pos_case(vmn, gpu_ctx, nat, n=24, seed=b"robust"):
    p, q, g, h, pkey, w, t = make_instance(512, n, 1, seed)
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    H, W = G.toElementArray(h), [G.toElementArray(c) for c in w]
    pi, s, e, v = t.permutation(n), [t.ring_array(n)], t.int_array(n, NE), t.int_array(1, NV)[0]
    S = [G.ringArray(s[0])]
    pr = nat.PoSBasicTW(G, NV, NE, NR, rand=Tape(seed + b"-prover", q))
    pr.precompute(g, H, pi)
    WP = nat.reencrypt_native(G, pkey, W, S, pi)
    pr.setInstance(pkey, W, WP, S)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)

    def verifier(commitment=com):
        ver = nat.PoSBasicTW(G, NV, NE, NR)
        ver.precompute(g, H)
        ver.setPermutationCommitment(pr.u)
        ver.setInstance(pkey, W, WP)
        ver.setBatchVector(e)
        ver.computeAF()
        ver.setCommitment(commitment)
        return ver
    return dict(G=G, p=p, q=q, g=g, H=H, W=W, WP=WP, pkey=pkey, e=e, v=v, com=com, rep=rep, verifier=verifier, pr=pr, keep=(S, pr))


def test_a_prepared_reply_side_does_not_survive_new_inputs(vmn, gpu_ctx, mods):
    """verify_prepare(reply) caches D = B_last / h0^prod(e), the right side of check (B) and the k_E products; a new
    commitment / batching vector / instance after it must not be combined with them."""
    nat = mods["native"]
    c = _pos_case(vmn, gpu_ctx, nat)
    G, q, p, com, rep, v = c["G"], c["q"], c["p"], c["com"], c["rep"], c["v"]
    bad_com = dict(com)
    b = com["B"].toInts()
    b[-1] = b[-1] * c["g"] % p                                   # another last B: D changes, check (B) fails at the last position
    bad_com["B"] = G.toElementArray(b)
    # prepared with the honest commitment, then the commitment is replaced: the verdict is the one of the new commitment
    ver = c["verifier"]()
    ver.verifyPrepare(rep.native)
    ver.setCommitment(bad_com)
    ver.setChallenge(v)
    assert not ver.verify(rep.native) and ver.verdicts == (True, False, True, False, True)
    # ... and the other way round
    ver = c["verifier"](bad_com)
    ver.verifyPrepare(rep.native)
    ver.setCommitment(com)
    ver.setChallenge(v)
    assert ver.verify(rep.native)
    # a new batching vector after the preparation: A, F and prod(e) are recomputed for it
    ver = c["verifier"]()
    ver.verifyPrepare(rep.native)
    e2 = list(c["e"])
    e2[0] ^= 1
    ver.setBatchVector(e2)
    ver.computeAF()
    ver.setChallenge(v)
    assert not ver.verify(rep.native)


def test_ccpos_prepared_state_is_bound_to_the_raised_arguments(vmn, gpu_ctx, mods):
    """vmn_ccpos_verify(reply, raisedh, rho) after vmn_ccpos_verify_prepare with OTHER raised generators must not
    silently use the prepared ones."""
    hv, mx = mods["native"], mods["mixnet"]
    n, width = 20, 1
    p, q, g, h, pkey, w, t = make_instance(512, n, width, b"robust-cc")
    pi, r, s = t.permutation(n), t.ring_array(n), [t.ring_array(n)]
    e, v = t.int_array(n, NE), t.int_array(1, NV)[0]
    rho = t.int_array(1, mx.RAISED_BITLENGTH)[0]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    H, W, S, R = G.toElementArray(h), [G.toElementArray(c) for c in w], [G.ringArray(c) for c in s], G.ringArray(r)
    U = G.toElementArray(P.permutation_commitment(g, h, r, pi, p))
    WP = hv.reencrypt_native(G, pkey, W, S, pi)
    cp = hv.CCPoSBasicW(G, NV, NE, NR, rand=Tape(b"robust-ccpos", q))
    cp.setInstance(g, H, U, pkey, W, WP, R, pi, S)
    cp.setBatchVector(e)
    com, rep = cp.commit(), cp.reply(v)
    RU, RH = U.exp(rho), H.exp(rho)
    wrong_RH = H.exp(rho + 1)
    cv = hv.CCPoSBasicW(G, NV, NE, NR)
    cv.setInstance(g, H, U, pkey, W, WP)
    cv.setBatchVector(e)
    cv.setCommitment(com.native)
    cv.computeAB(RU)
    cv.setChallenge(v)
    cv.verifyPrepare(rep.native, RH, rho)
    assert not cv.verify(rep.native, wrong_RH, rho)             # other generators than the prepared ones: recomputed, rejected
    cv.verifyPrepare(rep.native, wrong_RH, rho)
    assert cv.verify(rep.native, RH, rho)
    cv.verifyPrepare(rep.native, RH, rho)
    assert not cv.verify(rep.native, RH, rho + 1)


def test_reply_scalars_must_be_below_q(vmn, gpu_ctx, mods):
    """k_A + q is the same exponent but not the same field element: the reference's parser (pRing.toElement) refuses it,
    so this verifier must too -- through the byte-tree reader and for a message assembled by hand."""
    nat = mods["native"]
    c = _pos_case(vmn, gpu_ctx, nat, seed=b"robust-q")
    G, q, rep, v, n = c["G"], c["q"], c["rep"], c["v"], 24
    assert rep["k_A"] + q < 1 << (8 * G.exp_bytes)
    ver = c["verifier"]()
    ver.setChallenge(v)
    assert ver.verify(rep)
    for key in ("k_A", "k_C", "k_D"):
        bad = dict(rep)
        bad[key] = rep[key] + q
        assert not ver.verify(bad) and ver.verdicts == (False,) * 5, key
    bad = dict(rep)
    bad["k_F"] = [rep["k_F"][0] + q]
    assert not ver.verify(bad)
    assert ver.verify(rep)
    # the wire: a reply whose k_A leaf holds k_A + q is not a reply
    bt = bytearray(rep.native.toByteTree())
    ka = rep["k_A"].to_bytes(G.exp_bytes, "big")
    at = bytes(bt).index(ka)
    bt[at:at + G.exp_bytes] = (rep["k_A"] + q).to_bytes(G.exp_bytes, "big")
    assert ver.readReply(bytes(bt), n, 1) is None
    assert ver.readReply(rep.native.toByteTree(), n, 1) is not None
    # PoSC and CCPoS share the rule
    hv = nat
    p, g = c["p"], c["g"]
    t = Tape(b"robust-q2", q)
    h = c["H"].toInts()
    pi, r, e = t.permutation(n), t.ring_array(n), t.int_array(n, NE)
    U = G.toElementArray(P.permutation_commitment(g, h, r, pi, p))
    pc = hv.PoSCBasicTW(G, NV, NE, NR, rand=Tape(b"robust-posc", q))
    pc.setInstance(g, c["H"], U, G.ringArray(r), pi)
    pc.setBatchVector(e)
    com_c, rep_c = pc.commit(), pc.reply(v)
    vc = hv.PoSCBasicTW(G, NV, NE, NR)
    vc.setInstance(g, c["H"], U)
    vc.setBatchVector(e)
    vc.setCommitment(com_c)
    vc.setChallenge(v)
    assert vc.verify(rep_c)
    bad = dict(rep_c)
    bad["k_D"] = rep_c["k_D"] + q
    assert not vc.verify(bad)


def test_decryption_reply_above_q_is_a_false_verdict(vmn, gpu_ctx, mods):
    """DistrElGamalSessionBasic.setReply (:606-613): a reply that is not a field element becomes 0 and the party's verdict
    false -- it is not reduced."""
    nat = mods["native"]
    n, k, thr = 12, 3, 2
    p, q, g, h, pkey, w, t = make_instance(512, n, 1, b"robust-dec")
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    U = G.toElementArray(w[0])
    x = t.ring_element()
    y = pow(g, x, p)
    e, v = t.int_array(n, NE), t.int_array(1, NV)[0]
    F = nat.decryptionFactors(U, x, q, k)
    pr = nat.DistrElGamalSessionBasic(G, 1, k, thr, NE, rand=Tape(b"robust-dec-p", q))
    ys, fs = [None, y] + [y] * (k - 1), [None, F] + [None] * (k - 1)
    pr.setInstance(U, ys, fs)
    pr.setBatchVector(e)
    pr.batchInput()
    yp, Bp = pr.commit(x)
    kx = pr.reply(v)
    ver = nat.DistrElGamalSessionBasic(G, 2, k, thr, NE)
    ver.setInstance(U, ys, fs)
    ver.setBatchVector(e)
    ver.batchInput()
    ver.setCommitment(1, yp, Bp)
    ver.batch(1)
    ver.setReply(1, kx)
    assert ver.verify(1, v)
    ver.setReply(1, kx + q)
    assert not ver.verify(1, v)
    ver.setReply(1, kx)
    assert ver.verify(1, v)


def test_an_array_freed_by_the_helper_returns_to_the_lane_that_owns_it(vmn, gpu_ctx):
    """vmn_garray_free on the helper thread used to put a main-lane block into the HELPER's pool while
    the main stream still had kernels queued on it; the helper's next allocation of that size then overwrote it on the
    other stream.  Now the block goes back to its own lane (ordered behind the freeing thread's queued work)."""
    from oracle import pyref
    p, q, g = pyref.modp_group(2048)
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    n = 4096
    xs = pyref.exp_fixed(g, [k + 2 for k in range(n)], p)
    es = pyref.stream_ints(b"lane/e", n, q)
    X, E = G.toElementArray(xs), G.ringArray(es)
    live0 = gpu_ctx.memory_stats()["live_bytes"]
    gpu_ctx.helper_mark()
    errors = []
    junk = [p - 1 - k for k in range(n)]

    def helper_thread(victim):
        try:
            with gpu_ctx.helper():
                victim.free()                                   # the last reference dropped on the helper thread
                for _ in range(4):                              # same size class: would have been handed the block back
                    J = G.toElementArray(junk, checked=False)
                    J.free()
        except Exception as exc:      # pragma: no cover
            errors.append(exc)

    outs = []
    for rnd in range(3):
        Xc = X.copyOfRange(0, n)
        out = Xc.exp(E)                                         # queued on the main stream, reads Xc for ~ms
        th = threading.Thread(target=helper_thread, args=(Xc,))
        th.start()
        th.join()
        outs.append(out)
    assert not errors
    want = [pow(x, e, p) for x, e in zip(xs[:64], es[:64])]
    first = outs[0].toInts()
    for out in outs:
        got = out.toInts()
        assert got[:64] == want and got == first
        out.free()
    assert gpu_ctx.memory_stats()["live_bytes"] == live0        # the accounting of neither lane went negative / leaked


def test_mutated_byte_trees_never_crash_the_parsers(vmn, gpu_ctx, mods):
    """The byte trees of commitments, replies and arrays arrive from the bulletin board: whatever they hold, the readers must
    answer "not a message" (None / ValueError) or hand back something the verifier then judges -- never fault, hang or read
    past the buffer.  A few hundred seeded mutations of valid trees: truncations, flipped bytes, rewritten length fields
    (also to huge values), swapped tags, appended garbage."""
    import random
    nat = mods["native"]
    c = _pos_case(vmn, gpu_ctx, nat, n=20, seed=b"robust-fuzz")
    G, rep, com, v, n = c["G"], c["rep"], c["com"], c["v"], 20
    ver = c["verifier"]()
    ver.setChallenge(v)
    good_rep, good_com = rep.native.toByteTree(), com.native.toByteTree()
    good_arr = com["B"].toByteTree()
    rnd = random.Random(20261004)

    def mutate(bt):
        b = bytearray(bt)
        kind = rnd.randrange(7)
        if kind == 0:
            return bytes(b[:rnd.randrange(len(b))])                           # truncated
        if kind == 1:
            for _ in range(rnd.randrange(1, 4)):
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)             # flipped bits
            return bytes(b)
        if kind == 2:                                                          # a length field rewritten
            pos = rnd.choice([1, 6, 11] + [rnd.randrange(len(b) - 4)])
            b[pos:pos + 4] = rnd.choice([bytes([255] * 4), bytes([127, 255, 255, 255]), bytes(4),
                                         rnd.randrange(1 << 32).to_bytes(4, "big")])
            return bytes(b)
        if kind == 3:
            b[rnd.choice([0, 5, 10])] ^= 1                                     # node <-> leaf tag
            return bytes(b)
        if kind == 4:
            return bytes(b) + bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 40)))   # trailing garbage
        if kind == 5:
            i = rnd.randrange(len(b) - 8)
            return bytes(b[:i] + b[i + rnd.randrange(1, 8):])                  # a few bytes cut out of the middle
        return bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 64)))  # noise

    accepted = parsed = 0
    for _ in range(150):
        m = ver.readReply(mutate(good_rep), n, 1)
        if m is not None:
            parsed += 1
            accepted += bool(ver.verify(m))
    assert accepted <= parsed                                                  # (a flip inside padding can leave a valid reply)
    for _ in range(80):
        m = ver.readCommitment(mutate(good_com), n, 1)
        if m is not None:
            try:
                ver.setCommitment(m)
            except vmn.VmnError as exc:
                assert exc.status == -4                                        # not group elements: the caller's trivial-value path
    for _ in range(80):
        try:
            a = G.toElementArrayFromByteTree(mutate(good_arr))
            assert a.size() >= 0
        except (ValueError, vmn.VmnError):
            pass
    # the verifier still works afterwards
    ver2 = c["verifier"]()
    ver2.setChallenge(v)
    assert ver2.verify(ver2.readReply(good_rep, n, 1))


def test_round_three_entry_points_at_their_edges(vmn, gpu_ctx, mods):
    """The calls added for the small-array work take the ordinary path where their fast one does not apply, and refuse what
    does not fit: the paired powers over a curve and over empty arrays, several inner products with a mismatched pair, the
    precomputed re-encryption factors with arrays of different length, a verifier freed with its A / F still in flight."""
    nat = mods["native"]
    from oracle.pyref_ec import Curve
    c = Curve("P-256")
    E = vmn.ECqPGroup(gpu_ctx, "P-256")
    pts = [c.mul(k + 2, c.g) for k in range(9)]
    es = [3 * k + 1 for k in range(9)]
    gx, gy = E.toElementArray(pts).expPair(5, E.toElementArray(pts[::-1]), E.ringArray(es))      # curves: two ordinary launches
    assert gx.toInts() == [c.mul(5, P_) for P_ in pts] and gy.toInts() == [c.mul(e, P_) for e, P_ in zip(es, pts[::-1])]
    case = _pos_case(vmn, gpu_ctx, nat, n=24, seed=b"edges")
    G, q, p = case["G"], case["q"], case["p"]
    none = G.toElementArray([])
    ex, ey = none.expPair(7, G.toElementArray([]), G.ringArray([]))
    assert ex.toInts() == [] and ey.toInts() == []
    xs = [pow(case["g"], k + 1, p) for k in range(5)]
    ex, ey = G.toElementArray(xs).expPair(0, none, G.ringArray([]))                             # one side empty
    assert ex.toInts() == [1] * 5 and ey.toInts() == []
    a, b = G.ringArray([1, 2, 3]), G.ringArray([4, 5])
    with pytest.raises(vmn.VmnError):
        vmn.innerProducts([(a, a), (a, b)])
    assert vmn.innerProducts([(a, a), (b, None)]) == [14, 9]
    W, S = case["W"], case["keep"][0]
    F = nat.reencryption_factors_native(G, case["pkey"], S)
    pi = list(range(24))
    with pytest.raises(vmn.VmnError):
        nat.apply_factors_native(G, W, [f.copyOfRange(0, 23) for f in F], pi)
    with pytest.raises(vmn.VmnError):
        nat.apply_factors_native(G, W, F, [0] * 24)                                             # not a permutation
    assert [x.toInts() for x in nat.apply_factors_native(G, W, F, pi)] == [x.toInts() for x in nat.reencrypt_native(G, case["pkey"], W, S, pi)]
    ver = case["verifier"]()                                                                    # computeAF has begun A and F ...
    ver.free()                                                                                  # ... and the handle goes with the verifier
    again = case["verifier"]()
    again.setChallenge(case["v"])
    assert again.verify(case["rep"])                                                            # the landing buffers were released
