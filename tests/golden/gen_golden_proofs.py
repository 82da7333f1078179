#!/usr/bin/env python3
"""Generate the committed golden vectors for the curve groups and for whole proof transcripts
(tests/golden/ec_p224.json, ec_p256.json, ec_p384.json, ec_p521.json, proofs_n8.json) from the Python restatements only (oracle/pyref_ec.py:
affine textbook arithmetic; oracle/pyref_proofs.py: the proofs over Python integers / affine points).

The reference holds no known-answer vectors for this path; these fixtures are what SURVEY.md §8c
"golden vectors to create" lists: per-operation vectors for P-256 / P-384 with the exceptional cases, and full
PoS / PoSC / CCPoS transcripts at N = 8 with an explicit random tape (tests/tape.py), expected = the oracle's
messages and verdict.  Re-run:  python tests/golden/gen_golden_proofs.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyref, pyref_proofs as P          # noqa: E402
from oracle.pyref_ec import Curve                    # noqa: E402
from tape import Tape                                # noqa: E402


def hx(v):
    return format(v, "x")


def pt(Q):
    return None if Q is None else [hx(Q[0]), hx(Q[1])]


def pts(X):
    return [pt(Q) for Q in X]


def ec_cases(name, sizes):
    c = Curve(name)
    n_ord, g = c.n, c.g
    cases = []
    for n in sizes:
        s = f"{name}/{n}/".encode()
        xs = [c.mul(k, g) for k in pyref.stream_ints(s + b"x", n, n_ord)]
        ys = [c.mul(k, g) for k in pyref.stream_ints(s + b"y", n, n_ord)]
        es = pyref.stream_ints(s + b"e", n, n_ord)
        e128 = pyref.stream_ints(s + b"e128", n, 1 << 128)
        v = pyref.stream_ints(s + b"v", 1, 1 << 128)[0]
        base = c.mul(pyref.stream_ints(s + b"b", 1, n_ord)[0], g)
        perm = sorted(range(n), key=lambda i: pyref.stream_ints(s + b"perm", n, 1 << 64)[i])
        if n >= 6:                       # exceptional cases: exponents 0 / 1 / n-1, infinity, P + P, P + (-P)
            es[0], es[1], es[2] = 0, 1, n_ord - 1
            xs[3] = None
            ys[4] = xs[4]
            ys[5] = c.neg(xs[5])
        cases.append({"op": "exp_array", "n": n, "x": pts(xs), "e": list(map(hx, es)), "out": pts(c.exp_array(xs, es))})
        cases.append({"op": "exp_scalar", "n": n, "x": pts(xs), "e": hx(v), "out": pts([c.mul(v, Q) for Q in xs])})
        cases.append({"op": "exp_fixed", "n": n, "base": pt(base), "e": list(map(hx, es)), "out": pts(c.exp_fixed(base, es))})
        cases.append({"op": "exp_fixed", "n": n, "base": pt(g), "e": list(map(hx, es)), "out": pts(c.exp_fixed(g, es))})   # OpenSSL-checkable
        cases.append({"op": "mul", "n": n, "x": pts(xs), "y": pts(ys), "out": pts(c.mul_arrays(xs, ys))})
        cases.append({"op": "prod", "n": n, "x": pts(xs), "out": pt(c.prod(xs))})
        cases.append({"op": "exp_prod", "n": n, "ebits": 128, "x": pts(xs), "e": list(map(hx, e128)), "out": pt(c.exp_prod(xs, e128))})
        cases.append({"op": "exp_prod_ring", "n": n, "x": pts(xs), "e": list(map(hx, es)), "out": pt(c.exp_prod(xs, es))})
        cases.append({"op": "permute", "n": n, "x": pts(xs), "perm": perm, "out": pts(pyref.permute(xs, perm))})
        cases.append({"op": "inv", "n": n, "x": pts(xs), "out": pts([c.neg(Q) if Q is not None else None for Q in xs])})
    return {"curve": name, "p": hx(c.p), "n": hx(c.n), "b": hx(c.b), "g": pt(c.g), "cases": cases}


def enc_msg(msg, el):
    out = {}
    for k, val in msg.items():
        if k.startswith("k_"):
            out[k] = list(map(hx, val)) if isinstance(val, list) else hx(val)
        elif isinstance(val, list):
            out[k] = [el(x) for x in val]
        else:
            out[k] = el(val)
    return out


def proof_records():
    NV, NE, NR = 100, 100, 50
    n = 8
    recs = []
    # ---- ModPGroup, 512-bit test group (the size of the reference's own unit test, TestPoSCBasicTW.java:69-140)
    p = pyref.find_safe_prime(512, b"vmn-test-group-512")
    q, g = (p - 1) // 2, 4
    for width in (1, 2):
        t = Tape(b"golden-modp-w%d" % width, q)
        h = [pow(g, x, p) for x in t.ring_array(n)]
        y = pow(g, t.ring_element(), p)
        pkey = [g] * width + [y] * width
        w = [[pow(g, x, p) for x in t.ring_array(n)] for _ in range(2 * width)]
        pi = t.permutation(n)
        s = [t.ring_array(n) for _ in range(width)]
        r = t.ring_array(n)
        e = t.int_array(n, NE)
        v = t.int_array(1, NV)[0]
        wp = P.reencrypt(w, P.reenc_factors(pkey, s, p), pi, p)
        base = {"group": "modp512", "width": width, "n": n, "nbits": [NV, NE, NR], "g": hx(g), "h": list(map(hx, h)), "pkey": list(map(hx, pkey)),
                "w": [list(map(hx, c)) for c in w], "wp": [list(map(hx, c)) for c in wp], "pi": pi, "s": [list(map(hx, c)) for c in s],
                "e": list(map(hx, e)), "v": hx(v)}
        o = P.PoS(p, q, NV, NE, NR, rand=Tape(b"golden-prover", q))
        o.precompute(g, h, pi)
        o.setInstance(pkey, w, wp, s)
        o.setBatchVector(e)
        com, rep = o.commit(), o.reply(v)
        ov = P.PoS(p, q, NV, NE, NR)
        ov.precompute(g, h)
        ov.u = o.u
        ov.setInstance(pkey, w, wp)
        ov.setBatchVector(e)
        ov.computeAF()
        ov.setCommitment(com)
        assert ov.verify(rep, v)
        recs.append(dict(base, proof="PoS", tape="golden-prover", u=list(map(hx, o.u)), commitment=enc_msg(com, hx), reply=enc_msg(rep, hx), verdict=True))
        if width == 1:
            u = P.permutation_commitment(g, h, r, pi, p)
            oc = P.PoSC(p, q, NV, NE, NR, rand=Tape(b"golden-posc", q))
            oc.setInstance(g, h, u, r, pi)
            oc.setBatchVector(e)
            com, rep = oc.commit(), oc.reply(v)
            vc = P.PoSC(p, q, NV, NE, NR)
            vc.setInstance(g, h, u)
            vc.setBatchVector(e)
            vc.setCommitment(com)
            assert vc.verify(rep, v)
            recs.append(dict(base, proof="PoSC", tape="golden-posc", r=list(map(hx, r)), u=list(map(hx, u)), commitment=enc_msg(com, hx),
                             reply=enc_msg(rep, hx), verdict=True))
        u = P.permutation_commitment(g, h, r, pi, p)
        cc = P.CCPoS(p, q, NV, NE, NR, rand=Tape(b"golden-ccpos", q))
        cc.setInstance(g, h, u, pkey, w, wp, r, pi, s)
        cc.setBatchVector(e)
        com, rep = cc.commit(), cc.reply(v)
        cv = P.CCPoS(p, q, NV, NE, NR)
        cv.setInstance(g, h, u, pkey, w, wp)
        cv.setBatchVector(e)
        cv.setCommitment(com)
        cv.computeAB()
        assert cv.verify(rep, v)
        recs.append(dict(base, proof="CCPoS", tape="golden-ccpos", r=list(map(hx, r)), u=list(map(hx, u)), commitment=enc_msg(com, hx),
                         reply=enc_msg(rep, hx), verdict=True))
    # ---- ECqPGroup P-256 (the reference's default group), width 1
    c = Curve("P-256")
    K = P.ECAdapter(c)
    q, g = c.n, c.g
    t = Tape(b"golden-p256", q)
    h = [c.mul(x, g) for x in t.ring_array(n)]
    y = c.mul(t.ring_element(), g)
    pkey = [g, y]
    w = [[c.mul(x, g) for x in t.ring_array(n)] for _ in range(2)]
    pi = t.permutation(n)
    s = [t.ring_array(n)]
    r = t.ring_array(n)
    e = t.int_array(n, NE)
    v = t.int_array(1, NV)[0]
    wp = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi)
    base = {"group": "P-256", "width": 1, "n": n, "nbits": [NV, NE, NR], "g": pt(g), "h": pts(h), "pkey": pts(pkey), "w": [pts(col) for col in w],
            "wp": [pts(col) for col in wp], "pi": pi, "s": [list(map(hx, col)) for col in s], "e": list(map(hx, e)), "v": hx(v)}
    o = P.GPoS(K, NV, NE, NR, rand=Tape(b"golden-prover", q))
    o.precompute(g, h, pi)
    o.setInstance(pkey, w, wp, s)
    o.setBatchVector(e)
    com, rep = o.commit(), o.reply(v)
    ov = P.GPoS(K, NV, NE, NR)
    ov.precompute(g, h)
    ov.u = o.u
    ov.setInstance(pkey, w, wp)
    ov.setBatchVector(e)
    ov.computeAF()
    ov.setCommitment(com)
    assert ov.verify(rep, v)
    recs.append(dict(base, proof="PoS", tape="golden-prover", u=pts(o.u), commitment=enc_msg(com, pt), reply=enc_msg(rep, pt), verdict=True))
    u = P.g_permutation_commitment(K, g, h, r, pi)
    cc = P.GCCPoS(K, NV, NE, NR, rand=Tape(b"golden-ccpos", q))
    cc.setInstance(g, h, u, pkey, w, wp, r, pi, s)
    cc.setBatchVector(e)
    com, rep = cc.commit(), cc.reply(v)
    cv = P.GCCPoS(K, NV, NE, NR)
    cv.setInstance(g, h, u, pkey, w, wp)
    cv.setBatchVector(e)
    cv.setCommitment(com)
    cv.computeAB()
    assert cv.verify(rep, v)
    recs.append(dict(base, proof="CCPoS", tape="golden-ccpos", r=list(map(hx, r)), u=pts(u), commitment=enc_msg(com, pt), reply=enc_msg(rep, pt),
                     verdict=True))
    return {"modp512": {"p": hx(p), "q": hx((p - 1) // 2), "g": "4"}, "records": recs}


def config_records():
    """Transcripts on the BASELINE.json configurations' own groups (tests/golden/proofs_configs.json):
      configs[2]  3072-bit ModPGroup (RFC 3526 group 15), width 1: permutation commitment precomputed for N_max = 8,
                  PoSC on it, shrink to N = 6 (keep list, PermutationCommitment.java:390-471), re-encryption, CCPoS with the
                  plain and the raised verifier (CCPoSBasicW.java:493-506, 571-580);
      configs[4]  ECqPGroup P-256, width 3: PoS and CCPoS.
    Group-generic restatement (oracle/pyref_proofs.py GPoS / GCCPoS, PoSC over integers)."""
    NV, NE, NR = 256, 256, 100
    recs = []
    # ---- configs[2]
    p, q, g = pyref.modp_group(3072)
    K = P.ModPAdapter(p, q)
    n_max, n = 8, 6
    t = Tape(b"golden-cfg2", q)
    h = K.exp_fixed(g, t.ring_array(n_max))
    y = K.exp(g, t.ring_element())
    pkey = [g, y]
    w = [K.exp_fixed(g, t.ring_array(n)) for _ in range(2)]
    pi, r, rho = t.permutation(n_max), t.ring_array(n_max), t.int_array(1, 50)[0]
    e_max, v1 = t.int_array(n_max, NE), t.int_array(1, NV)[0]
    u = P.permutation_commitment(g, h, r, pi, p)
    oc = P.PoSC(p, q, NV, NE, NR, rand=Tape(b"golden-cfg2-posc", q))
    oc.setInstance(g, h, u, r, pi)
    oc.setBatchVector(e_max)
    com1, rep1 = oc.commit(), oc.reply(v1)
    vc = P.PoSC(p, q, NV, NE, NR)
    vc.setInstance(g, h, u)
    vc.setBatchVector(e_max)
    vc.setCommitment(com1)
    assert vc.verify(rep1, v1)
    keep, pi_s = P.shrink_permutation(pi, n)
    u_s = P.extract(u, keep)
    assert u_s == P.permutation_commitment(g, h[:n], r[:n], pi_s, p)
    s = [t.ring_array(n)]
    e, v2 = t.int_array(n, NE), t.int_array(1, NV)[0]
    wp = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi_s)
    cc = P.GCCPoS(K, NV, NE, NR, rand=Tape(b"golden-cfg2-ccpos", q))
    cc.setInstance(g, h[:n], u_s, pkey, w, wp, r[:n], pi_s, s)
    cc.setBatchVector(e)
    com2, rep2 = cc.commit(), cc.reply(v2)
    for raised in (False, True):
        cv = P.GCCPoS(K, NV, NE, NR)
        cv.setInstance(g, h[:n], u_s, pkey, w, wp)
        cv.setBatchVector(e)
        cv.setCommitment(com2)
        if raised:
            cv.computeAB(K.exp_scalar(u_s, rho))
            assert cv.verify(rep2, v2, K.exp_scalar(h[:n], rho), rho)
        else:
            cv.computeAB()
            assert cv.verify(rep2, v2)
    L = lambda xs: list(map(hx, xs))
    recs.append({"config": 2, "group": "modp3072", "width": 1, "nbits": [NV, NE, NR], "n_max": n_max, "n": n, "g": hx(g), "h": L(h),
                 "pkey": L(pkey), "w": [L(c) for c in w], "wp": [L(c) for c in wp], "pi": pi, "r": L(r), "rho": hx(rho), "u": L(u),
                 "e_max": L(e_max), "v_posc": hx(v1), "tape_posc": "golden-cfg2-posc", "posc_commitment": enc_msg(com1, hx),
                 "posc_reply": enc_msg(rep1, hx), "keep": [int(k) for k in keep], "pi_shrunk": pi_s, "u_shrunk": L(u_s),
                 "s": [L(c) for c in s], "e": L(e), "v": hx(v2), "tape_ccpos": "golden-cfg2-ccpos", "ccpos_commitment": enc_msg(com2, hx),
                 "ccpos_reply": enc_msg(rep2, hx), "verdict": True})
    # ---- configs[4]
    c = Curve("P-256")
    K = P.ECAdapter(c)
    q, g = c.n, c.g
    n, width = 8, 3
    t = Tape(b"golden-cfg4", q)
    h = K.exp_fixed(g, t.ring_array(n))
    y = K.exp(g, t.ring_element())
    pkey = [g] * width + [y] * width
    w = [K.exp_fixed(g, t.ring_array(n)) for _ in range(2 * width)]
    pi, r = t.permutation(n), t.ring_array(n)
    s = [t.ring_array(n) for _ in range(width)]
    e, v = t.int_array(n, NE), t.int_array(1, NV)[0]
    wp = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi)
    o = P.GPoS(K, NV, NE, NR, rand=Tape(b"golden-cfg4-pos", q))
    o.precompute(g, h, pi)
    o.setInstance(pkey, w, wp, s)
    o.setBatchVector(e)
    com, rep = o.commit(), o.reply(v)
    ov = P.GPoS(K, NV, NE, NR)
    ov.precompute(g, h)
    ov.u = o.u
    ov.setInstance(pkey, w, wp)
    ov.setBatchVector(e)
    ov.computeAF()
    ov.setCommitment(com)
    assert ov.verify(rep, v)
    u = P.g_permutation_commitment(K, g, h, r, pi)
    cc = P.GCCPoS(K, NV, NE, NR, rand=Tape(b"golden-cfg4-ccpos", q))
    cc.setInstance(g, h, u, pkey, w, wp, r, pi, s)
    cc.setBatchVector(e)
    com2, rep2 = cc.commit(), cc.reply(v)
    cv = P.GCCPoS(K, NV, NE, NR)
    cv.setInstance(g, h, u, pkey, w, wp)
    cv.setBatchVector(e)
    cv.setCommitment(com2)
    cv.computeAB()
    assert cv.verify(rep2, v)
    recs.append({"config": 4, "group": "P-256", "width": width, "nbits": [NV, NE, NR], "n": n, "g": pt(g), "h": pts(h), "pkey": pts(pkey),
                 "w": [pts(col) for col in w], "wp": [pts(col) for col in wp], "pi": pi, "s": [L(col) for col in s], "r": L(r),
                 "e": L(e), "v": hx(v), "tape_pos": "golden-cfg4-pos", "pos_u": pts(o.u), "pos_commitment": enc_msg(com, pt),
                 "pos_reply": enc_msg(rep, pt), "u": pts(u), "tape_ccpos": "golden-cfg4-ccpos", "ccpos_commitment": enc_msg(com2, pt),
                 "ccpos_reply": enc_msg(rep2, pt), "verdict": True})
    return {"records": recs}


def main():
    rec = config_records()
    path = os.path.join(HERE, "proofs_configs.json")
    with open(path, "w") as f:
        json.dump(rec, f, separators=(",", ":"))
    print(path, len(rec["records"]), "records", os.path.getsize(path), "bytes")
    for name, fname in (("P-224", "ec_p224.json"), ("P-256", "ec_p256.json"), ("P-384", "ec_p384.json"), ("P-521", "ec_p521.json")):
        rec = ec_cases(name, [1, 2, 9, 64] if name == "P-256" else [1, 7, 33] if name == "P-384" else [1, 7, 20])
        path = os.path.join(HERE, fname)
        with open(path, "w") as f:
            json.dump(rec, f, separators=(",", ":"))
        print(path, len(rec["cases"]), "cases", os.path.getsize(path), "bytes")
    rec = proof_records()
    path = os.path.join(HERE, "proofs_n8.json")
    with open(path, "w") as f:
        json.dump(rec, f, separators=(",", ":"))
    print(path, len(rec["records"]), "transcripts", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
