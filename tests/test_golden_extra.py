"""Golden vectors for the curve groups and whole proof transcripts (tests/golden/ec_p256.json, ec_p384.json,
proofs_n8.json, made by tests/golden/gen_golden_proofs.py).

CPU part: the fixtures against independent CPU code paths (OpenSSL for curve points, the Jacobian model, the C+GMP
oracle for the modular transcripts' building blocks, the oracle verifier).  GPU part: the HIP kernels and both
proof-driver implementations must reproduce the fixtures bit for bit."""
import importlib.util
import json
import os
import sys

import pytest

from conftest import ROOT
from oracle import pyref, pyref_proofs as P
from oracle.pyref_ec import Curve, jac_add, jac_to_affine
from tape import Tape


def load(name):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


def pt(v):
    return None if v is None else (int(v[0], 16), int(v[1], 16))


def pts(vs):
    return [pt(v) for v in vs]


def ints(vs):
    return [int(v, 16) for v in vs]


# ---------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("fname", ["ec_p224.json", "ec_p256.json", "ec_p384.json", "ec_p521.json"])
def test_curve_vectors_against_openssl_and_the_jacobian_model(fname):
    from test_oracle_ec import openssl_mul
    rec = load(fname)
    c = Curve(rec["curve"])
    assert (c.p, c.n, c.b, c.g) == (int(rec["p"], 16), int(rec["n"], 16), int(rec["b"], 16), pt(rec["g"]))
    J = lambda Q: (1, 1, 0, True) if Q is None else (Q[0], Q[1], 1, False)
    checked = 0
    for case in rec["cases"]:
        if case["op"] == "exp_fixed" and pt(case["base"]) == c.g:
            for e, Q in zip(ints(case["e"]), pts(case["out"])):      # multiples of the generator: OpenSSL computes them too
                if e % c.n:
                    assert Q == openssl_mul(rec["curve"], e)
                    checked += 1
                else:
                    assert Q is None
        if case["op"] == "mul":
            for X, Y, Z in zip(pts(case["x"]), pts(case["y"]), pts(case["out"])):
                assert jac_to_affine(c, jac_add(c, J(X), J(Y))) == Z          # incl. P + P, P + (-P), infinity
                assert Z is None or c.on_curve(Z)
        if case["op"] in ("exp_array", "exp_scalar", "permute", "inv"):
            assert all(Q is None or c.on_curve(Q) for Q in pts(case["out"]))
    assert checked >= 25


def test_proof_transcripts_are_reproducible_and_accepted_by_the_oracle_verifier():
    rec = load("proofs_n8.json")
    p, q, g = (int(rec["modp512"][k], 16) for k in ("p", "q", "g"))
    assert pyref.is_probable_prime(p) and p == 2 * q + 1
    seen = set()
    for r in rec["records"]:
        seen.add((r["group"], r["proof"], r["width"]))
        NV, NE, NR = r["nbits"]
        if r["group"] != "modp512":
            continue
        h, pkey, pi, e, v = ints(r["h"]), ints(r["pkey"]), r["pi"], ints(r["e"]), int(r["v"], 16)
        w, wp, s = [ints(c) for c in r["w"]], [ints(c) for c in r["wp"]], [ints(c) for c in r["s"]]
        assert wp == P.reencrypt(w, P.reenc_factors(pkey, s, p), pi, p)
        dec = lambda m: {k: (ints(x) if isinstance(x, list) else int(x, 16)) for k, x in m.items()}
        com, rep = dec(r["commitment"]), dec(r["reply"])
        if r["proof"] == "PoS":
            o = P.PoS(p, q, NV, NE, NR, rand=Tape(r["tape"].encode(), q))
            o.precompute(g, h, pi)
            o.setInstance(pkey, w, wp, s)
            o.setBatchVector(e)
            assert (o.commit(), o.reply(v)) == (com, rep) and o.u == ints(r["u"])
            ver = P.PoS(p, q, NV, NE, NR)
            ver.precompute(g, h)
            ver.u = ints(r["u"])
            ver.setInstance(pkey, w, wp)
            ver.setBatchVector(e)
            ver.computeAF()
            ver.setCommitment(com)
            assert ver.verify(rep, v) is r["verdict"]
            rep["k_C"] = (rep["k_C"] + 1) % q
            assert not ver.verify(rep, v)
        elif r["proof"] == "PoSC":
            ver = P.PoSC(p, q, NV, NE, NR)
            ver.setInstance(g, h, ints(r["u"]))
            ver.setBatchVector(e)
            ver.setCommitment(com)
            assert ver.verify(rep, v) is r["verdict"]
        else:
            ver = P.CCPoS(p, q, NV, NE, NR)
            ver.setInstance(g, h, ints(r["u"]), pkey, w, wp)
            ver.setBatchVector(e)
            ver.setCommitment(com)
            ver.computeAB()
            assert ver.verify(rep, v) is r["verdict"]
    assert {("modp512", "PoS", 1), ("modp512", "PoS", 2), ("modp512", "PoSC", 1), ("modp512", "CCPoS", 2), ("P-256", "PoS", 1),
            ("P-256", "CCPoS", 1)} <= seen


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("first_level_rows", ["normalised", "as they are"])
@pytest.mark.parametrize("fname", ["ec_p224.json", "ec_p256.json", "ec_p384.json", "ec_p521.json"])
def test_hip_curve_kernels_reproduce_the_golden_vectors(fname, first_level_rows, vmn, gpu_ctx, monkeypatch):
    # both first levels of a multi-exponentiation over a curve (tests/test_gpu_ec.py: first_level_rows)
    monkeypatch.setenv("VMN_EC_NORMALISE_MIN", "0" if first_level_rows == "normalised" else "1000000000")
    rec = load(fname)
    G = vmn.ECqPGroup(gpu_ctx, rec["curve"])
    for case in rec["cases"]:
        op = case["op"]
        X = G.toElementArray(pts(case["x"])) if "x" in case else None
        if op == "exp_array":
            got = X.exp(G.ringArray(ints(case["e"]))).toInts()
        elif op == "exp_scalar":
            got = X.exp(int(case["e"], 16)).toInts()
        elif op == "exp_fixed":
            got = G.exp(pt(case["base"]), G.ringArray(ints(case["e"]))).toInts()
        elif op == "mul":
            got = X.mul(G.toElementArray(pts(case["y"]))).toInts()
        elif op == "prod":
            assert X.prod() == pt(case["out"]), (op, case["n"])
            continue
        elif op == "exp_prod":
            assert X.expProd(ints(case["e"]), case["ebits"]) == pt(case["out"]), (op, case["n"])
            continue
        elif op == "exp_prod_ring":
            assert X.expProd(G.ringArray(ints(case["e"]))) == pt(case["out"]), (op, case["n"])
            continue
        elif op == "permute":
            got = X.permute(case["perm"]).toInts()
        elif op == "inv":
            got = X.inv().toInts()
        assert got == pts(case["out"]), (op, case["n"])


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["python", "native"])
def test_proof_drivers_reproduce_the_golden_transcripts(impl, vmn, gpu_ctx, entry):
    import mirror
    mods = mirror.load(entry, ("hvzk", "mixnet", "native"))
    hv = mods["hvzk" if impl == "python" else "native"]
    rec = load("proofs_n8.json")
    p, q, g = (int(rec["modp512"][k], 16) for k in ("p", "q", "g"))
    groups = {"modp512": (vmn.ModPGroup(gpu_ctx, p, q, g), ints, lambda x: int(x, 16)),
              "P-256": (vmn.ECqPGroup(gpu_ctx, "P-256"), pts, pt)}
    for r in rec["records"]:
        G, arr, el = groups[r["group"]]
        NV, NE, NR = r["nbits"]
        gg, H, pkey, pi = el(r["g"]), G.toElementArray(arr(r["h"])), arr(r["pkey"]), r["pi"]
        W, WP = [G.toElementArray(arr(c)) for c in r["w"]], [G.toElementArray(arr(c)) for c in r["wp"]]
        S = [G.ringArray(ints(c)) for c in r["s"]]
        e, v = ints(r["e"]), int(r["v"], 16)
        tape = Tape(r["tape"].encode(), G.q)

        def same(msg, want):
            for k, val in want.items():
                gotv = msg[k].toInts() if hasattr(msg[k], "toInts") else msg[k]
                if k.startswith("k_"):
                    exp = ints(val) if isinstance(val, list) else int(val, 16)
                else:
                    exp = arr(val) if (isinstance(val, list) and (r["group"] == "modp512" or val and (val[0] is None or isinstance(val[0], list)))) else el(val)
                assert gotv == exp, (r["group"], r["proof"], r["width"], k)

        if r["proof"] == "PoS":
            pr = hv.PoSBasicTW(G, NV, NE, NR, rand=tape)
            pr.precompute(gg, H, pi)
            assert pr.u.toInts() == arr(r["u"])
            pr.setInstance(pkey, W, WP, S)
            pr.setBatchVector(e)
            com, rep = pr.commit(), pr.reply(v)
            same(com, r["commitment"])
            same(rep, r["reply"])
            ver = hv.PoSBasicTW(G, NV, NE, NR)
            ver.precompute(gg, H)
            ver.setPermutationCommitment(pr.u)
            ver.setInstance(pkey, W, WP)
            ver.setBatchVector(e)
            ver.computeAF()
            ver.setCommitment(com)
            ver.setChallenge(v)
            assert ver.verify(rep) is r["verdict"]
        elif r["proof"] == "PoSC":
            U, R = G.toElementArray(arr(r["u"])), G.ringArray(ints(r["r"]))
            pr = hv.PoSCBasicTW(G, NV, NE, NR, rand=tape)
            pr.setInstance(gg, H, U, R, pi)
            pr.setBatchVector(e)
            com, rep = pr.commit(), pr.reply(v)
            same(com, r["commitment"])
            same(rep, r["reply"])
            ver = hv.PoSCBasicTW(G, NV, NE, NR)
            ver.setInstance(gg, H, U)
            ver.setBatchVector(e)
            ver.setCommitment(com)
            ver.setChallenge(v)
            assert ver.verify(rep) is r["verdict"]
        else:
            U, R = G.toElementArray(arr(r["u"])), G.ringArray(ints(r["r"]))
            pr = hv.CCPoSBasicW(G, NV, NE, NR, rand=tape)
            pr.setInstance(gg, H, U, pkey, W, WP, R, pi, S)
            pr.setBatchVector(e)
            com, rep = pr.commit(), pr.reply(v)
            same(com, r["commitment"])
            same(rep, r["reply"])
            ver = hv.CCPoSBasicW(G, NV, NE, NR)
            ver.setInstance(gg, H, U, pkey, W, WP)
            ver.setBatchVector(e)
            ver.setCommitment(com)
            ver.setChallenge(v)
            ver.computeAB()
            assert ver.verify(rep) is r["verdict"]
