"""Golden transcripts on the BASELINE configurations' own groups (tests/golden/proofs_configs.json, made by
tests/golden/gen_golden_proofs.py: configs[2] = 3072-bit ModPGroup with precompute -> shrink -> CCPoS plain + raised,
configs[4] = P-256 width 3).

CPU part: an independent check of the fixture -- the multi-exponentiations and fixed-base powers through the C + GMP
oracle, the verifiers of oracle/pyref_proofs.py accept, a tampered reply is rejected.  GPU part: the HIP kernels and
both proof-driver implementations reproduce every message bit for bit."""
import json
import os

import pytest

from conftest import ROOT
from oracle import pyref, pyref_proofs as P
from oracle.pyref_ec import Curve
from tape import Tape


def load():
    with open(os.path.join(ROOT, "tests", "golden", "proofs_configs.json")) as f:
        return {r["config"]: r for r in json.load(f)["records"]}


def ints(vs):
    return [int(v, 16) for v in vs]


def pt(v):
    return None if v is None else (int(v[0], 16), int(v[1], 16))


def pts(vs):
    return [pt(v) for v in vs]


def dec_msg(m, arr, el):
    out = {}
    for k, x in m.items():
        if k.startswith("k_"):
            out[k] = ints(x) if isinstance(x, list) else int(x, 16)
        elif isinstance(x, list) and (arr is ints or not x or x[0] is None or isinstance(x[0], list)):
            out[k] = arr(x)
        else:
            out[k] = el(x)
    return out


# ---------------------------------------------------------------------------------------------- CPU
def test_config2_record_against_gmp_and_the_oracle_verifiers(oracle_for):
    r = load()[2]
    p, q, g = pyref.modp_group(3072)
    assert int(r["g"], 16) == g and p.bit_length() == 3072
    orc = oracle_for(p, q)
    NV, NE, NR = r["nbits"]
    n_max, n = r["n_max"], r["n"]
    h, pkey, pi, rr, rho = ints(r["h"]), ints(r["pkey"]), r["pi"], ints(r["r"]), int(r["rho"], 16)
    u, keep, pi_s, u_s = ints(r["u"]), [bool(k) for k in r["keep"]], r["pi_shrunk"], ints(r["u_shrunk"])
    # the commitment and its shrinking, through GMP
    assert u == pyref.permute(orc.mul(h, orc.exp_fixed(g, rr)), pi)
    assert (keep, pi_s) == P.shrink_permutation(pi, n) and u_s == P.extract(u, keep)
    assert u_s == pyref.permute(orc.mul(h[:n], orc.exp_fixed(g, rr[:n])), pi_s)
    w, wp, s = [ints(c) for c in r["w"]], [ints(c) for c in r["wp"]], [ints(c) for c in r["s"]]
    assert wp == [pyref.permute(orc.mul(w[c], orc.exp_fixed(pkey[c], s[0])), P.inv_perm(pi_s)) for c in range(2)]
    # PoSC on the full commitment
    com1, rep1 = dec_msg(r["posc_commitment"], ints, lambda x: int(x, 16)), dec_msg(r["posc_reply"], ints, lambda x: int(x, 16))
    vc = P.PoSC(p, q, NV, NE, NR)
    vc.setInstance(g, h, u)
    vc.setBatchVector(ints(r["e_max"]))
    vc.setCommitment(com1)
    assert vc.verify(rep1, int(r["v_posc"], 16))
    bad = dict(rep1)
    bad["k_A"] = (rep1["k_A"] + 1) % q
    assert not vc.verify(bad, int(r["v_posc"], 16))
    # CCPoS on the shrunk instance: A' through GMP's Pippenger, both verifier forms
    com2, rep2 = dec_msg(r["ccpos_commitment"], ints, lambda x: int(x, 16)), dec_msg(r["ccpos_reply"], ints, lambda x: int(x, 16))
    e, v = ints(r["e"]), int(r["v"], 16)
    K = P.ModPAdapter(p, q)
    for raised in (False, True):
        cv = P.GCCPoS(K, NV, NE, NR)
        cv.setInstance(g, h[:n], u_s, pkey, w, wp)
        cv.setBatchVector(e)
        cv.setCommitment(com2)
        if raised:
            cv.computeAB(orc.exp_scalar(u_s, rho))
            assert cv.verify(rep2, v, orc.exp_scalar(h[:n], rho), rho)
        else:
            cv.computeAB()
            assert cv.A == orc.exp_prod(u_s, e, 256, pippenger_c=4)
            assert cv.verify(rep2, v) is r["verdict"]
    # and the prover is reproducible from its tape
    cc = P.GCCPoS(K, NV, NE, NR, rand=Tape(r["tape_ccpos"].encode(), q))
    cc.setInstance(g, h[:n], u_s, pkey, w, wp, rr[:n], pi_s, s)
    cc.setBatchVector(e)
    assert (cc.commit(), cc.reply(v)) == (com2, rep2)


def test_config4_record_is_accepted_by_the_oracle_verifiers():
    r = load()[4]
    c = Curve("P-256")
    K = P.ECAdapter(c)
    NV, NE, NR = r["nbits"]
    assert r["width"] == 3 and pt(r["g"]) == c.g
    h, pkey = pts(r["h"]), pts(r["pkey"])
    w, wp = [pts(col) for col in r["w"]], [pts(col) for col in r["wp"]]
    assert all(Q is None or c.on_curve(Q) for col in wp for Q in col)
    e, v = ints(r["e"]), int(r["v"], 16)
    com, rep = dec_msg(r["pos_commitment"], pts, pt), dec_msg(r["pos_reply"], pts, pt)
    ov = P.GPoS(K, NV, NE, NR)
    ov.precompute(c.g, h)
    ov.u = pts(r["pos_u"])
    ov.setInstance(pkey, w, wp)
    ov.setBatchVector(e)
    ov.computeAF()
    ov.setCommitment(com)
    assert ov.verify(rep, v) is r["verdict"]
    bad = dict(rep)
    bad["k_F"] = [rep["k_F"][0], rep["k_F"][1], (rep["k_F"][2] + 1) % c.n]
    assert not ov.verify(bad, v) and ov.verdicts == (True, True, True, True, False)
    com2, rep2 = dec_msg(r["ccpos_commitment"], pts, pt), dec_msg(r["ccpos_reply"], pts, pt)
    cv = P.GCCPoS(K, NV, NE, NR)
    cv.setInstance(c.g, h, pts(r["u"]), pkey, w, wp)
    cv.setBatchVector(e)
    cv.setCommitment(com2)
    cv.computeAB()
    assert cv.verify(rep2, v)


# ---------------------------------------------------------------------------------------------- GPU
def _same(msg, want):
    assert set(msg) == set(want)
    for k, exp in want.items():
        got = msg[k].toInts() if hasattr(msg[k], "toInts") else msg[k]
        assert got == exp, k


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["native"])
def test_drivers_reproduce_the_config2_record(impl, vmn, gpu_ctx, entry):
    from proof_cases import load_driver_modules
    mods = load_driver_modules(entry)
    hv, mx, nat = mods["hvzk" if impl == "python" else "native"], mods["mixnet"], mods["native"]
    r = load()[2]
    p, q, g = pyref.modp_group(3072)
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    NV, NE, NR = r["nbits"]
    n = r["n"]
    el = lambda x: int(x, 16)
    H = G.toElementArray(ints(r["h"]))
    pi, rho, pkey = r["pi"], int(r["rho"], 16), ints(r["pkey"])
    pc = mx.PermutationCommitment(G, H)
    U = pc.precompute(ints(r["r"]), pi)
    assert U.toInts() == ints(r["u"])
    pc.raise_(rho)
    pr = hv.PoSCBasicTW(G, NV, NE, NR, rand=Tape(r["tape_posc"].encode(), q))
    pr.setInstance(g, H, U, pc.exponents, pi)
    pr.setBatchVector(ints(r["e_max"]))
    com1, rep1 = pr.commit(), pr.reply(int(r["v_posc"], 16))
    _same(com1, dec_msg(r["posc_commitment"], ints, el))
    _same(rep1, dec_msg(r["posc_reply"], ints, el))
    keep = pc.shrink(n)
    assert [int(k) for k in keep] == r["keep"] and list(pc.permutation) == r["pi_shrunk"]
    assert pc.commitment.toInts() == ints(r["u_shrunk"])
    if impl == "native":
        assert nat.permutation_shrink_native(pi, n) == ([bool(k) for k in r["keep"]], r["pi_shrunk"])
    H_s = H.copyOfRange(0, n)
    W = [G.toElementArray(ints(c)) for c in r["w"]]
    S = [G.ringArray(ints(c)) for c in r["s"]]
    WP = nat.reencrypt_native(G, pkey, W, S, pc.permutation) if impl == "native" else \
        mx.reencrypt(W, mx.reencFactors(G, pkey, S), pc.permutation)
    assert [c.toInts() for c in WP] == [ints(c) for c in r["wp"]]
    cp = hv.CCPoSBasicW(G, NV, NE, NR, rand=Tape(r["tape_ccpos"].encode(), q))
    cp.setInstance(g, H_s, pc.commitment, pkey, W, WP, pc.exponents, pc.permutation, S)
    cp.setBatchVector(ints(r["e"]))
    v = int(r["v"], 16)
    com2, rep2 = cp.commit(), cp.reply(v)
    _same(com2, dec_msg(r["ccpos_commitment"], ints, el))
    _same(rep2, dec_msg(r["ccpos_reply"], ints, el))
    for raised in (False, True):
        cv = hv.CCPoSBasicW(G, NV, NE, NR)
        cv.setInstance(g, H_s, pc.commitment, pkey, W, WP)
        cv.setBatchVector(ints(r["e"]))
        cv.setCommitment(com2)
        cv.setChallenge(v)
        if raised:
            cv.computeAB(pc.raisedCommitment)
            assert cv.verify(rep2, H_s.exp(rho), rho) is r["verdict"]
        else:
            cv.computeAB()
            assert cv.verify(rep2) is r["verdict"]


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["native"])
def test_drivers_reproduce_the_config4_record(impl, vmn, gpu_ctx, entry):
    from proof_cases import load_driver_modules
    mods = load_driver_modules(entry)
    hv, nat = mods["hvzk" if impl == "python" else "native"], mods["native"]
    r = load()[4]
    c = Curve("P-256")
    G = vmn.ECqPGroup(gpu_ctx, "P-256")
    NV, NE, NR = r["nbits"]
    H, pkey, pi = G.toElementArray(pts(r["h"])), pts(r["pkey"]), r["pi"]
    W, WP = [G.toElementArray(pts(col)) for col in r["w"]], [G.toElementArray(pts(col)) for col in r["wp"]]
    S = [G.ringArray(ints(col)) for col in r["s"]]
    e, v = ints(r["e"]), int(r["v"], 16)
    assert [col.toInts() for col in nat.reencrypt_native(G, pkey, W, S, pi)] == [pts(col) for col in r["wp"]]
    pr = hv.PoSBasicTW(G, NV, NE, NR, rand=Tape(r["tape_pos"].encode(), c.n))
    pr.precompute(c.g, H, pi)
    assert pr.u.toInts() == pts(r["pos_u"])
    pr.setInstance(pkey, W, WP, S)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    _same(com, dec_msg(r["pos_commitment"], pts, pt))
    _same(rep, dec_msg(r["pos_reply"], pts, pt))
    ver = hv.PoSBasicTW(G, NV, NE, NR)
    ver.precompute(c.g, H)
    ver.setPermutationCommitment(pr.u)
    ver.setInstance(pkey, W, WP)
    ver.setBatchVector(e)
    ver.computeAF()
    ver.setCommitment(com)
    ver.setChallenge(v)
    assert ver.verify(rep) is r["verdict"]
    U, R = G.toElementArray(pts(r["u"])), G.ringArray(ints(r["r"]))
    cp = hv.CCPoSBasicW(G, NV, NE, NR, rand=Tape(r["tape_ccpos"].encode(), c.n))
    cp.setInstance(c.g, H, U, pkey, W, WP, R, pi, S)
    cp.setBatchVector(e)
    com2, rep2 = cp.commit(), cp.reply(v)
    _same(com2, dec_msg(r["ccpos_commitment"], pts, pt))
    _same(rep2, dec_msg(r["ccpos_reply"], pts, pt))
    cv = hv.CCPoSBasicW(G, NV, NE, NR)
    cv.setInstance(c.g, H, U, pkey, W, WP)
    cv.setBatchVector(e)
    cv.setCommitment(com2)
    cv.setChallenge(v)
    cv.computeAB()
    assert cv.verify(rep2) is r["verdict"]
