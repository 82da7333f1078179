"""CPU suite: the host logic of the Python mirror drivers (hvzk.py, mixnet.py) on the integer-backed stand-in for the
device arrays (tests/fake_backend.py), through the same flows the GPU suite runs on the real kernels
(tests/test_gpu_configs.py): precompute(N_max) -> shrink(N) -> CCPoS plain and raised, widths 3 and 4, and the
verifier's use of every bit of a received k_E.  No compute call reaches the HIP library here."""
import importlib.util
import os
import sys

import pytest

from conftest import load_golden
from fake_backend import FakeGroup
from oracle import pyref_proofs as P
from proof_cases import check_ccpos, check_pos, make_instance
from tape import Tape


@pytest.fixture(scope="module")
def mods(entry):
    import mirror
    out = mirror.load(entry, ("hvzk", "mixnet"))
    return out


def group(bits=512):
    grp, _ = load_golden(bits)
    p, q, g = grp["p"], grp["q"], grp["g"]
    return FakeGroup(p, q, g), P.ModPAdapter(p, q), p, q, g


@pytest.mark.parametrize("width", [1, 3, 4])
def test_pos_and_ccpos_flows_widths(width, mods):
    G, K, p, q, g = group()
    bits3 = (100, 100, 50)
    n = 9
    h, pkey, w, t = make_instance(K, g, n, width, b"cpu-width%d" % width)
    H, W, WP, wp_o, s, S, pi = check_pos("python", mods, G, K, g, h, pkey, w, t, bits3)
    r, rho = t.ring_array(n), t.int_array(1, 50)[0]
    u_o = P.g_permutation_commitment(K, g, h, r, pi)
    pc = mods["mixnet"].PermutationCommitment(G, H)
    U = pc.precompute(r, pi)
    assert U.toInts() == u_o
    check_ccpos("python", mods, G, K, g, h, H, u_o, U, pkey, w, W, wp_o, WP, r, pc.exponents, pi, s, S, t, bits3, rho=rho)


def test_precompute_shrink_then_ccpos(mods):
    """PermutationCommitment.shrink (mixnet/PermutationCommitment.java:390-471) on both sides of the keep list."""
    G, K, p, q, g = group()
    mx = mods["mixnet"]
    bits3 = (100, 100, 50)
    n_max, n = 14, 9
    h, pkey, w_max, t = make_instance(K, g, n_max, 1, b"cpu-shrink")
    H = G.toElementArray(h)
    pi, r, rho = t.permutation(n_max), t.ring_array(n_max), t.int_array(1, 50)[0]
    u_o = P.permutation_commitment(g, h, r, pi, p)
    prover = mx.PermutationCommitment(G, H)
    assert prover.precompute(r, pi).toInts() == u_o
    prover.raise_(rho)
    verifier = mx.PermutationCommitment(G, H)                   # another party: holds u and u^rho only
    verifier.commitment, verifier.raisedCommitment = G.toElementArray(u_o), G.toElementArray(K.exp_scalar(u_o, rho))
    keep_o, pi_s_o = P.shrink_permutation(pi, n)
    keep = prover.shrink(n)
    assert keep == keep_o and list(prover.permutation) == pi_s_o and prover.exponents.toInts() == r[:n]
    assert verifier.shrink(n, keep) == keep
    u_s = P.extract(u_o, keep_o)
    assert prover.commitment.toInts() == verifier.commitment.toInts() == u_s
    assert u_s == P.permutation_commitment(g, h[:n], r[:n], pi_s_o, p)
    assert verifier.raisedCommitment.toInts() == K.exp_scalar(u_s, rho)
    # a keep list with the wrong number of flags is replaced by the trivial one (:437-445)
    other = mx.PermutationCommitment(G, H)
    other.commitment = G.toElementArray(u_o)
    wrong = list(keep_o)
    wrong[wrong.index(True)] = False
    assert other.shrink(n, wrong) == [i < n for i in range(n_max)] == P.sanitize_keep_list(wrong, n_max, n)
    # committed shuffle of the n ciphertexts
    w = [c[:n] for c in w_max]
    s = [t.ring_array(n)]
    W, S = [G.toElementArray(c) for c in w], [G.ringArray(c) for c in s]
    wp_o = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi_s_o)
    WP = mx.reencrypt(W, mx.reencFactors(G, pkey, S), pi_s_o)
    assert [c.toInts() for c in WP] == wp_o
    check_ccpos("python", mods, G, K, g, h[:n], H.copyOfRange(0, n), u_s, prover.commitment, pkey, w, W, wp_o, WP, r[:n],
                prover.exponents, pi_s_o, s, S, t, bits3, rho=rho)


def test_raised_form_of_the_generic_oracle_equals_the_integer_one():
    """GCCPoS.verify's raised branch (added for the curve tests) against the integer CCPoS it restates."""
    G, K, p, q, g = group()
    n = 7
    h, pkey, w, t = make_instance(K, g, n, 2, b"cpu-raised")
    pi, r, s = t.permutation(n), t.ring_array(n), [t.ring_array(n), t.ring_array(n)]
    e, v, rho = t.int_array(n, 100), t.int_array(1, 100)[0], t.int_array(1, 50)[0]
    u = P.permutation_commitment(g, h, r, pi, p)
    wp = P.reencrypt(w, P.reenc_factors(pkey, s, p), pi, p)
    a = P.CCPoS(p, q, 100, 100, 50, rand=Tape(b"x", q))
    b = P.GCCPoS(K, 100, 100, 50, rand=Tape(b"x", q))
    for o in (a, b):
        o.setInstance(g, h, u, pkey, w, wp, r, pi, s)
        o.setBatchVector(e)
    ca, cb = a.commit(), b.commit()
    ra, rb = a.reply(v), b.reply(v)
    assert ca == cb and ra == rb
    ru, rh = K.exp_scalar(u, rho), K.exp_scalar(h, rho)
    for o in (a, b):
        o.setCommitment(ca)
        o.computeAB(ru)
        assert o.verify(ra, v, rh, rho)
        bad = dict(ra)
        bad["k_B"] = [ra["k_B"][0], (ra["k_B"][1] + 1) % q]
        assert not o.verify(bad, v, rh, rho)
