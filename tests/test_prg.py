"""N1: PRGHeuristic / RandomOracle over SHA-256, the random vector of a proof and the independent
generators.

CPU: the Python restatement (oracle/pyref_prg.py) and the library's host functions against the known-answer vectors
published for these two constructions (Verificatum verifier specification, test vectors for PRGHeuristic and
RandomOracle with SHA-256).  GPU: the device generators against the Python restatement."""
import ctypes
import os

import pytest

from conftest import ROOT, load_golden
from oracle import pyref_prg

SEED = bytes(range(32))
# PRGHeuristic(SHA-256), seed 000102...1f: the first 128 output bytes
PRG_SHA256 = ("70f4003d52b6eb03da852e93256b5986b5d4883098bb7973bc5318cc66637a84"
              "04a6950a06d3e3308ad7d3606ef810eb124e3943404ca746a12c51c7bf776839"
              "0f8d842ac9cb62349779a7537a78327d545aaeb33b2d42c7d1dc3680a4b23628"
              "627e9db8ad47bfe76dbe653d03d2c0a35999ed28a5023924150d72508668d244")
# RandomOracle(SHA-256, 65 bits) of the same 32 bytes: 9 bytes, 7 leading bits cleared
RO_SHA256_65 = "001a8d6b6f65899ba5"


def test_python_restatement_reproduces_the_published_vectors():
    assert pyref_prg.prg_bytes(SEED, 128).hex() == PRG_SHA256
    assert pyref_prg.random_oracle(SEED, 65).hex() == RO_SHA256_65
    assert pyref_prg.random_oracle(SEED, 261)[0] < 32 and len(pyref_prg.random_oracle(SEED, 261)) == 33


def test_library_host_functions_reproduce_the_published_vectors(entry):
    lib = ctypes.CDLL(os.path.join(ROOT, "verificatum-vmn_amd", "libvmnhip.so"))
    out = ctypes.create_string_buffer(128)
    assert lib.vmn_prg_bytes(SEED, ctypes.c_size_t(32), out, ctypes.c_size_t(128)) == 0
    assert out.raw.hex() == PRG_SHA256
    for nout in (65, 261, 519, 1024):
        nb = (nout + 7) // 8
        out = ctypes.create_string_buffer(nb)
        assert lib.vmn_random_oracle(SEED, ctypes.c_size_t(32), ctypes.c_int(nout), out) == 0
        assert out.raw == pyref_prg.random_oracle(SEED, nout)
    out = ctypes.create_string_buffer(9)
    lib.vmn_random_oracle(SEED, ctypes.c_size_t(32), ctypes.c_int(65), out)
    assert out.raw.hex() == RO_SHA256_65
    big = bytes(range(256)) * 5                       # multi-block input through the host SHA-256
    out = ctypes.create_string_buffer(32)
    assert lib.vmn_random_oracle(big, ctypes.c_size_t(len(big)), ctypes.c_int(256), out) == 0
    assert out.raw == pyref_prg.random_oracle(big, 256)
    assert lib.vmn_prg_bytes(SEED, ctypes.c_size_t(31), out, ctypes.c_size_t(32)) == -5          # VMN_ERR_UNSUPPORTED


def test_sha384_and_sha512_variants_of_prg_and_random_oracle(entry):
    """The reference lets the operator pick SHA-256, SHA-384 or SHA-512 for the PRG and for the random oracles
    (elgamal/ProtocolElGamal.java:352-371, 413-434).  A seed has the digest's length, which selects the hash; the library's
    host functions against the Python restatement over hashlib."""
    lib = ctypes.CDLL(os.path.join(ROOT, "verificatum-vmn_amd", "libvmnhip.so"))
    for name, bits in (("sha384", 384), ("sha512", 512), ("sha256", 256)):
        seed = bytes(range(bits // 8))
        out = ctypes.create_string_buffer(1000)
        assert lib.vmn_prg_bytes(seed, ctypes.c_size_t(len(seed)), out, ctypes.c_size_t(1000)) == 0
        assert out.raw == pyref_prg.prg_bytes(seed, 1000, name)
        for data in (b"", b"abc", bytes(range(256)) * 3, b"x" * 111, b"y" * 112, b"z" * 127, b"w" * 128):     # padding boundaries of the 128-byte block
            for nout in (8 * (bits // 8), 65, 1000):
                nb = (nout + 7) // 8
                out = ctypes.create_string_buffer(nb)
                assert lib.vmn_random_oracle_hash(ctypes.c_int(bits), data, ctypes.c_size_t(len(data)), ctypes.c_int(nout), out) == 0
                assert out.raw == pyref_prg.random_oracle(data, nout, name), (name, len(data), nout)
    assert lib.vmn_random_oracle_hash(ctypes.c_int(224), b"", ctypes.c_size_t(0), ctypes.c_int(8), out) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [512, 2048, 3072])
def test_device_random_vector_and_generators(bits, vmn, gpu_ctx):
    grp, _ = load_golden(bits)
    p, q, g = grp["p"], grp["q"], grp["g"]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    seed = pyref_prg.random_oracle(b"generators" + bytes([bits % 251]), 256)
    for n, ebits in ((1, 256), (77, 256), (300, 100), (33, 37)):
        assert G.ringArrayFromPRG(seed, n, ebits).toInts() == pyref_prg.random_integers(seed, n, ebits)
    full = q.bit_length()                              # integers that can reach q are reduced
    assert G.ringArrayFromPRG(seed, 50, full).toInts() == [x % q for x in pyref_prg.random_integers(seed, 50, full)]
    for n, rbitlen in ((1, 100), (130, 100), (64, 0), (9, 3)):
        H = G.elementArrayFromPRG(seed, n, rbitlen)
        want = pyref_prg.modp_generators(seed, n, p, q, rbitlen)
        assert H.toInts() == want
        assert H.isMember()                            # squares: in the order-q subgroup


@pytest.mark.gpu
def test_parts_of_a_prg_array_without_the_rest(vmn, gpu_ctx):
    """vmn_rarray_from_prg_range / _gather (what a rank of a sharded proof generates: its positions and the rows it reads
    through the permutation) against slices of the Python PRG integers, over a ModP order and a curve order, for values
    narrower than, as wide as and wider than the order (one part, reduced, several parts)."""
    import random
    grp, _ = load_golden(2048)
    groups = [vmn.ModPGroup(gpu_ctx, grp["p"], grp["q"], grp["g"]), vmn.ECqPGroup(gpu_ctx, "P-256")]
    for G in groups:
        q = G.q
        for hashname, seedlen in (("sha256", 32), ("sha512", 64)):
            seed = pyref_prg.random_oracle(b"parts", 8 * seedlen, hashname)
            for bits in (37, 256, q.bit_length(), q.bit_length() + 100, 612):
                n = 500
                vb = (bits + 7) // 8
                stream = pyref_prg.prg_bytes(seed, n * vb, hashname)
                want = [(int.from_bytes(stream[i * vb:(i + 1) * vb], "big") & ((1 << bits) - 1)) % q for i in range(n)]
                assert G.ringArrayFromPRG(seed, n, bits).toInts() == want
                for first, cnt in ((0, 1), (0, n), (123, 77), (n - 1, 1), (499, 0), (64, 256)):
                    assert G.ringArrayFromPRGRange(seed, first, cnt, bits).toInts() == want[first:first + cnt], (bits, first, cnt)
                rng = random.Random(bits)
                idx = [rng.randrange(n) for _ in range(300)] + [0, n - 1, 7, 7]
                assert G.ringArrayFromPRGGather(seed, idx, bits).toInts() == [want[i] for i in idx], bits
                assert G.ringArrayFromPRGGather(seed, [], bits).toInts() == []


@pytest.mark.gpu
@pytest.mark.parametrize("hashname,seedlen", [("sha384", 48), ("sha512", 64)])
def test_device_generators_with_sha384_and_sha512_seeds(hashname, seedlen, vmn, gpu_ctx):
    grp, _ = load_golden(2048)
    p, q, g = grp["p"], grp["q"], grp["g"]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    seed = pyref_prg.random_oracle(b"wide-hash", 8 * seedlen, hashname)
    assert len(seed) == seedlen

    def ints(n, bits):
        vb = (bits + 7) // 8
        stream = pyref_prg.prg_bytes(seed, n * vb, hashname)
        return [int.from_bytes(stream[i * vb:(i + 1) * vb], "big") & ((1 << bits) - 1) for i in range(n)]
    for n, ebits in ((1, 256), (91, 256), (40, 613), (17, 2047 + 100)):
        assert G.ringArrayFromPRG(seed, n, ebits).toInts() == [x % q for x in ints(n, ebits)]
    H = G.elementArrayFromPRG(seed, 33, 100)
    assert H.toInts() == [pow(t % p, 2, p) for t in ints(33, p.bit_length() + 100)]


@pytest.mark.gpu
def test_random_vector_over_a_curve_order(vmn, gpu_ctx):
    G = vmn.ECqPGroup(gpu_ctx, "P-256")
    seed = pyref_prg.random_oracle(b"curve", 256)
    assert G.ringArrayFromPRG(seed, 200, 128).toInts() == pyref_prg.random_integers(seed, 200, 128)
    assert G.ringArrayFromPRG(seed, 200, 256).toInts() == [x % G.q for x in pyref_prg.random_integers(seed, 200, 256)]
    # wider than twice the order (the 612-bit epsilon of a proof, bits(q) + rbitlen random bits, three and four parts)
    for bits in (356, 612, 700, 1025):
        assert G.ringArrayFromPRG(seed, 77, bits).toInts() == [x % G.q for x in pyref_prg.random_integers(seed, 77, bits)], bits


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name,hashname,seedlen", [("P-256", "sha256", 32), ("P-384", "sha256", 32), ("P-256", "sha512", 64),
                                                         ("P-224", "sha256", 32), ("P-521", "sha256", 32)])
def test_independent_generators_over_curves(curve_name, hashname, seedlen, vmn, gpu_ctx):
    """IndependentGeneratorsRO over ECqPGroup (P-256 is the reference's default group): random points derived on the GPU,
    candidates tested in parallel and compacted in order, against the sequential Python restatement."""
    from oracle.pyref_ec import Curve
    c = Curve(curve_name)
    G = vmn.ECqPGroup(gpu_ctx, curve_name)
    seed = pyref_prg.random_oracle(b"curve-generators", 8 * seedlen, hashname)
    for n, rbitlen in ((1, 100), (2, 100), (257, 100), (1000, 50), (33, 0)):
        H = G.elementArrayFromPRG(seed, n, rbitlen)
        want = pyref_prg.ec_generators(seed, n, c, rbitlen, hashname)
        assert H.toInts() == want, (n, rbitlen)
        assert all(c.on_curve(P) for P in want)
    assert G.elementArrayFromPRG(seed, 0, 100).size() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["native"])
def test_proof_with_the_batching_vector_derived_from_a_seed(impl, vmn, gpu_ctx, entry):
    """setBatchVector(byte[] prgSeed) as the reference calls it: both drivers derive e on the GPU and produce the
    transcript of the oracle run on e = the PRG integers."""
    import importlib.util, sys
    from oracle import pyref_proofs as P
    from tape import Tape
    import mirror
    mods = mirror.load(entry, ("hvzk", "native"))
    hv = mods["hvzk" if impl == "python" else "native"]
    grp, _ = load_golden(512)
    p, q, g = grp["p"], grp["q"], grp["g"]
    NV, NE, NR, n = 100, 100, 50, 40
    seed = pyref_prg.random_oracle(b"posc-seed", 256)
    h = pyref_prg.modp_generators(pyref_prg.random_oracle(b"gens", 256), n, p, q, NR)
    t = Tape(b"seeded", q)
    pi, r, v = t.permutation(n), t.ring_array(n), t.int_array(1, NV)[0]
    e = pyref_prg.random_integers(seed, n, NE)
    u = P.permutation_commitment(g, h, r, pi, p)
    o = P.PoSC(p, q, NV, NE, NR, rand=Tape(b"pr", q))
    o.setInstance(g, h, u, r, pi)
    o.setBatchVector(e)
    com_o, rep_o = o.commit(), o.reply(v)
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    H = G.elementArrayFromPRG(pyref_prg.random_oracle(b"gens", 256), n, NR)
    assert H.toInts() == h
    U = G.toElementArray(u)
    pr = hv.PoSCBasicTW(G, NV, NE, NR, rand=Tape(b"pr", q))
    pr.setInstance(g, H, U, G.ringArray(r), pi)
    pr.setBatchVectorSeed(seed)
    com, rep = pr.commit(), pr.reply(v)
    ints = lambda x: x.toInts() if hasattr(x, "toInts") else x
    assert {k: ints(x) for k, x in com.items()} == com_o and {k: ints(x) for k, x in rep.items()} == rep_o
    ver = hv.PoSCBasicTW(G, NV, NE, NR)
    ver.setInstance(g, H, U)
    ver.setBatchVectorSeed(seed)
    ver.setCommitment(com)
    ver.setChallenge(v)
    assert ver.verify(rep)
