"""GPU suite: the BASELINE.json configurations and the reference's own protocol-test shapes, each on its own workload.

  configs[2]  3072-bit ModPGroup, commitment-consistent path: PermutationCommitment.precompute(N_max) + PoSC offline,
              then maxciph -> shrink(N) -> re-encrypt -> CCPoS plain and raised
              (P/mixnet/ShufflerElGamalSession.java:645-661, 673-712, 972-1038; P/mixnet/PermutationCommitment.java:390-471)
  configs[4]  ECqPGroup P-256, width 3, PoS and CCPoS
  widths 1-4  the reference's protocol test runs widths 1-4 with and without precomputation
              (P/mixnet/DemoShufflerElGamal.java:163-265)
  configs[0]  N = 10 000 ciphertexts over the 2048-bit group: whole transcript against the C + GMP oracle
plus the verifier-side guarantees: every bit of a received k_E is used, a batching vector wider than ebitlen is refused.
Every transcript is compared message by message with oracle/pyref_proofs.py on the same random tape, for both driver
implementations (the C++ drivers behind include/vmnproofs.h and their Python mirror)."""
import pytest

from conftest import load_golden
from oracle import pyref, pyref_proofs as P
from oracle.pyref_ec import Curve
from proof_cases import check_ccpos, check_pos, ints_of, load_driver_modules, make_instance, same_msg
from tape import Tape

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods(entry, vmn):
    return load_driver_modules(entry)


def modp(vmn, gpu_ctx, bits):
    grp, _ = load_golden(bits)
    p, q, g = grp["p"], grp["q"], grp["g"]
    return vmn.ModPGroup(gpu_ctx, p, q, g), P.ModPAdapter(p, q), p, q, g


def check_posc(impl, mods, G, p, q, g, h, H, u_o, U, r, R, pi, t, bits3):
    """A2 on the (full-size) permutation commitment; returns nothing, asserts transcript equality and verdicts."""
    NV, NE, NR = bits3
    hv = mods["hvzk" if impl == "python" else "native"]
    n = len(h)
    e = t.int_array(n, NE)
    v = t.int_array(1, NV)[0]
    o = P.PoSC(p, q, NV, NE, NR, rand=Tape(b"poscprover", q))
    o.setInstance(g, h, u_o, r, pi)
    o.setBatchVector(e)
    com_o, rep_o = o.commit(), o.reply(v)
    pr = hv.PoSCBasicTW(G, NV, NE, NR, rand=Tape(b"poscprover", q))
    pr.setInstance(g, H, U, R, pi)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    same_msg(com, com_o)
    same_msg(rep, rep_o)
    ver = hv.PoSCBasicTW(G, NV, NE, NR)
    ver.setInstance(g, H, U)
    ver.setBatchVector(e)
    ver.setCommitment(com)
    ver.setChallenge(v)
    assert ver.verify(rep)
    bad = dict(rep)
    bad["k_D"] = (rep["k_D"] + 1) % q
    assert not ver.verify(bad)
    ov = P.PoSC(p, q, NV, NE, NR)
    ov.setInstance(g, h, u_o)
    ov.setBatchVector(e)
    ov.setCommitment({k: ints_of(x) for k, x in com.items()})
    assert ov.verify({k: ints_of(x) for k, x in rep.items()}, v)


@pytest.mark.parametrize("impl", ["native"])
def test_config2_3072bit_precompute_shrink_ccpos(impl, vmn, gpu_ctx, mods):
    """BASELINE configs[2] on its own group size: offline phase for N_max ciphertexts, online phase for N < N_max."""
    bits3 = (256, 256, 100)
    G, K, p, q, g = modp(vmn, gpu_ctx, 3072)
    n_max, n, width = 36, 23, 1
    mx, nat = mods["mixnet"], mods["native"]
    h, pkey, w_max, t = make_instance(K, g, n_max, width, b"cfg2")
    H = G.toElementArray(h)
    # ---- offline (vmn -precomp): permutation commitment for N_max, raised commitment / generators, PoSC
    pi, r = t.permutation(n_max), t.ring_array(n_max)
    rho = t.int_array(1, mx.RAISED_BITLENGTH)[0]
    u_o = P.permutation_commitment(g, h, r, pi, p)
    pc = mx.PermutationCommitment(G, H)
    U = pc.precompute(r, pi)
    assert U.toInts() == u_o
    assert pc.raise_(rho).toInts() == K.exp_scalar(u_o, rho)
    RH = mx.raisedGenerators(H, rho)
    assert RH.toInts() == K.exp_scalar(h, rho)
    check_posc(impl, mods, G, p, q, g, h, H, u_o, U, r, pc.exponents, pi, t, bits3)
    # ---- online: only n ciphertexts arrive -> shrink (ShufflerElGamalSession.java:673-712)
    keep_o, pi_s_o = P.shrink_permutation(pi, n)
    assert sum(keep_o) == n and sorted(pi_s_o) == list(range(n))
    if impl == "native":
        keep, pi_s = nat.permutation_shrink_native(pi, n)
        assert (keep, pi_s) == (keep_o, pi_s_o)
        # a verifier's view of a keep list from the bulletin board (:424-447)
        assert nat.keep_list_sanitize_native(keep, n_max, n) == (keep, False)
        assert nat.keep_list_sanitize_native(keep[:-1], n_max, n) == ([i < n for i in range(n_max)], True)
        wrong = list(keep)
        wrong[wrong.index(False)] = True
        assert nat.keep_list_sanitize_native(wrong, n_max, n) == ([i < n for i in range(n_max)], True)
        U_s, RU_s = U.extract(keep), pc.raisedCommitment.extract(keep)
        R_s = pc.exponents.copyOfRange(0, n)
    else:
        keep = pc.shrink(n)
        assert keep == keep_o and list(pc.permutation) == pi_s_o
        assert P.sanitize_keep_list(keep[:-1], n_max, n) == [i < n for i in range(n_max)]
        U_s, RU_s, R_s, pi_s = pc.commitment, pc.raisedCommitment, pc.exponents, pc.permutation
    u_s_o = P.extract(u_o, keep_o)
    assert U_s.toInts() == u_s_o == P.permutation_commitment(g, h[:n], r[:n], pi_s_o, p)      # still a commitment, to pi_s
    assert RU_s.toInts() == K.exp_scalar(u_s_o, rho)
    H_s, RH_s = H.copyOfRange(0, n), RH.copyOfRange(0, n)
    assert RH_s.toInts() == K.exp_scalar(h[:n], rho)
    # ---- committed shuffle of the n ciphertexts: re-encrypt, CCPoS plain and raised
    w = [c[:n] for c in w_max]
    s = [t.ring_array(n) for _ in range(width)]
    W, S = [G.toElementArray(c) for c in w], [G.ringArray(c) for c in s]
    wp_o = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi_s_o)
    if impl == "native":
        WP = nat.reencrypt_native(G, pkey, W, S, pi_s)
        # the precomputed form (ShufflerElGamalSession.java:645-661, 673-712, 789-792): factors for more ciphertexts than
        # arrive, cut with copyOfRange(0, n), applied when the ciphertexts are there -- the same w'
        s_long = [c + t.ring_array(5) for c in s]
        F = nat.reencryption_factors_native(G, pkey, [G.ringArray(c) for c in s_long])
        assert [f.toInts()[:n] for f in F] == P.g_reenc_factors(K, pkey, s)
        WP2 = nat.apply_factors_native(G, W, [f.copyOfRange(0, n) for f in F], pi_s)
        assert [c.toInts() for c in WP2] == wp_o
    else:
        WP = mx.reencrypt(W, mx.reencFactors(G, pkey, S), pi_s)
    assert [c.toInts() for c in WP] == wp_o
    check_ccpos(impl, mods, G, K, g, h[:n], H_s, u_s_o, U_s, pkey, w, W, wp_o, WP, r[:n], R_s, pi_s_o, s, S, t, bits3, rho=rho)


@pytest.mark.parametrize("impl", ["native"])
@pytest.mark.parametrize("width", [3, 4])
def test_modp_widths_3_and_4(width, impl, vmn, gpu_ctx, mods):
    """Widths 3 and 4 (DemoShufflerElGamal.java:163-265 runs 1-4), PoS and CCPoS plain + raised, 512-bit group."""
    bits3 = (100, 100, 50)
    G, K, p, q, g = modp(vmn, gpu_ctx, 512)
    n = 19
    h, pkey, w, t = make_instance(K, g, n, width, b"width%d" % width)
    H, W, WP, wp_o, s, S, pi = check_pos(impl, mods, G, K, g, h, pkey, w, t, bits3)
    r = t.ring_array(n)
    rho = t.int_array(1, 50)[0]
    u_o = P.g_permutation_commitment(K, g, h, r, pi)
    R = G.ringArray(r)
    U = mods["native"].permutation_commitment_native(G, g, H, R, pi)
    assert U.toInts() == u_o
    check_ccpos(impl, mods, G, K, g, h, H, u_o, U, pkey, w, W, wp_o, WP, r, R, pi, s, S, t, bits3, rho=rho)


@pytest.mark.parametrize("impl", ["native"])
def test_config4_p256_width3(impl, vmn, gpu_ctx, mods):
    """BASELINE configs[4]'s group and width: ECqPGroup P-256, width 3 (six point arrays per ciphertext array)."""
    bits3 = (128, 128, 64)
    c = Curve("P-256")
    K = P.ECAdapter(c)
    G = vmn.ECqPGroup(gpu_ctx, "P-256")
    n, width = 17, 3
    h, pkey, w, t = make_instance(K, c.g, n, width, b"cfg4")
    H, W, WP, wp_o, s, S, pi = check_pos(impl, mods, G, K, c.g, h, pkey, w, t, bits3)
    r = t.ring_array(n)
    rho = t.int_array(1, 50)[0]
    u_o = P.g_permutation_commitment(K, c.g, h, r, pi)
    R = G.ringArray(r)
    U = mods["native"].permutation_commitment_native(G, c.g, H, R, pi)
    assert U.toInts() == u_o
    check_ccpos(impl, mods, G, K, c.g, h, H, u_o, U, pkey, w, W, wp_o, WP, r, R, pi, s, S, t, bits3, rho=rho)


def test_config0_size_10000_ciphertexts_2048bit(vmn, gpu_ctx, mods, oracle_for):
    """BASELINE configs[0]'s workload (N = 10 000, 2048-bit ModPGroup, width 1) through the C++ drivers; the oracle is
    the C + GMP library on all host cores (the reference's Java/GMP path cannot run here: no JDK / VCR)."""
    from oracle.cbind import GmpAdapter
    grp, _ = load_golden(2048)
    p, q, g = grp["p"], grp["q"], grp["g"]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    # (fixed-base exponentiations through the oracle's precomputed tables, what VCR + GMPMEE's fpowm do: the same values as
    # mpz_powm per element -- tests/test_proofs_oracle.py pins the two against each other -- in a fifth of the time)
    K = GmpAdapter(oracle_for(p, q), pippenger_c=10, fixed_tables=True)
    n = 10_000
    h, pkey, w, t = make_instance(K, g, n, 1, b"cfg0")
    check_pos("native", mods, G, K, g, h, pkey, w, t, (256, 256, 100))


def test_proof_of_shuffle_over_rfc3526_group_18(vmn, gpu_ctx, mods, oracle_for):
    """Above north_star's range but inside the reference's (safe primes up to 15 424 bits): a whole PoS transcript over
    the 8192-bit group (eight lanes per element; single elements on 128 host limbs) against the C + GMP oracle."""
    from oracle import pyref
    from oracle.cbind import GmpAdapter
    p, q, g = pyref.modp_group(8192)
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    K = GmpAdapter(oracle_for(p, q))
    h, pkey, w, t = make_instance(K, g, 8, 1, b"rfc18")
    check_pos("native", mods, G, K, g, h, pkey, w, t, (256, 256, 100))


class WideEpsilonTape(Tape):
    """A prover whose epsilon is uniform in Z_q instead of n_e + n_v + n_r bits: its proofs are still valid (the
    verification equations hold for every epsilon), and its k_E fills the whole field."""

    def int_array(self, n, bits):
        return self.ring_array(n) if bits > 200 else Tape.int_array(self, n, bits)


@pytest.mark.parametrize("impl", ["native"])
def test_verifier_uses_every_bit_of_received_exponents(impl, vmn, gpu_ctx, mods):
    """The reference parses k_E as full field elements and uses every bit (PoSBasicTW.java:985-989, 1021, 1032;
    CCPoSBasicW.java:533-544, 554-580): a VALID proof whose k_E is wider than an honest prover's must be accepted, and
    a change in a high bit of k_E must be seen.  (A verifier that truncated k_E to n_e + n_v + n_r + 1 bits would fail
    both.)"""
    bits3 = NV, NE, NR = (100, 100, 50)
    G, K, p, q, g = modp(vmn, gpu_ctx, 512)
    hv = mods["hvzk" if impl == "python" else "native"]
    n, width = 29, 1
    h, pkey, w, t = make_instance(K, g, n, width, b"fullwidth")
    pi, s = t.permutation(n), [t.ring_array(n)]
    e, v = t.int_array(n, NE), t.int_array(1, NV)[0]
    r, rho = t.ring_array(n), t.int_array(1, 50)[0]
    wp_o = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi)
    H, W, WP = G.toElementArray(h), [G.toElementArray(c) for c in w], [G.toElementArray(c) for c in wp_o]
    # ---- PoS by the oracle's prover with a full-width epsilon
    o = P.GPoS(K, NV, NE, NR, rand=WideEpsilonTape(b"wide", q))
    o.precompute(g, h, pi)
    o.setInstance(pkey, w, wp_o, s)
    o.setBatchVector(e)
    com, rep = o.commit(), o.reply(v)
    assert max(x.bit_length() for x in rep["k_E"]) > NV + NE + NR + 1 + 100
    ov = P.GPoS(K, NV, NE, NR)
    ov.precompute(g, h)
    ov.u = o.u
    ov.setInstance(pkey, w, wp_o)
    ov.setBatchVector(e)
    ov.computeAF()
    ov.setCommitment(com)
    assert ov.verify(rep, v)
    ver = hv.PoSBasicTW(G, NV, NE, NR)
    ver.precompute(g, H)
    ver.setPermutationCommitment(G.toElementArray(o.u))
    ver.setInstance(pkey, W, WP)
    ver.setBatchVector(e)
    ver.computeAF()
    ver.setCommitment({k: (G.toElementArray(x) if isinstance(x, list) and k in ("B", "Bp") else x) for k, x in com.items()})
    ver.setChallenge(v)
    as_reply = lambda rp: {k: (G.ringArray(x) if k in ("k_B", "k_E") else x) for k, x in rp.items()}
    assert ver.verify(as_reply(rep)) and ver.verdicts == (True,) * 5
    bad = dict(rep)
    bad["k_E"] = list(rep["k_E"])
    bad["k_E"][n // 3] ^= 1 << 400                       # far above an honest prover's 251 bits
    bad["k_E"][n // 3] %= q
    assert not ov.verify(bad, v) and ov.verdicts == (False, False, True, True, False)
    assert not ver.verify(as_reply(bad)) and ver.verdicts == ov.verdicts
    # ---- CCPoS, plain and raised
    u_o = P.g_permutation_commitment(K, g, h, r, pi)
    U = G.toElementArray(u_o)
    oc = P.GCCPoS(K, NV, NE, NR, rand=WideEpsilonTape(b"widecc", q))
    oc.setInstance(g, h, u_o, pkey, w, wp_o, r, pi, s)
    oc.setBatchVector(e)
    cc, cr = oc.commit(), oc.reply(v)
    assert max(x.bit_length() for x in cr["k_E"]) > 400
    cbad = dict(cr)
    cbad["k_E"] = list(cr["k_E"])
    cbad["k_E"][0] = (cbad["k_E"][0] ^ (1 << 300)) % q
    for raised in (False, True):
        cv = hv.CCPoSBasicW(G, NV, NE, NR)
        cv.setInstance(g, H, U, pkey, W, WP)
        cv.setBatchVector(e)
        cv.setCommitment(cc)
        cv.setChallenge(v)
        as_cr = lambda rp: {k: (G.ringArray(x) if k == "k_E" else x) for k, x in rp.items()}
        if raised:
            cv.computeAB(U.exp(rho))
            RH = H.exp(rho)
            assert cv.verify(as_cr(cr), RH, rho) and not cv.verify(as_cr(cbad), RH, rho)
        else:
            cv.computeAB()
            assert cv.verify(as_cr(cr)) and not cv.verify(as_cr(cbad))


def test_batching_vector_wider_than_ebitlen_is_refused(vmn, gpu_ctx, mods):
    """An explicit batching vector with an entry of more than ebitlen bits would make A / F (computed over ebitlen bits)
    and D (the full product) disagree silently: the drivers refuse it with VMN_ERR_FORMAT."""
    nat = mods["native"]
    G, K, p, q, g = modp(vmn, gpu_ctx, 512)
    n = 12
    h, pkey, w, t = make_instance(K, g, n, 1, b"wide-e")
    ver = nat.PoSBasicTW(G, 100, 100, 50)
    ver.precompute(g, G.toElementArray(h))
    e = t.int_array(n, 100)
    ver.setBatchVector(e)                                  # fine
    e[5] |= 1 << 100
    with pytest.raises(vmn.VmnError) as ei:
        ver.setBatchVector(e)
    assert ei.value.status == -4


def test_default_wire_widths_are_the_references(vmn, gpu_ctx, mods, entry):
    """Without an explicit width a group uses the reference's: Java's BigInteger.toByteArray().length of p for group
    elements and of q for exponents.  For an RFC 3526 group (top bit of p set) they differ: 257 and 256 bytes at 2048
    bits -- the in-tree fixture (a 15 492-bit p, 1 937-byte leaves for p, q and g alike) cannot show that."""
    import importlib.util, os, sys
    spec = importlib.util.spec_from_file_location("verificatum_vmn_amd.eio", os.path.join(entry.PKG_DIR, "eio.py"))
    eio = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = eio
    spec.loader.exec_module(eio)
    nat = mods["native"]
    for bits, ew, xw in ((2048, 257, 256), (3072, 385, 384), (512, 65, 64)):
        grp, _ = load_golden(bits)
        p, q, g = grp["p"], grp["q"], grp["g"]
        assert len(p.to_bytes(p.bit_length() // 8 + 1, "big", signed=True)) == ew       # Java's width
        G = vmn.ModPGroup(gpu_ctx, p, q, g)
        assert (G.nbytes, G.exp_bytes) == (ew, xw)
        t = Tape(b"widths%d" % bits, q)
        xs, es = [pow(g, x, p) for x in t.ring_array(5)], t.ring_array(5)
        X, E = G.toElementArray(xs), G.ringArray(es)
        assert X.toByteTree() == eio.encode([eio.int_leaf(x, ew) for x in xs])
        assert E.toByteTree() == eio.encode([eio.int_leaf(x, xw) for x in es])
        assert G.ringArrayFromByteTree(E.toByteTree()).toInts() == es
        assert X.exp(E).toInts() == [pow(x, e, p) for x, e in zip(xs, es)]
    # a whole reply: ring leaves are exponent-wide, element leaves element-wide
    n, NV, NE, NR = 6, 100, 100, 50
    h = pyref.exp_fixed(g, t.ring_array(n), p)
    r, pi, e, v = t.ring_array(n), t.permutation(n), t.int_array(n, NE), t.int_array(1, NV)[0]
    H, R = G.toElementArray(h), G.ringArray(r)
    U = nat.permutation_commitment_native(G, g, H, R, pi)
    pr = nat.PoSCBasicTW(G, NV, NE, NR, rand=Tape(b"w", q))
    pr.setInstance(g, H, U, R, pi)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    el, rl = (lambda x: eio.int_leaf(x, ew)), (lambda x: eio.int_leaf(x, xw))
    assert com.native.toByteTree() == eio.encode([[el(x) for x in com["B"].toInts()], el(com["Ap"]), [el(x) for x in com["Bp"].toInts()],
                                                  el(com["Cp"]), el(com["Dp"])])
    assert rep.native.toByteTree() == eio.encode([rl(rep["k_A"]), [rl(x) for x in rep["k_B"].toInts()], rl(rep["k_C"]), rl(rep["k_D"]),
                                                  [rl(x) for x in rep["k_E"].toInts()]])


@pytest.mark.parametrize("bits,bits3", [(512, (100, 100, 50)), (2048, (256, 256, 100)), (3072, (256, 256, 100))])
def test_prover_randomness_expanded_on_the_device(bits, bits3, vmn, gpu_ctx, mods):
    """The N-sized draws of a prover (r, s, b, beta, epsilon; PoSBasicTW.java:446, 473, 583, 612;
    ShufflerElGamalSession.java:408-409) generated on the GPU from 32-byte seeds: the transcript equals the oracle's
    run on the same seeds expanded with the Python PRG."""
    from tape import SeedTape
    nat = mods["native"]
    NV, NE, NR = bits3
    G, K, p, q, g = modp(vmn, gpu_ctx, bits)
    n, width = 41, 2
    h, pkey, w, t = make_instance(K, g, n, width, b"devrand%d" % bits)
    pi = t.permutation(n)
    e, v = t.int_array(n, NE), t.int_array(1, NV)[0]
    s_tape_o, s_tape = SeedTape(b"s", q, NR, expanding=True), SeedTape(b"s", q, NR)
    s = [s_tape_o.ring_array(n) for _ in range(width)]
    S = [nat.random_ring_array_native(G, s_tape, n, NR) for _ in range(width)]
    assert [x.toInts() for x in S] == s and all(0 <= x < q for col in s for x in col)
    o = P.GPoS(K, NV, NE, NR, rand=SeedTape(b"prover", q, NR, expanding=True))
    o.precompute(g, h, pi)
    wp_o = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi)
    o.setInstance(pkey, w, wp_o, s)
    o.setBatchVector(e)
    com_o, rep_o = o.commit(), o.reply(v)
    H, W = G.toElementArray(h), [G.toElementArray(c) for c in w]
    pr = nat.PoSBasicTW(G, NV, NE, NR, rand=SeedTape(b"prover", q, NR))
    pr.precompute(g, H, pi)
    assert pr.u.toInts() == o.u
    WP = nat.reencrypt_native(G, pkey, W, S, pi)
    pr.setInstance(pkey, W, WP, S)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    same_msg(com, com_o)
    same_msg(rep, rep_o)
    ver = nat.PoSBasicTW(G, NV, NE, NR)
    ver.precompute(g, H)
    ver.setPermutationCommitment(pr.u)
    ver.setInstance(pkey, W, WP)
    ver.setBatchVector(e)
    ver.computeAF()
    ver.setCommitment(com)
    ver.setChallenge(v)
    assert ver.verify(rep)


def test_secure_random_source_is_the_default_kind(vmn, gpu_ctx, mods):
    """mixnet.SecureRandomSource (os.urandom; arrays expanded on the device): an honest proof with it verifies, and two
    runs differ."""
    nat, mx = mods["native"], mods["mixnet"]
    G, K, p, q, g = modp(vmn, gpu_ctx, 512)
    NV, NE, NR, n = 100, 100, 50, 25
    h, pkey, w, t = make_instance(K, g, n, 1, b"secure")
    H = G.toElementArray(h)
    pi, e, v = t.permutation(n), t.int_array(n, NE), t.int_array(1, NV)[0]
    seen = []
    for _ in range(2):
        src = mx.SecureRandomSource(q, NR)
        R = nat.random_ring_array_native(G, src, n, NR)
        U = nat.permutation_commitment_native(G, g, H, R, pi)
        pr = nat.PoSCBasicTW(G, NV, NE, NR, rand=src)
        pr.setInstance(g, H, U, R, pi)
        pr.setBatchVector(e)
        com, rep = pr.commit(), pr.reply(v)
        ver = nat.PoSCBasicTW(G, NV, NE, NR)
        ver.setInstance(g, H, U)
        ver.setBatchVector(e)
        ver.setCommitment(com)
        ver.setChallenge(v)
        assert ver.verify(rep)
        seen.append(rep["k_A"])
        assert all(0 <= x < q for x in R.toInts())
    assert seen[0] != seen[1]
