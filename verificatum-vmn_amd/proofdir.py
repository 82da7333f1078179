"""proofdir.py — the reference's proof directory ("nizkp") around one shuffle, written and verified by the C++ drivers.

The reference's standalone verifier reads everything it checks from files (``vmnv``):

    <nizkp>/Ciphertexts.bt                       the input list L_0       mixnet/MixNetElGamalSession.java:391-393
    <nizkp>/FullPublicKey.bt                     pk = (g, y)              (read by the verifier's readPublicKey)
    <nizkp>/proofs/PermutationCommitment%02d.bt  u of party l             hvzk/PoSTW.java:281-284
    <nizkp>/proofs/PoSCommitment%02d.bt          (B, A', B', C', D', F')  hvzk/PoSTW.java:293-296
    <nizkp>/proofs/PoSReply%02d.bt               (k_A, ..., k_F)          hvzk/PoSTW.java:305-307
    <nizkp>/proofs/Ciphertexts%02d.bt            the output list L_l      mixnet/ShufflerElGamalSession.java:1077-1079

``write_shuffle`` is what ``PoSTW.prove`` does with ``nizkp != null`` (hvzk/PoSTW.java:95-165: publish u, derive the seed of
the batching vector from node(g, h, u, pk, w, w'), commit, derive the challenge from node(leaf(seed), commitment), reply --
each message also written to its file); ``verify_shuffle`` is ``MixNetElGamalVerifyFiatShamirSession.verifyPoS``
(:843-937) including the values it prints under ``vmnv -t`` (``checkPrintTestVector``: der.rho, PoS.s, PoS.A, PoS.F, PoS.B,
PoS.Ap ... PoS.Fp, PoS.v, PoS.C, PoS.D, PoS.k_A ... PoS.k_F).  Both run the proof on the GPU through
``native.PoSBasicTW``; the hashing is hashlib on the host over the files' own bytes (they ARE the byte trees).

What the reference takes from its protocol-info XML (session id, bit lengths, the descriptions of group / PRG / hash
that go into the global prefix, :158-189) is kept here in ``params.json`` next to the files; the XML itself and the
bulletin board are out of scope.
"""
from __future__ import annotations

import hashlib
import json
import os
from typing import Dict, List, Optional, Sequence

from . import PGroupElementArray
from . import fiatshamir as fs
from . import native

PARAMS = "params.json"


def _p(nizkp: str, *names) -> str:
    return os.path.join(nizkp, *names)


def pc_file(nizkp, l):
    return _p(nizkp, "proofs", "PermutationCommitment%02d.bt" % l)          # PoSTW.PCfile :281-284


def posc_file(nizkp, l):
    return _p(nizkp, "proofs", "PoSCommitment%02d.bt" % l)                  # PoSTW.PoSCfile :293-296


def posr_file(nizkp, l):
    return _p(nizkp, "proofs", "PoSReply%02d.bt" % l)                       # PoSTW.PoSRfile :305-307


def l_file(nizkp, l):
    """The list party l reads: L_0 at the top of the directory, L_l below proofs/ (ShufflerElGamalSession.Lfile :1077-1079)."""
    return _p(nizkp, "Ciphertexts.bt") if l == 0 else _p(nizkp, "proofs", "Ciphertexts%02d.bt" % l)


def pk_file(nizkp):
    return _p(nizkp, "FullPublicKey.bt")


# ---- the global prefix and what is derived from it ----------------------------------------------------------------------
def global_prefix(params: dict) -> bytes:
    """``setGlobalPrefix`` (MixNetElGamalVerifyFiatShamirSession.java:158-189): rho = H(bytetree(node(version, sid.auxsid,
    n_r, n_v, n_e, s_PRG, s_Gq, s_H))) -- strings as leaves of their bytes, integers as 4-byte leaves."""
    s = lambda x: fs.leaf(x.encode("utf-8"))
    i = lambda x: fs.leaf(int(x).to_bytes(4, "big"))
    rosid = params["sid"] + "." + params["auxsid"]
    parts = [s(params["version"]), s(rosid), i(params["rbitlen"]), i(params["vbitlenro"]), i(params["ebitlenro"]),
             s(params["prg"]), s(params["pgroup"]), s(params["rohash"])]
    return hashlib.new(_hashname(params), fs._hdr(0, len(parts)) + b"".join(parts)).digest()


def _hashname(params: dict) -> str:
    return {"SHA-256": "sha256", "SHA-384": "sha384", "SHA-512": "sha512"}[params.get("rohash_name", "SHA-256")]


def derive_generators(grp, params: dict, rho: bytes, n: int) -> PGroupElementArray:
    """``IndependentGeneratorsRO("generators", H, rho, n_r).generate(pGroup, n)`` (distr/IndependentGeneratorsRO.java:110-130;
    called at MixNetElGamalVerifyFiatShamirSession.java:557-566): seed = RO(rho || leaf("generators")), then
    pGroup.randomElementArray(n, PRG(seed), n_r) on the GPU (vmn_garray_from_prg)."""
    hn = _hashname(params)
    seed_bits = 8 * hashlib.new(hn).digest_size
    seed = fs.Challenger(rho, hn).challenge(fs.leaf(b"generators"), seed_bits)
    return grp.elementArrayFromPRG(seed, n, int(params["rbitlen"]))


# ---- arrays as files ------------------------------------------------------------------------------------------------------
def write_ciphertexts(path: str, comps: Sequence[PGroupElementArray]) -> None:
    """A PPGroupElementArray of width w as 2w component arrays: node(u-part, v-part), a part being the array's own tree
    at width 1 and node(w array trees) otherwise."""
    half = len(comps) // 2
    with open(path, "wb") as f:
        f.write(fs._hdr(0, 2))
        for part in (comps[:half], comps[half:]):
            if half > 1:
                f.write(fs._hdr(0, half))
            for a in part:
                f.write(a.toByteTree())


def read_ciphertexts(grp, path: str, width: int, expected_n: int = 0) -> Optional[List[PGroupElementArray]]:
    """The 2w component arrays of a ciphertext list file, parsed (range + membership) on the GPU; None when the file is not
    such a list (the verifier then fails the party, :1446-1460)."""
    with open(path, "rb") as f:
        buf = f.read()
    pos = 0

    def header(tag, count=None):
        nonlocal pos
        if len(buf) < pos + 5 or buf[pos] != tag:
            raise ValueError("byte tree framing")
        c = int.from_bytes(buf[pos + 1:pos + 5], "big")
        if count is not None and c != count:
            raise ValueError("byte tree child count")
        pos += 5
        return c
    try:
        header(0, 2)
        comps = []
        for _ in range(2):
            if width > 1:
                header(0, width)
            for _ in range(width):
                n = int.from_bytes(buf[pos + 1:pos + 5], "big")
                if buf[pos] != 0 or (expected_n and n != expected_n):
                    raise ValueError("array node")
                size = 5 + n * (5 + grp.elem_bytes)
                comps.append(grp.toElementArrayFromByteTree(buf[pos:pos + size], n))
                pos += size
        if pos != len(buf) or len({c.size() for c in comps}) != 1:
            raise ValueError("trailing bytes")
        return comps
    except (ValueError, native.VmnError):
        return None


def _hex(x) -> str:
    """A group element / ring element as text: hexadecimal for integers, (x, y) in hexadecimal for curve points.  VCR's own
    toString() of these objects is not part of the reference tree: a maintainer who diffs against `vmnv -t` compares VALUES."""
    if x is None:
        return "INFINITY"
    if isinstance(x, tuple):
        return "(" + ", ".join(format(int(c), "x") for c in x) + ")"
    return format(int(x), "x")


def _arr(a) -> str:
    return "(" + ", ".join(_hex(x) for x in a.toInts()) + ")"


# ---- prover: PoSTW.prove with nizkp != null ----------------------------------------------------------------------------
def write_shuffle(nizkp: str, l: int, grp, params: dict, pkey: Sequence, W: Sequence[PGroupElementArray], rand,
                  H: Optional[PGroupElementArray] = None) -> List[PGroupElementArray]:
    """Party l shuffles the list in l_file(nizkp, l - 1) (given as W) and proves it: re-encryption + permutation
    (ShufflerElGamalSession.java:400-409, 273-278), then PoSTW.prove (:95-165).  Returns w'; writes L_l and the three
    proof files.  `rand`: the prover's random source (native.RandomSource semantics)."""
    os.makedirs(_p(nizkp, "proofs"), exist_ok=True)
    NV, NE, NR = int(params["vbitlenro"]), int(params["ebitlenro"]), int(params["rbitlen"])
    n, width = W[0].size(), len(W) // 2
    hn = _hashname(params)
    rho = global_prefix(params)
    chal = fs.Challenger(rho, hn)
    own_h = H is None
    if own_h:
        H = derive_generators(grp, params, rho, n)
    g = grp.g
    pi = rand.permutation(n)
    S = [native.random_ring_array_native(grp, rand, n, NR) for _ in range(width)]
    prover = native.PoSBasicTW(grp, NV, NE, NR, rand=rand)
    prover.precompute(g, H, pi)
    WP = native.reencrypt_native(grp, pkey, W, S, pi)
    write_ciphertexts(l_file(nizkp, l), WP)
    prover.setInstance(pkey, W, WP, S)
    u_bt = prover.u.toByteTree()
    with open(pc_file(nizkp, l), "wb") as f:                                     # "PermutationCommitment" :107-112
        f.write(u_bt)
    d = chal.start(8 * hashlib.new(hn).digest_size)                            # the seed of the batching vector :114-129
    d.update(fs._hdr(0, 6) + fs.leaf(grp.enc_el(g)))
    d.update(H.toByteTree())
    d.update(u_bt)
    d.update(fs.element_tree(grp, pkey))
    for path in (l_file(nizkp, l - 1), l_file(nizkp, l)):
        with open(path, "rb") as f:
            d.update(f.read())
    seed = chal.finish(d, 8 * hashlib.new(hn).digest_size)
    prover.setBatchVectorSeed(seed)
    com = prover.commit()
    com_bt = com.native.toByteTree()
    with open(posc_file(nizkp, l), "wb") as f:                                   # "Commitment" :131-139
        f.write(com_bt)
    v = int.from_bytes(chal.challenge(fs._hdr(0, 2) + fs.leaf(seed) + com_bt, NV), "big")     # :141-149
    rep = prover.reply(v)
    with open(posr_file(nizkp, l), "wb") as f:                                   # "Reply" :151-159
        f.write(rep.native.toByteTree())
    com = rep = None
    prover.free()
    for a in S:
        a.free()
    if own_h:
        H.free()
    return WP


def write_inputs(nizkp: str, grp, params: dict, pkey: Sequence, W: Sequence[PGroupElementArray]) -> None:
    """params.json, FullPublicKey.bt and the input list L_0."""
    os.makedirs(_p(nizkp, "proofs"), exist_ok=True)
    with open(_p(nizkp, PARAMS), "w") as f:
        json.dump(params, f, indent=1)
    with open(pk_file(nizkp), "wb") as f:
        f.write(fs.element_tree(grp, pkey))
    write_ciphertexts(l_file(nizkp, 0), W)


# ---- verifier: MixNetElGamalVerifyFiatShamirSession.verifyPoS -----------------------------------------------------------
def verify_shuffle(nizkp: str, l: int, grp, params: dict, pkey: Sequence, vectors: Optional[Dict[str, str]] = None,
                   with_arrays: bool = False) -> bool:
    """verifyPoS(l, g, generators, input = L_(l-1), output = L_l) (:843-937).  `vectors` (a dict) receives the test vectors
    the reference prints under `vmnv -t` at the places it prints them; `with_arrays` adds the N-sized ones (PoS.B, PoS.Bp,
    PoS.k_B, PoS.k_E, bas.h).  A file that cannot be parsed makes the verdict False (the reference substitutes trivial values
    and the equations then fail)."""
    tv = vectors if vectors is not None else {}
    NV, NE, NR = int(params["vbitlenro"]), int(params["ebitlenro"]), int(params["rbitlen"])
    width = len(pkey) // 2
    hn = _hashname(params)
    rho = global_prefix(params)
    tv["der.rho"] = rho.hex()                                                  # :188
    chal = fs.Challenger(rho, hn)
    W = read_ciphertexts(grp, l_file(nizkp, l - 1), width)
    if W is None:
        return False
    n = W[0].size()
    WP = read_ciphertexts(grp, l_file(nizkp, l), width, n)
    if WP is None:
        return False
    H = derive_generators(grp, params, rho, n)                                 # :557-566
    if with_arrays:
        tv["bas.h"] = _arr(H)
    g = grp.g
    V = native.PoSBasicTW(grp, NV, NE, NR)
    V.precompute(g, H)                                                         # :857
    V.setInstance(pkey, W, WP)
    with open(pc_file(nizkp, l), "rb") as f:                                   # :861-866
        u_bt = f.read()
    try:
        U = grp.toElementArrayFromByteTree(u_bt, n)
        if not U.isMember():
            raise ValueError("u outside the group")
    except (ValueError, native.VmnError):
        return False
    V.setPermutationCommitment(U)
    d = chal.start(8 * hashlib.new(hn).digest_size)                            # :869-879
    d.update(fs._hdr(0, 6) + fs.leaf(grp.enc_el(g)))
    d.update(H.toByteTree())
    d.update(u_bt)
    d.update(fs.element_tree(grp, pkey))
    for path in (l_file(nizkp, l - 1), l_file(nizkp, l)):
        with open(path, "rb") as f:
            d.update(f.read())
    seed = chal.finish(d, 8 * hashlib.new(hn).digest_size)
    tv["PoS.s"] = seed.hex()                                                   # :881
    V.setBatchVectorSeed(seed)                                                 # :883
    V.computeAF()                                                              # :886
    tv["PoS.A"] = _hex(V.getA())                                               # :888-889
    tv["PoS.F"] = "(" + ", ".join(_hex(x) for x in V.getF()) + ")"
    with open(posc_file(nizkp, l), "rb") as f:                                 # :892-895
        com_bt = f.read()
    com = V.readCommitment(com_bt, n, width)
    if com is None:
        return False
    V.setCommitment(com)
    if with_arrays:                                                            # :897-902
        tv["PoS.B"], tv["PoS.Bp"] = _arr(com.item(0)), _arr(com.item(2))
    tv["PoS.Ap"], tv["PoS.Cp"], tv["PoS.Dp"] = (_hex(com.item(k)[0]) for k in (1, 3, 4))
    tv["PoS.Fp"] = "(" + ", ".join(_hex(x) for x in com.item(5)) + ")"
    vch = int.from_bytes(chal.challenge(fs._hdr(0, 2) + fs.leaf(seed) + com_bt, NV), "big")      # :905-913
    tv["PoS.v"] = format(vch, "x")                                             # :915
    V.setChallenge(vch)
    with open(posr_file(nizkp, l), "rb") as f:                                 # :921-925
        rep = V.readReply(f.read(), n, width)
    if rep is None:
        return False
    verdict = V.verify(rep)
    tv["PoS.C"], tv["PoS.D"] = _hex(V.getC()), _hex(V.getD())                  # :927-928
    tv["PoS.k_A"], tv["PoS.k_C"], tv["PoS.k_D"] = (_hex(rep.item(k)[0]) for k in (0, 2, 3))      # :930-934
    tv["PoS.k_F"] = "(" + ", ".join(_hex(x) for x in rep.item(5)) + ")"
    if with_arrays:
        tv["PoS.k_B"], tv["PoS.k_E"] = _arr(rep.item(1)), _arr(rep.item(4))
    tv["verdicts(A,B,C,D,F)"] = str(V.verdicts)
    com = rep = None
    V.free()
    for a in W + WP + [H, U]:
        a.free()
    return bool(verdict)


# =============================================================================================================================
# The precomputed shuffle: permutation commitment + PoSC offline, shrink + re-encryption + CCPoS online (BASELINE configs[2], [4])
#     <nizkp>/proofs/PermutationCommitment%02d.bt   u for N_0 ciphertexts     mixnet/PermutationCommitment.java:228-230, 364
#     <nizkp>/proofs/PoSCCommitment%02d.bt, PoSCReply%02d.bt                  hvzk/PoSCTW.java:221-235
#     <nizkp>/proofs/KeepList%02d.bt                the positions kept for N  mixnet/PermutationCommitment.java:240-242, 413, 455
#     <nizkp>/proofs/CCPoSCommitment%02d.bt, CCPoSReply%02d.bt                hvzk/CCPoSW.java:274-288
#     <nizkp>/proofs/Ciphertexts%02d.bt             the output list
# Prover: PermutationCommitment.generate :251-366 + PoSCTW.prove :73-134; shrink :390-471; ShufflerElGamalSession committed
# shuffle :789-792 + CCPoSW.prove :75-158.  Verifier: readPermutationCommitment :618-636, verifyPoSC :652-703, shrinkPermComm
# :714-746, verifyCCPoS :757-830 of MixNetElGamalVerifyFiatShamirSession.java.
# =============================================================================================================================
def poscc_file(nizkp, l):
    return _p(nizkp, "proofs", "PoSCCommitment%02d.bt" % l)


def poscr_file(nizkp, l):
    return _p(nizkp, "proofs", "PoSCReply%02d.bt" % l)


def kl_file(nizkp, l):
    return _p(nizkp, "proofs", "KeepList%02d.bt" % l)


def ccposc_file(nizkp, l):
    return _p(nizkp, "proofs", "CCPoSCommitment%02d.bt" % l)


def ccposr_file(nizkp, l):
    return _p(nizkp, "proofs", "CCPoSReply%02d.bt" % l)


def _booleans_tree(flags) -> bytes:
    """A boolean array as a byte tree: one leaf, one byte per flag (ByteTree.booleanArrayToByteTree is VCR code, not in the
    reference tree: [NOT-IN-REF], restated from the verifier specification)."""
    return fs.leaf(bytes(1 if f else 0 for f in flags))


def _read_booleans(path: str, n: int):
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) != 5 + n or buf[0] != 1 or int.from_bytes(buf[1:5], "big") != n or any(b > 1 for b in buf[5:]):
        return None
    return [b == 1 for b in buf[5:]]


def _seed_and_challenge(chal, hn, head: bytes, parts, commit_fn, NV):
    """The two random-oracle calls around a commitment: seed = RO(rho || node(parts...)); v = RO(rho || node(leaf(seed),
    commitment)) as a positive integer.  `parts`: byte strings / file paths hashed in order after `head`."""
    bits = 8 * hashlib.new(hn).digest_size
    d = chal.start(bits)
    d.update(head)
    for part in parts:
        if isinstance(part, str):
            with open(part, "rb") as f:
                d.update(f.read())
        else:
            d.update(part)
    seed = chal.finish(d, bits)
    com_bt = commit_fn(seed)
    v = int.from_bytes(chal.challenge(fs._hdr(0, 2) + fs.leaf(seed) + com_bt, NV), "big")
    return seed, com_bt, v


def write_precomputation(nizkp: str, l: int, grp, params: dict, n_max: int, rand, H: Optional[PGroupElementArray] = None):
    """`vmn -precomp` of party l for N_0 = n_max ciphertexts: the permutation commitment u and its proof of a shuffle of
    commitments.  Returns the prover's secrets for the online phase: (pi, R, U, H)."""
    os.makedirs(_p(nizkp, "proofs"), exist_ok=True)
    NV, NE, NR = int(params["vbitlenro"]), int(params["ebitlenro"]), int(params["rbitlen"])
    hn = _hashname(params)
    rho = global_prefix(params)
    chal = fs.Challenger(rho, hn)
    if H is None:
        H = derive_generators(grp, params, rho, n_max)
    g = grp.g
    pi = rand.permutation(n_max)
    R = native.random_ring_array_native(grp, rand, n_max, NR)
    U = native.permutation_commitment_native(grp, g, H, R, pi)
    u_bt = U.toByteTree()
    with open(pc_file(nizkp, l), "wb") as f:                                     # PermutationCommitment.java:364
        f.write(u_bt)
    P = native.PoSCBasicTW(grp, NV, NE, NR, rand=rand)                           # PoSCTW.prove :73-134
    P.setInstance(g, H, U, R, pi)
    box = {}

    def commit(seed):
        P.setBatchVectorSeed(seed)
        box["com"] = P.commit()
        bt = box["com"].native.toByteTree()
        with open(poscc_file(nizkp, l), "wb") as f:
            f.write(bt)
        return bt
    _, _, v = _seed_and_challenge(chal, hn, fs._hdr(0, 3) + fs.leaf(grp.enc_el(g)), [H.toByteTree(), u_bt], commit, NV)
    rep = P.reply(v)
    with open(poscr_file(nizkp, l), "wb") as f:
        f.write(rep.native.toByteTree())
    box.clear()
    rep = None
    P.free()
    return pi, R, U, H


def write_committed_shuffle(nizkp: str, l: int, grp, params: dict, pkey: Sequence, W: Sequence[PGroupElementArray], rand,
                            pi, R, U, H) -> List[PGroupElementArray]:
    """The online phase for the N <= N_0 ciphertexts that arrived (W = the list in l_file(nizkp, l - 1)): shrink
    (PermutationCommitment.java:390-471, keep list to its file), re-encrypt and permute (ShufflerElGamalSession.java:789-792),
    CCPoSW.prove (:75-158).  Returns w'."""
    NV, NE, NR = int(params["vbitlenro"]), int(params["ebitlenro"]), int(params["rbitlen"])
    n, width = W[0].size(), len(W) // 2
    hn = _hashname(params)
    chal = fs.Challenger(global_prefix(params), hn)
    g = grp.g
    keep, pi_s = native.permutation_shrink_native(pi, n)
    with open(kl_file(nizkp, l), "wb") as f:
        f.write(_booleans_tree(keep))
    U_s, R_s, H_s = U.extract(keep), R.copyOfRange(0, n), H.copyOfRange(0, n)
    S = [native.random_ring_array_native(grp, rand, n, NR) for _ in range(width)]
    WP = native.reencrypt_native(grp, pkey, W, S, pi_s)
    write_ciphertexts(l_file(nizkp, l), WP)
    P = native.CCPoSBasicW(grp, NV, NE, NR, rand=rand)
    P.setInstance(g, H_s, U_s, pkey, W, WP, R_s, pi_s, S)
    box = {}

    def commit(seed):
        P.setBatchVectorSeed(seed)
        box["com"] = P.commit()
        bt = box["com"].native.toByteTree()
        with open(ccposc_file(nizkp, l), "wb") as f:
            f.write(bt)
        return bt
    _, _, v = _seed_and_challenge(chal, hn, fs._hdr(0, 6) + fs.leaf(grp.enc_el(g)),
                                  [H_s.toByteTree(), U_s.toByteTree(), fs.element_tree(grp, pkey), l_file(nizkp, l - 1), l_file(nizkp, l)],
                                  commit, NV)
    rep = P.reply(v)
    with open(ccposr_file(nizkp, l), "wb") as f:
        f.write(rep.native.toByteTree())
    box.clear()
    rep = None
    P.free()
    for a in S + [U_s, R_s, H_s]:
        a.free()
    return WP


def verify_precomputed_shuffle(nizkp: str, l: int, grp, params: dict, pkey: Sequence, n_max: int,
                               vectors: Optional[Dict[str, str]] = None) -> bool:
    """The standalone verifier's path for a precomputed shuffle of party l (MixNetElGamalVerifyFiatShamirSession.java
    :1395-1500): readPermutationCommitment(N_0), verifyPoSC, read the output, shrinkPermComm through the keep list, verifyCCPoS.
    `vectors` receives der.rho, PoSC.s, PoSC.v, CCPoS.s, CCPoS.v (the names the reference registers, Tool.java:155-175) and the
    verifiers' intermediates (PoSC.A/C/D, CCPoS.A, CCPoS.B: not printed by the reference; for diffing two builds of this code)."""
    tv = vectors if vectors is not None else {}
    NV, NE, NR = int(params["vbitlenro"]), int(params["ebitlenro"]), int(params["rbitlen"])
    width = len(pkey) // 2
    hn = _hashname(params)
    rho = global_prefix(params)
    tv["der.rho"] = rho.hex()
    chal = fs.Challenger(rho, hn)
    bits = 8 * hashlib.new(hn).digest_size
    g = grp.g
    H = derive_generators(grp, params, rho, n_max)
    with open(pc_file(nizkp, l), "rb") as f:                                     # readPermutationCommitment :618-636
        u_bt = f.read()
    try:
        U = grp.toElementArrayFromByteTree(u_bt, n_max)
        if not U.isMember():
            raise ValueError("u outside the group")
    except (ValueError, native.VmnError):
        return False
    # ---- verifyPoSC :652-703
    V = native.PoSCBasicTW(grp, NV, NE, NR)
    V.setInstance(g, H, U)
    seed = chal.challenge(fs._hdr(0, 3) + fs.leaf(grp.enc_el(g)) + H.toByteTree() + u_bt, bits)
    tv["PoSC.s"] = seed.hex()
    V.setBatchVectorSeed(seed)
    with open(poscc_file(nizkp, l), "rb") as f:
        com_bt = f.read()
    com = native.Message.fromByteTree(grp, com_bt, native.PoSCBasicTW._com_kinds, [n_max, 1, n_max, 1, 1])
    if com is None:
        return False
    V.setCommitment(com)
    v = int.from_bytes(chal.challenge(fs._hdr(0, 2) + fs.leaf(seed) + com_bt, NV), "big")
    tv["PoSC.v"] = format(v, "x")
    V.setChallenge(v)
    with open(poscr_file(nizkp, l), "rb") as f:
        rep = native.Message.fromByteTree(grp, f.read(), native.PoSCBasicTW._rep_kinds, [1, n_max, 1, 1, n_max])
    if rep is None:
        return False
    ok_posc = V.verify(rep)
    tv["PoSC.A"], tv["PoSC.C"], tv["PoSC.D"] = _hex(V.getA()), _hex(V.getC()), _hex(V.getD())
    com = rep = None
    V.free()
    if not ok_posc:            # (the reference then takes the generators for u and goes on; a harness stops with the verdict)
        return False
    # ---- the lists, the keep list, verifyCCPoS :757-830
    W = read_ciphertexts(grp, l_file(nizkp, l - 1), width)
    if W is None:
        return False
    n = W[0].size()
    WP = read_ciphertexts(grp, l_file(nizkp, l), width, n)
    if WP is None:
        return False
    keep = _read_booleans(kl_file(nizkp, l), n_max)
    if keep is None or sum(keep) != n:                                           # shrinkPermComm :714-746 fails the party
        return False
    U_s, H_s = U.extract(keep), H.copyOfRange(0, n)
    C = native.CCPoSBasicW(grp, NV, NE, NR)
    C.setInstance(g, H_s, U_s, pkey, W, WP)
    d = chal.start(bits)
    d.update(fs._hdr(0, 6) + fs.leaf(grp.enc_el(g)))
    d.update(H_s.toByteTree())
    d.update(U_s.toByteTree())
    d.update(fs.element_tree(grp, pkey))
    for path in (l_file(nizkp, l - 1), l_file(nizkp, l)):
        with open(path, "rb") as f:
            d.update(f.read())
    seed2 = chal.finish(d, bits)
    tv["CCPoS.s"] = seed2.hex()
    C.setBatchVectorSeed(seed2)
    C.computeAB()
    A, B = C.getAB()
    tv["CCPoS.A"], tv["CCPoS.B"] = _hex(A), "(" + ", ".join(_hex(x) for x in B) + ")"
    with open(ccposc_file(nizkp, l), "rb") as f:
        com_bt = f.read()
    com = native.Message.fromByteTree(grp, com_bt, native.CCPoSBasicW._com_kinds, [1, 2 * width])
    if com is None:
        return False
    C.setCommitment(com)
    v2 = int.from_bytes(chal.challenge(fs._hdr(0, 2) + fs.leaf(seed2) + com_bt, NV), "big")
    tv["CCPoS.v"] = format(v2, "x")
    C.setChallenge(v2)
    with open(ccposr_file(nizkp, l), "rb") as f:
        rep = native.Message.fromByteTree(grp, f.read(), native.CCPoSBasicW._rep_kinds, [1, width, n])
    if rep is None:
        return False
    verdict = C.verify(rep)
    com = rep = None
    C.free()
    for a in W + WP + [H, U, U_s, H_s]:
        a.free()
    return bool(verdict)
