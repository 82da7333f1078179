#!/usr/bin/env python3
"""vmnv_vectors.py — verify the shuffle of party l in a proof directory on the GPU and print the test vectors the reference's
standalone verifier prints under `vmnv -t` (mixnet/MixNetElGamalVerifyFiatShamirSession.java:843-937; names and descriptions:
MixNetElGamalVerifyFiatShamirTool.java:82-223; output format `\\nTEST VECTOR\\n<name> - <description>\\n<value>`,
MixNetElGamalVerifyFiatShamir.java:382-388).

    python tools/vmnv_vectors.py <nizkp dir> [-l PARTY] [-t der.rho,PoS] [--arrays]
    python tools/vmnv_vectors.py --demo <new dir> [-n 100]      write a directory with the C++ prover first (synthetic list)

The directory layout is the reference's (verificatum-vmn_amd/proofdir.py); what `vmnv` reads from the protocol-info XML is
in <dir>/params.json.  THE diff that would pin parity: on a machine with a JDK + VCR, run `vmnv -t der.rho,PoS ...` on a proof
directory the Java mix-net wrote, run this tool on the same directory, and compare the values name by name (VCR's
toString() of group elements is not part of the reference tree; this tool prints hexadecimal)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DESCRIPTIONS = {                      # MixNetElGamalVerifyFiatShamirTool.java:82-223
    "der.rho": "Derived prefix bytes to all random oracle queries.", "bas.h": "Independent generators.",
    "PoS.s": "PoS. Seed to derive batching vector in hexadecimal notation.",
    "PoS.v": "PoS. Integer challenge in hexadecimal notation.", "PoS.A": "PoS. Batched permutation commitment.",
    "PoS.F": "PoS. Batched input ciphertexts.", "PoS.B": "PoS. Commitment components.",
    "PoS.C": "PoS. Derived intermediate values.", "PoS.D": "PoS. Derived intermediate values.",
    "PoS.Ap": "PoS. Commitment components.", "PoS.Bp": "PoS. Commitment components.", "PoS.Cp": "PoS. Commitment components.",
    "PoS.Dp": "PoS. Commitment components.", "PoS.Fp": "PoS. Commitment components.", "PoS.k_A": "PoS. Reply components.",
    "PoS.k_B": "PoS. Reply components.", "PoS.k_C": "PoS. Reply components.", "PoS.k_D": "PoS. Reply components.",
    "PoS.k_E": "PoS. Reply components.", "PoS.k_F": "PoS. Reply components.",
    "PoSC.s": "PoSC. Seed to derive batching vector in hexadecimal notation.",
    "PoSC.v": "PoSC. Integer challenge in hexadecimal notation.",
    "CCPoS.s": "CCPoS. Seed to derive batching vector in hexadecimal notation.",
    "CCPoS.v": "CCPoS. Integer challenge in hexadecimal notation.",
    # not registered by the reference (private fields of its classes): printed for diffing two builds of THIS code
    "PoSC.A": "[not in the reference] PoSC. Batched permutation commitment.", "PoSC.C": "[not in the reference] PoSC. Derived intermediate values.",
    "PoSC.D": "[not in the reference] PoSC. Derived intermediate values.", "CCPoS.A": "[not in the reference] CCPoS. Batched permutation commitment.",
    "CCPoS.B": "[not in the reference] CCPoS. Batched input ciphertexts."}
ORDER = ["der.rho", "bas.h", "PoSC.s", "PoSC.v", "PoSC.A", "PoSC.C", "PoSC.D", "CCPoS.s", "CCPoS.A", "CCPoS.B", "CCPoS.v",
         "PoS.s", "PoS.A", "PoS.F", "PoS.B", "PoS.Ap", "PoS.Bp", "PoS.Cp", "PoS.Dp", "PoS.Fp", "PoS.v", "PoS.C",
         "PoS.D", "PoS.k_A", "PoS.k_B", "PoS.k_C", "PoS.k_D", "PoS.k_E", "PoS.k_F"]


def selected(name, wanted):
    """checkTestVector (MixNetElGamalVerifyFiatShamir.java:397-409): the name itself or its prefix before the dot."""
    return name in wanted or name.split(".")[0] in wanted


def group_of(vmn, ctx, params):
    if params["group"]["kind"] == "modp":
        g = params["group"]
        return vmn.ModPGroup(ctx, int(g["p"], 16), int(g["q"], 16), int(g["g"], 16))
    return vmn.ECqPGroup(ctx, params["group"]["curve"], java_widths=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("nizkp")
    ap.add_argument("-l", type=int, default=1)
    ap.add_argument("-t", default="der,PoS", help="comma-separated test vector names (a prefix selects its group)")
    ap.add_argument("--arrays", action="store_true", help="also the N-sized vectors (bas.h, PoS.B, PoS.Bp, PoS.k_B, PoS.k_E)")
    ap.add_argument("--demo", action="store_true", help="write the directory first: synthetic list, C++ prover")
    ap.add_argument("-n", type=int, default=100)
    ap.add_argument("--precomputed", type=int, default=0, metavar="N_0",
                    help="the directory holds a PRECOMPUTED shuffle for N_0 ciphertexts (PoSC + keep list + CCPoS files) instead of a PoS; "
                         "with --demo: write one (N_0 >= n)")
    ap.add_argument("--bits", type=int, default=2048)
    args = ap.parse_args()
    import json
    import __graft_entry__ as entry
    vmn = entry.load_package()
    from verificatum_vmn_amd import proofdir, randomsource, stdgroups
    ctx = vmn.Context(0)
    if args.demo:
        p, q, g = stdgroups.modp_group(args.bits)
        params = {"version": "3.1.0", "sid": "MyDemo", "auxsid": "default", "rbitlen": 100, "vbitlenro": 256, "ebitlenro": 256,
                  "prg": "SHA-256", "rohash": "SHA-256", "rohash_name": "SHA-256", "width": 1,
                  "pgroup": f"ModPGroup(safe-prime modulus=2*order+1. order bit-length = {args.bits - 1})",
                  "group": {"kind": "modp", "p": format(p, "x"), "q": format(q, "x"), "g": format(g, "x")}}
        grp = group_of(vmn, ctx, params)
        rnd = randomsource.InsecureShaRandomSource(b"vmnv-demo", q)
        y = pow(g, rnd.ring_element(), p)
        pkey = [g, y]
        T = grp.ringArray(rnd.ring_array(args.n))
        M = grp.exp(g, grp.ringArray(rnd.ring_array(args.n)))
        W = [grp.exp(g, T), M.mul(grp.exp(y, T))]
        params["N_0"] = args.precomputed
        proofdir.write_inputs(args.nizkp, grp, params, pkey, W)
        prover = randomsource.SecureRandomSource(q)
        if args.precomputed:
            pi, R, U, H = proofdir.write_precomputation(args.nizkp, args.l, grp, params, args.precomputed, prover)
            proofdir.write_committed_shuffle(args.nizkp, args.l, grp, params, pkey, W, prover, pi, R, U, H)
        else:
            proofdir.write_shuffle(args.nizkp, args.l, grp, params, pkey, W, prover)
        print(f"wrote {args.nizkp}: {args.n} ciphertexts, party {args.l}", file=sys.stderr)
    with open(os.path.join(args.nizkp, proofdir.PARAMS)) as f:
        params = json.load(f)
    grp = group_of(vmn, ctx, params)
    from verificatum_vmn_amd import eio
    with open(proofdir.pk_file(args.nizkp), "rb") as f:
        tree, _ = eio.decode(f.read())
    flat = [leaf for part in tree for leaf in (part if isinstance(part, list) else [part])]
    pkey = [grp.dec_el(b) for b in flat]
    vectors = {}
    n0 = args.precomputed or int(params.get("N_0", 0))
    if n0:
        verdict = proofdir.verify_precomputed_shuffle(args.nizkp, args.l, grp, params, pkey, n0, vectors)
        if args.t == "der,PoS":
            args.t = "der,PoSC,CCPoS"
    else:
        verdict = proofdir.verify_shuffle(args.nizkp, args.l, grp, params, pkey, vectors, with_arrays=args.arrays)
    wanted = set(args.t.split(","))
    for name in ORDER:
        if name in vectors and selected(name, wanted):
            print(f"\nTEST VECTOR\n{name} - {DESCRIPTIONS[name]}\n{vectors[name]}")
    print(f"\nverdict of party {args.l}: {'accepted' if verdict else 'REJECTED'}" +
          (f"   verdicts(A,B,C,D,F) = {vectors['verdicts(A,B,C,D,F)']}" if "verdicts(A,B,C,D,F)" in vectors else ""))
    sys.exit(0 if verdict else 1)


if __name__ == "__main__":
    main()
