"""Worker of the multi-process tests: runs the sharded proof of a shuffle on this rank and, on rank 0,
checks the gathered transcript against the single-process oracle.  Backend "fake" = integer arrays on
the CPU (gloo) under the Python mirror of the sharded drivers; backend "hip..." = the real library on this rank's GPU
under the sharded C++ drivers (vmn_*_set_comm), or -- with "-mirror" -- under the Python mirror.

argv: backend bits n width out_path [flow]      flow = pos (default) | ccpos | posc | pos-seeded | ccpos-seeded
("-seeded": the prover's N-sized draws, the shuffler's exponents and the batching vector are 32-byte seeds -- each rank
generates only the rows it reads, vmn_rarray_from_prg_range / _gather -- against the oracle run on the expanded arrays)"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch
import torch.distributed as dist

import __graft_entry__ as entry
from conftest import load_golden
from oracle import pyref, pyref_proofs as P
from tape import SeedTape, Tape


def load_parallel():
    """The product's Comm / shard_bounds together with the test-only Python mirror of the sharded drivers."""
    import mirror
    return mirror.load(entry, ("parallel_mirror",))["parallel_mirror"]


def load_native():
    import mirror
    return mirror.load(entry, ("native",))["native"]


def gather_and_check(dist, rank, world, mine, expect, out_path):
    """All ranks' shards to rank 0; compare with the oracle's transcript."""
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        checks = {}
        for key, want in expect["arrays"].items():
            checks[key] = [x for gsh in gathered for x in gsh["arrays"][key]] == want
        checks["scalars"] = all(gsh["scalars"] == expect["scalars"] for gsh in gathered)
        for key, want in expect["flags"].items():
            checks[key] = all(gsh["flags"][key] == want for gsh in gathered)
        with open(out_path, "w") as f:
            json.dump({"pass": all(checks.values()), "why": json.dumps(checks), "world": world, "comm": comm_state(),
                       "exchanges": [gsh.get("exchanges") for gsh in gathered]}, f)
    dist.barrier()
    raise CaseDone(0)


_COMM = []


def comm_state():
    """Which transport the case's communicator really used (rank 0's view)."""
    c = _COMM[-1] if _COMM else None
    return {"backend_used": getattr(c, "backend_used", None), "fell_back": getattr(c, "fell_back", None),
            "torch_backend": dist.get_backend() if dist.is_initialized() else None}


class CaseDone(Exception):
    """A case has written its result: the launch goes on with the next one (several cases share one process group -- starting
    the ranks costs more than most cases)."""

    def __init__(self, code=0):
        Exception.__init__(self, code)
        self.code = code


def flow_ccpos_or_posc(flow, native, par, nat, comm, dist, rank, world, G, K, g, h, pkey, w, t, q, n, width, bits3, out_path):
    """CCPoS (plain + raised) or PoSC on a permutation commitment, sharded; transcript vs the oracle."""
    NV, NE, NR = bits3
    pi, r, s = t.permutation(n), t.ring_array(n), [t.ring_array(n) for _ in range(width)]
    e, v, rho = t.int_array(n, NE), t.int_array(1, NV)[0], t.int_array(1, 50)[0]
    u = P.g_permutation_commitment(K, g, h, r, pi)
    wp = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi)
    lo, hi = par.shard_bounds(n, world, rank)
    H, W = G.toElementArray(h), [G.toElementArray(c) for c in w]
    R, S = G.ringArray(r), [G.ringArray(c) for c in s]
    ints = lambda a: a.toInts() if hasattr(a, "toInts") else a
    if native:
        ncomm = nat.NativeComm(comm)
        U = nat.permutation_commitment_shard_native(G, g, H, R, pi, lo, hi)          # this rank's shard, from the whole h and r
        WP = nat.reencrypt_shard_native(G, pkey, W, S, pi, lo, hi)
    else:
        ncomm = None
        U = G.toElementArray(u[lo:hi])
        WP = [G.toElementArray(c[lo:hi]) for c in wp]
    arrays = {"u": ints(U)}
    for c in range(2 * width):
        arrays["wp%d" % c] = ints(WP[c])
    expect = {"arrays": {"u": u}, "flags": {}}
    for c in range(2 * width):
        expect["arrays"]["wp%d" % c] = wp[c]
    if flow == "posc":
        if K.__class__.__name__ != "ModPAdapter":
            raise SystemExit("the PoSC oracle is integer-only")
        o = P.PoSC(K.p, q, NV, NE, NR, rand=Tape(b"prover", q))
        o.setInstance(g, h, u, r, pi)
        o.setBatchVector(e)
        com_o, rep_o = o.commit(), o.reply(v)
        pr = nat.PoSCBasicTW(G, NV, NE, NR, rand=Tape(b"prover", q))
        pr.setComm(ncomm)
        pr.setInstance(g, H, U, R, pi)
        pr.setBatchVector(e)
        com, rep = pr.commit(), pr.reply(v)
        ver = nat.PoSCBasicTW(G, NV, NE, NR)
        ver.setComm(ncomm)
        ver.setInstance(g, H, U)
        ver.setBatchVector(e)
        ver.setCommitment(com)
        ver.setChallenge(v)
        ok = ver.verify(rep)
        bad = dict(rep)
        if rank == world - 1 and hi > lo:
            kb = rep["k_B"].toInts()
            kb[-1] = (kb[-1] + 1) % q
            bad["k_B"] = G.ringArray(kb)
        bad_ok = ver.verify(bad)
        arrays.update({"B": ints(com["B"]), "Bp": ints(com["Bp"]), "k_B": ints(rep["k_B"]), "k_E": ints(rep["k_E"])})
        expect["arrays"].update({"B": com_o["B"], "Bp": com_o["Bp"], "k_B": rep_o["k_B"], "k_E": rep_o["k_E"]})
        scal = [com["Ap"], com["Cp"], com["Dp"], rep["k_A"], rep["k_C"], rep["k_D"]]
        expect["scalars"] = [com_o["Ap"], com_o["Cp"], com_o["Dp"], rep_o["k_A"], rep_o["k_C"], rep_o["k_D"]]
        expect["flags"] = {"ok": True, "bad_ok": False}
        mine = {"arrays": arrays, "scalars": scal, "flags": {"ok": ok, "bad_ok": bad_ok}, "exchanges": ncomm.exchanges}
        gather_and_check(dist, rank, world, mine, expect, out_path)
    oc = P.GCCPoS(K, NV, NE, NR, rand=Tape(b"prover", q))
    oc.setInstance(g, h, u, pkey, w, wp, r, pi, s)
    oc.setBatchVector(e)
    com_o, rep_o = oc.commit(), oc.reply(v)
    if native:
        pr = nat.CCPoSBasicW(G, NV, NE, NR, rand=Tape(b"prover", q))
        pr.setComm(ncomm)
        mk = lambda: nat.CCPoSBasicW(G, NV, NE, NR)
    else:
        pr = par.ShardedCCPoSBasicW(G, NV, NE, NR, comm, rand=Tape(b"prover", q))
        mk = lambda: par.ShardedCCPoSBasicW(G, NV, NE, NR, comm)
    pr.setInstance(g, H, U, pkey, W, WP, R, pi, S)
    pr.setBatchVector(e)
    com, rep = pr.commit(), pr.reply(v)
    flags = {}
    RU_full = G.toElementArray(K.exp_scalar(u, rho))              # raised arrays handed over whole: the drivers cut their shard
    RH_full = G.toElementArray(K.exp_scalar(h, rho))
    for raised in (False, True):
        ver = mk()
        if native:
            ver.setComm(ncomm)
        ver.setInstance(g, H, U, pkey, W, WP)
        ver.setBatchVector(e)
        ver.setCommitment(com)
        ver.setChallenge(v)
        bad = dict(rep)
        if rank == 0 and hi > lo:
            ke = ints(rep["k_E"])
            ke[0] = (ke[0] + 1) % q
            bad["k_E"] = G.ringArray(ke)
        if raised:
            ver.computeAB(RU_full)
            flags["ok_raised"] = ver.verify(rep, RH_full, rho)
            flags["bad_raised"] = ver.verify(bad, RH_full, rho)
        else:
            ver.computeAB()
            flags["ok_plain"] = ver.verify(rep)
            flags["bad_plain"] = ver.verify(bad)
    arrays["k_E"] = ints(rep["k_E"])
    expect["arrays"]["k_E"] = rep_o["k_E"]
    expect["scalars"] = [com_o["Ap"], com_o["Bp"], rep_o["k_A"], rep_o["k_B"]]
    expect["flags"] = {"ok_plain": True, "bad_plain": False, "ok_raised": True, "bad_raised": False}
    mine = {"arrays": arrays, "scalars": [com["Ap"], com["Bp"], rep["k_A"], rep["k_B"]], "flags": flags,
            "exchanges": ncomm.exchanges if ncomm else None}
    gather_and_check(dist, rank, world, mine, expect, out_path)


def flow_seeded(flow, par, nat, comm, dist, rank, world, G, K, g, h, pkey, w, q, n, width, bits3, out_path):
    """PoS or CCPoS through the sharded C++ drivers with every N-sized random array a PRG draw: what bench.py's
    multi-GPU legs run.  The oracle expands the same seeds in full (SeedTape(expanding=True))."""
    from oracle import pyref_prg
    NV, NE, NR = bits3
    ncomm = nat.NativeComm(comm)
    lo, hi = par.shard_bounds(n, world, rank)
    H, W = G.toElementArray(h), [G.toElementArray(c) for c in w]
    ints = lambda a: a.toInts() if hasattr(a, "toInts") else a
    small = Tape(b"seeded-small", q)
    pi, v = small.permutation(n), small.int_array(1, NV)[0]
    e_seed = pyref_prg.random_oracle(b"seeded-e", 256)
    e = pyref_prg.random_integers(e_seed, n, NE)
    shuffler_o = SeedTape(b"shuffler", q, NR, expanding=True)
    if flow == "pos-seeded":
        s = [shuffler_o.ring_array(n) for _ in range(width)]
        o = P.GPoS(K, NV, NE, NR, rand=SeedTape(b"prover", q, NR, expanding=True))
        o.precompute(g, h, pi)
        wp_o = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi)
        o.setInstance(pkey, w, wp_o, s)
        o.setBatchVector(e)
        com_o, rep_o = o.commit(), o.reply(v)
        pr = nat.PoSBasicTW(G, NV, NE, NR, rand=SeedTape(b"prover", q, NR))
        pr.setComm(ncomm)
        pr.precompute(g, H, pi)
        WP, S = nat.reencrypt_shard_seeded_native(G, pkey, W, SeedTape(b"shuffler", q, NR), NR, pi, lo, hi)
        pr.setInstance(pkey, W, WP, S)
        pr.setBatchVectorSeed(e_seed)
        com, rep = pr.commit(), pr.reply(v)
        ver = nat.PoSBasicTW(G, NV, NE, NR)
        ver.setComm(ncomm)
        ver.precompute(g, H)
        ver.setPermutationCommitment(pr.u)
        ver.setInstance(pkey, W, WP)
        ver.setBatchVectorSeed(e_seed)
        ver.computeAF()
        ver.setCommitment(com)
        ver.setChallenge(v)
        ok = ver.verify(rep)
        arrays = {"u": ints(pr.u), "B": ints(com["B"]), "Bp": ints(com["Bp"]), "k_B": ints(rep["k_B"]), "k_E": ints(rep["k_E"])}
        expect = {"arrays": {"u": o.u, "B": com_o["B"], "Bp": com_o["Bp"], "k_B": rep_o["k_B"], "k_E": rep_o["k_E"]},
                  "scalars": [com_o["Ap"], com_o["Cp"], com_o["Dp"], com_o["Fp"], rep_o["k_A"], rep_o["k_C"], rep_o["k_D"], rep_o["k_F"]],
                  "flags": {"ok": True}}
        for c in range(2 * width):
            arrays["wp%d" % c] = ints(WP[c])
            expect["arrays"]["wp%d" % c] = wp_o[c]
        for c in range(width):
            arrays["s%d" % c] = ints(S[c])
            expect["arrays"]["s%d" % c] = s[c]
        mine = {"arrays": arrays, "scalars": [com["Ap"], com["Cp"], com["Dp"], com["Fp"], rep["k_A"], rep["k_C"], rep["k_D"], rep["k_F"]],
                "flags": {"ok": ok}, "exchanges": ncomm.exchanges}
        gather_and_check(dist, rank, world, mine, expect, out_path)
    # ccpos-seeded: permutation commitment and re-encryption from seeds, then the commitment-consistent proof
    r = shuffler_o.ring_array(n)
    s = [shuffler_o.ring_array(n) for _ in range(width)]
    u = P.g_permutation_commitment(K, g, h, r, pi)
    wp = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi)
    oc = P.GCCPoS(K, NV, NE, NR, rand=SeedTape(b"prover", q, NR, expanding=True))
    oc.setInstance(g, h, u, pkey, w, wp, r, pi, s)
    oc.setBatchVector(e)
    com_o, rep_o = oc.commit(), oc.reply(v)
    shuffler = SeedTape(b"shuffler", q, NR)
    U, R = nat.permutation_commitment_shard_seeded_native(G, g, H, shuffler, NR, pi, lo, hi)
    WP, S = nat.reencrypt_shard_seeded_native(G, pkey, W, shuffler, NR, pi, lo, hi)
    pr = nat.CCPoSBasicW(G, NV, NE, NR, rand=SeedTape(b"prover", q, NR))
    pr.setComm(ncomm)
    pr.setInstance(g, H, U, pkey, W, WP, R, pi, S)
    pr.setBatchVectorSeed(e_seed)
    com, rep = pr.commit(), pr.reply(v)
    ver = nat.CCPoSBasicW(G, NV, NE, NR)
    ver.setComm(ncomm)
    ver.setInstance(g, H, U, pkey, W, WP)
    ver.setBatchVectorSeed(e_seed)
    ver.setCommitment(com)
    ver.setChallenge(v)
    ver.computeAB()
    ok = ver.verify(rep)
    arrays = {"u": ints(U), "r": ints(R), "k_E": ints(rep["k_E"])}
    expect = {"arrays": {"u": u, "r": r, "k_E": rep_o["k_E"]}, "scalars": [com_o["Ap"], com_o["Bp"], rep_o["k_A"], rep_o["k_B"]],
              "flags": {"ok": True}}
    for c in range(2 * width):
        arrays["wp%d" % c] = ints(WP[c])
        expect["arrays"]["wp%d" % c] = wp[c]
    mine = {"arrays": arrays, "scalars": [com["Ap"], com["Bp"], rep["k_A"], rep["k_B"]], "flags": {"ok": ok}, "exchanges": ncomm.exchanges}
    gather_and_check(dist, rank, world, mine, expect, out_path)


def main():
    """argv: backend bits n width out_path [flow], or `--cases FILE` (a JSON list of such argument lists: the cases of one
    launch, run one after the other on the same process group; all of them over gloo or all over RCCL)."""
    if sys.argv[1] == "--cases":
        cases = json.load(open(sys.argv[2]))
    else:
        cases = [sys.argv[1:]]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("VMN_COMBINED_MIN", "1")     # the sharded verifiers take check (B) in its combined form (with the carried-in B) too
    device = None
    first_backend = cases[0][0][: -len("-mirror")] if cases[0][0].endswith("-mirror") else cases[0][0]
    if first_backend in ("hip", "hip-ec"):     # one GPU per rank, RCCL
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=device)
    else:                                      # "fake": CPU arrays; "hip-gloo": real kernels, all ranks share GPU 0
        dist.init_process_group("gloo")
    code = 0
    for case in cases:
        try:
            run_case(device, *case)
        except CaseDone as done:
            code = code or done.code
        if code:
            break
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(code)


def run_case(device, backend, bits, n, width, out_path, flow="pos"):
    bits, n, width = int(bits), int(n), int(width)
    mirror = backend.endswith("-mirror")
    if mirror:
        backend = backend[: -len("-mirror")]
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    par = load_parallel()
    if flow == "comm-fallback":
        # parallel.Comm told to use a device path that cannot work here (no GPU in the CPU suite): the first exchange must fall
        # back to the gloo group, say so, and keep returning the right blocks
        c = par.Comm(dist, torch.device("cuda", 0), fallback=dist.group.WORLD)
        got = c.all_gather_bytes(bytes([rank]) * 5)
        again = c.all_gather_ints([rank + 7, 1 << 70], 16)
        ok = (got == [bytes([k]) * 5 for k in range(world)] and again == [[k + 7, 1 << 70] for k in range(world)]
              and c.fell_back is not None and c.backend_used == "gloo (fallback)" and c.max_over_ranks(float(rank)) == float(world - 1)
              and c.all_true(True) and not c.all_true(rank != 0))
        if rank == 0:
            with open(out_path, "w") as f:
                json.dump({"pass": bool(ok), "why": f"fell_back={c.fell_back!r} backend={c.backend_used}", "world": world, "exchanges": []}, f)
        dist.barrier()
        raise CaseDone(0 if ok else 1)
    # on the device path (RCCL) the communicator gets a gloo group as its safety net, as in bench.py: the constructor's
    # probe exchange + the ranks' collective decision run here too
    comm = par.Comm(dist, device, fallback=dist.new_group(backend="gloo") if device is not None else None)
    _COMM.append(comm)
    ec = backend.endswith("-ec")
    native = backend.startswith("hip") and not mirror
    if backend.startswith("hip"):
        import fast_pyref
        fast_pyref.install()                   # GPU box: the oracle's array exponentiations through GMP (see the module)
    nat = load_native() if native else None
    if ec:
        from oracle.pyref_ec import Curve
        curve = Curve("P-256")
        K = P.ECAdapter(curve)
        p, q, g = curve.p, curve.n, curve.g
    else:
        grp, _ = load_golden(bits)
        p, q, g = grp["p"], grp["q"], grp["g"]
        K = P.ModPAdapter(p, q)
    NV, NE, NR = (100, 100, 50) if bits == 512 else (256, 256, 100)
    # public instance + secrets, identical on every rank (deterministic tape)
    t = Tape(b"dist%d" % bits, q)
    h = K.exp_fixed(g, t.ring_array(n))
    y = K.exp(g, t.ring_element())
    pkey = [g] * width + [y] * width
    msgs = [K.exp_fixed(g, t.ring_array(n)) for _ in range(width)]
    enc_r = [t.ring_array(n) for _ in range(width)]
    w = [K.exp_fixed(g, enc_r[c]) for c in range(width)] + \
        [K.mul_arrays(msgs[c], K.exp_fixed(y, enc_r[c])) for c in range(width)]
    pi = t.permutation(n)
    s = [t.ring_array(n) for _ in range(width)]
    e = t.int_array(n, NE)
    v = t.int_array(1, NV)[0]

    if backend.startswith("hip"):
        vmn = entry.load_package()
        ctx = vmn.Context(int(os.environ.get("LOCAL_RANK", "0")) if backend in ("hip", "hip-ec") else 0)
        G = vmn.ECqPGroup(ctx, "P-256") if ec else vmn.ModPGroup(ctx, p, q, g)
    elif ec:
        from fake_backend import FakeECGroup
        G = FakeECGroup(curve)
    else:
        from fake_backend import FakeGroup
        G = FakeGroup(p, q, g)
    if flow.endswith("-seeded"):
        assert native, "seeded flows run the C++ drivers"
        flow_seeded(flow, par, nat, comm, dist, rank, world, G, K, g, h, pkey, w, q, n, width, (NV, NE, NR), out_path)
    if flow != "pos":
        t2 = Tape(b"dist-%s%d" % (flow.encode(), bits), q)
        flow_ccpos_or_posc(flow, native, par, nat, comm, dist, rank, world, G, K, g, h, pkey, w, t2, q, n, width, (NV, NE, NR), out_path)
    H = G.toElementArray(h)
    W = [G.toElementArray(c) for c in w]

    if native:
        ncomm = nat.NativeComm(comm)
        lo, hi = par.shard_bounds(n, world, rank)
        assert (lo, hi) == nat.shard_bounds_native(n, world, rank)
        S = [G.ringArray(c) for c in s]
        pr = nat.PoSBasicTW(G, NV, NE, NR, rand=Tape(b"prover", q))
        pr.setComm(ncomm)
        pr.precompute(g, H, pi)
        WP = nat.reencrypt_shard_native(G, pkey, W, S, pi, lo, hi)
        pr.setInstance(pkey, W, WP, S)                 # w, s: whole arrays, w': this rank's shard
        ver = nat.PoSBasicTW(G, NV, NE, NR)
        ver.setComm(ncomm)
    else:
        pr = par.ShardedPoSBasicTW(G, NV, NE, NR, comm, rand=Tape(b"prover", q))
        pr.precompute(g, H, pi)
        WP = pr.reencrypt(pkey, W, s)
        pr.setInstance(pkey, W, WP)
        ver = par.ShardedPoSBasicTW(G, NV, NE, NR, comm)
    pr.setBatchVector(e)
    com = pr.commit()
    rep = pr.reply(v)
    ver.precompute(g, H)
    ver.setPermutationCommitment(pr.u)
    ver.setInstance(pkey, W, WP)
    ver.setBatchVector(e)
    ver.computeAF()
    ver.setCommitment(com)
    ver.setChallenge(v)
    ok = ver.verify(rep)
    # tampered reply: only this rank's shard of k_B is disturbed on the last rank
    bad = dict(rep)
    tamper_rank = max(k for k in range(world) if par.shard_bounds(n, world, k)[1] > par.shard_bounds(n, world, k)[0])
    if rank == tamper_rank:
        kb = rep["k_B"].toInts()
        kb[0] = (kb[0] + 1) % q
        bad["k_B"] = G.ringArray(kb)
    bad_ok = ver.verify(bad)
    bad_verdicts = ver.verdicts

    # gather the sharded transcript on rank 0 and compare with the single-process oracle
    ints = lambda a: a.toInts()
    mine = {"u": ints(pr.u), "wp": [ints(c) for c in WP], "B": ints(com["B"]), "Bp": ints(com["Bp"]),
            "k_B": ints(rep["k_B"]), "k_E": ints(rep["k_E"]),
            "scalars": [com["Ap"], com["Cp"], com["Dp"], com["Fp"], rep["k_A"], rep["k_C"], rep["k_D"], rep["k_F"]],
            "ok": ok, "bad_ok": bad_ok, "bad_verdicts": list(bad_verdicts), "exchanges": ncomm.exchanges if native else None}
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    result = {"pass": True, "why": ""}
    if rank == 0:
        o = P.GPoS(K, NV, NE, NR, rand=Tape(b"prover", q))
        o.precompute(g, h, pi)
        wp_o = P.g_reencrypt(K, w, P.g_reenc_factors(K, pkey, s), pi)
        o.setInstance(pkey, w, wp_o, s)
        o.setBatchVector(e)
        com_o = o.commit()
        rep_o = o.reply(v)
        cat = lambda key: [x for gsh in gathered for x in gsh[key]]
        checks = {
            "u": cat("u") == o.u,
            "wp": [[x for gsh in gathered for x in gsh["wp"][c]] for c in range(2 * width)] == wp_o,
            "B": cat("B") == com_o["B"], "Bp": cat("Bp") == com_o["Bp"],
            "k_B": cat("k_B") == rep_o["k_B"], "k_E": cat("k_E") == rep_o["k_E"],
            "scalars": all(gsh["scalars"] == [com_o["Ap"], com_o["Cp"], com_o["Dp"], com_o["Fp"], rep_o["k_A"],
                                              rep_o["k_C"], rep_o["k_D"], rep_o["k_F"]] for gsh in gathered),
            "accept": all(gsh["ok"] for gsh in gathered),
            "reject_tampered": all(not gsh["bad_ok"] and gsh["bad_verdicts"] == [True, False, True, True, True] for gsh in gathered),
        }
        result = {"pass": all(checks.values()), "why": json.dumps(checks), "world": world, "comm": comm_state(),
                  "exchanges": [gsh.get("exchanges") for gsh in gathered]}
        with open(out_path, "w") as f:
            json.dump(result, f)
    dist.barrier()
    raise CaseDone(0)


if __name__ == "__main__":
    main()
