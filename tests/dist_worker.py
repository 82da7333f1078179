"""Worker of the multi-process tests: runs the sharded proof of a shuffle on this rank and, on rank 0,
checks the gathered transcript against the single-process oracle.  Backend "fake" = integer arrays on
the CPU (gloo); backend "hip" = the real library on this rank's GPU (nccl)."""
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
from tape import Tape


def load_parallel():
    spec = importlib.util.spec_from_file_location("verificatum_vmn_amd.parallel", os.path.join(entry.PKG_DIR, "parallel.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = m
    spec.loader.exec_module(m)
    return m


def main():
    backend, bits, n, width, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    device = None
    if backend in ("hip", "hip-ec"):           # one GPU per rank, RCCL
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=device)
    else:                                      # "fake": CPU arrays; "hip-gloo": real kernels, all ranks share GPU 0
        dist.init_process_group("gloo")
    par = load_parallel()
    comm = par.Comm(dist, device)
    ec = backend.endswith("-ec")
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
    H = G.toElementArray(h)
    W = [G.toElementArray(c) for c in w]

    pr = par.ShardedPoSBasicTW(G, NV, NE, NR, comm, rand=Tape(b"prover", q))
    pr.precompute(g, H, pi)
    WP = pr.reencrypt(pkey, W, s)
    pr.setInstance(pkey, W, WP)
    pr.setBatchVector(e)
    com = pr.commit()
    rep = pr.reply(v)
    ver = par.ShardedPoSBasicTW(G, NV, NE, NR, comm)
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
            "ok": ok, "bad_ok": bad_ok, "bad_verdicts": list(bad_verdicts)}
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
        result = {"pass": all(checks.values()), "why": json.dumps(checks), "world": world}
        with open(out_path, "w") as f:
            json.dump(result, f)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)


if __name__ == "__main__":
    main()
