"""Repeated P-256 re-encryption passes (the A0 lines) with per-pass wall and kernel-family time, to look for
intermittent stalls.  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
import bench

vmn = entry.load_package()
ctx = vmn.Context(0)
nat, mx = bench.load_sub(entry, "native"), bench.load_sub(entry, "randomsource")
n, width = 400_000, 3
for rep in range(3):
    grp = vmn.ECqPGroup(ctx, "P-256")
    g, q = grp.g, grp.q
    bulk = mx.InsecureBulkRandomSource(100 + rep, q, grp.exp_bytes)
    y = grp.k_exp(g, bulk.ring_element())
    pkey = [g] * width + [y] * width
    W = [grp.exp(g if c < width else y, grp.ringArray(bulk.ring_array(n))) for c in range(2 * width)]
    for it in range(4):
        pin = it % 2 == 0
        blocks = [bulk.ring_array(n) for _ in range(width)]
        if pin:
            blocks = [torch.frombuffer(bytearray(b), dtype=torch.uint8).pin_memory() for b in blocks]
        pi = bulk.permutation(n)
        ctx.timing_reset(); ctx.timing_enable(True); ctx.synchronize()
        t0 = time.perf_counter()
        S = [grp.ringArray(b) for b in blocks]
        ctx.synchronize(); t1 = time.perf_counter()
        WP = nat.reencrypt_native(grp, pkey, W, S, pi)
        ctx.synchronize(); t2 = time.perf_counter()
        ctx.timing_enable(False)
        fam = {k: round(v[1], 2) for k, v in ctx.timing_report().items()}
        print(f"group {rep} pass {it} pinned={pin}: import {1e3*(t1-t0):.1f} ms, reencrypt {1e3*(t2-t1):.1f} ms, kernels {fam}", flush=True)
        for a in WP + S:
            a.free()
