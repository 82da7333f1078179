#!/usr/bin/env python3
"""Subgroup membership of an array: the Jacobi-symbol kernels against x^q = 1 (VMN_MEMBER_BY_POWER=1), per modulus size.
usage: python3 tools/member_timing.py [N]      (run once with and once without VMN_MEMBER_BY_POWER=1)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry          # noqa: E402

vmn = entry.load_package()
from oracle import pyref                 # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = vmn.Context(0)
mode = "x^q" if os.environ.get("VMN_MEMBER_BY_POWER") else "jacobi"
for bits in (2048, 3072, 4096):
    p, q, g = pyref.modp_group(bits)
    G = vmn.ModPGroup(ctx, p, q, g)
    X = G.exp(g, G.ringArray(pyref.stream_ints(b"member-timing", n, q)))
    X.isMember()
    ctx.synchronize()
    t0 = time.perf_counter()
    ok = X.isMember()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode} {bits} bits: N = {n}: {dt * 1e3:.1f} ms ({n / dt:.3g} elements/s), all members: {ok}")
