#!/usr/bin/env python3
"""Rates of the untuned large-modulus geometries (eight / sixteen lanes per element): batched modpow with full-length
exponents at 8192 bits and over the reference's 15 492-bit benchmark group.  usage: python3 tools/big_modulus_rate.py [N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry          # noqa: E402

vmn = entry.load_package()
from oracle import pyref                 # noqa: E402
import importlib.util                    # noqa: E402

spec = importlib.util.spec_from_file_location("eio", os.path.join(entry.PKG_DIR, "eio.py"))
eio = importlib.util.module_from_spec(spec)
spec.loader.exec_module(eio)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = vmn.Context(0)
raw = bytes.fromhex(open(os.path.join(ROOT, "tests", "golden", "reference_modpgroup_bytetree.hex")).read().strip())
p15, q15, g15, _, width = eio.unmarshal_modpgroup(raw)
for label, (p, q, g), nb in (("8192 bits (RFC 3526 group 18)", pyref.modp_group(8192), None), ("15492 bits (reference bench group)", (p15, q15, g15), width)):
    G = vmn.ModPGroup(ctx, p, q, g, nbytes=nb) if nb else vmn.ModPGroup(ctx, p, q, g)
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(5))
    eb = G.exp_bytes
    e = rng.integers(0, 256, size=(n, eb), dtype=np.uint8)
    e[:, 0] = 0
    e[:, 1] &= 0x3F
    E = G.ringArrayFromBytes(e.tobytes()) if hasattr(G, "ringArrayFromBytes") else G.ringArray([int.from_bytes(r.tobytes(), "big") for r in e])
    X = G.exp(g, E)
    ctx.synchronize()
    t0 = time.perf_counter()
    Y = X.exp(E)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: N = {n}: modpow {dt * 1e3:.0f} ms = {n / dt:.3g} modexp/s")
