#!/usr/bin/env python3
"""Ad-hoc probe: throughput of the elliptic-curve array kernels."""
import os, sys, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
vmn = entry.load_package()
spec = importlib.util.spec_from_file_location("mx", os.path.join(entry.PKG_DIR, "randomsource.py")); mx = importlib.util.module_from_spec(spec); spec.loader.exec_module(mx)
name = sys.argv[1] if len(sys.argv) > 1 else "P-256"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
ctx = vmn.Context(0)
G = vmn.ECqPGroup(ctx, name)
rnd = mx.InsecureBulkRandomSource(1, G.q, G.exp_bytes)
E = G.ringArray(rnd.ring_array(n)); E2 = G.ringArray(rnd.ring_array(n))
ctx.timing_enable(True)
X = G.exp(G.g, E); ctx.synchronize()
for rep in range(2):
    ctx.timing_reset(); Y = G.exp(G.g, E2); ctx.synchronize(); c, ms = ctx.timing_get("fixed"); print(f"fixed-base  n={n}: {ms:.2f} ms -> {n/ms*1e3:,.0f}/s"); Y.free()
for rep in range(2):
    ctx.timing_reset(); R = X.exp(E2); ctx.synchronize(); c, ms = ctx.timing_get("modpow"); print(f"var-base    n={n}: {ms:.2f} ms -> {n/ms*1e3:,.0f}/s"); R.free()
ctx.timing_reset(); Z = X.mul(X); ctx.synchronize(); c, ms = ctx.timing_get("modmul"); print(f"point add   n={n}: {ms:.2f} ms -> {n/ms*1e3:,.0f}/s")
for bits in (256,):
    t0 = time.perf_counter(); ctx.timing_reset(); r = X.expProd(E2); ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"expProd     n={n}: {dt*1e3:.1f} ms wall -> {n/dt:,.0f} terms/s", {k: round(v[1], 2) for k, v in ctx.timing_report().items()})
