#!/usr/bin/env python3
"""Ad-hoc probe: batched modpow / fixed-base / expProd throughput for a given modulus size."""
import os, sys, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
vmn = entry.load_package()
from oracle import pyref
spec = importlib.util.spec_from_file_location("mx", os.path.join(entry.PKG_DIR, "randomsource.py")); mx = importlib.util.module_from_spec(spec); spec.loader.exec_module(mx)
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
p, q, g = pyref.modp_group(bits)
nb = bits // 8
ctx = vmn.Context(0)
G = vmn.ModPGroup(ctx, p, q, g, nbytes=nb)
rnd = mx.InsecureBulkRandomSource(1, q, nb)
E = G.ringArray(rnd.ring_array(n))
ctx.timing_enable(True)
X = G.exp(g, E)
E2 = G.ringArray(rnd.ring_array(n))
for rep in range(2):
    ctx.timing_reset()
    R = X.exp(E2); ctx.synchronize()
    cnt, ms = ctx.timing_get("modpow")
    print(f"{bits}-bit modpow n={n}: {ms/cnt:.1f} ms -> {n/(ms/cnt)*1e3:,.0f} modexp/s")
    R.free()
ctx.timing_reset(); Y = G.exp(g, E2); ctx.synchronize(); cnt, ms = ctx.timing_get("fixed"); print(f"fixed-base: {ms:.1f} ms -> {n/ms*1e3:,.0f} exp/s")
e612 = G.ringArray(rnd.int_array(n, 612))
t0 = time.perf_counter(); r = X.expProd(e612, 612); ctx.synchronize(); print(f"expProd 612-bit: {1e3*(time.perf_counter()-t0):.1f} ms")
