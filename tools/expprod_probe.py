#!/usr/bin/env python3
"""Ad-hoc probe: time expProd alone (used while tuning the multi-exponentiation)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
vmn = entry.load_package()
from oracle import pyref
import importlib.util
spec = importlib.util.spec_from_file_location("mx", os.path.join(entry.PKG_DIR, "randomsource.py")); mx = importlib.util.module_from_spec(spec); spec.loader.exec_module(mx)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 613
p, q, g = pyref.modp_group(2048)
ctx = vmn.Context(0)
G = vmn.ModPGroup(ctx, p, q, g, nbytes=256)
rnd = mx.InsecureBulkRandomSource(1, q, 256)
X = G.exp(g, G.ringArray(rnd.ring_array(n)))
E = G.ringArray(rnd.int_array(n, bits))
for rep in range(2):
    ctx.timing_reset(); ctx.timing_enable(True)
    t0 = time.perf_counter(); r = X.expProd(E, bits); ctx.synchronize(); t1 = time.perf_counter()
    ctx.timing_enable(False)
    print(f"expProd n={n} bits={bits}: {1e3*(t1-t0):.1f} ms wall; families:", {k: (v[0], round(v[1], 2)) for k, v in ctx.timing_report().items()})
