import os, sys, time, importlib.util
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
vmn = entry.load_package()
from oracle import pyref
spec = importlib.util.spec_from_file_location("mx", os.path.join(entry.PKG_DIR, "randomsource.py")); mx = importlib.util.module_from_spec(spec); spec.loader.exec_module(mx)
n = 1000000
p, q, g = pyref.modp_group(2048)
ctx = vmn.Context(0)
G = vmn.ModPGroup(ctx, p, q, g, nbytes=256)
rnd = mx.InsecureBulkRandomSource(1, q, 256)
X = G.exp(g, G.ringArray(rnd.ring_array(n)))
for bits in (256, 612):
    E = G.ringArray(rnd.int_array(n, bits))
    for timing in (True, False):
        ctx.timing_enable(timing)
        r = X.exp(E, bits); r.free(); ctx.synchronize()
        ctx.timing_reset()
        t0 = time.perf_counter()
        for _ in range(2):
            r = X.exp(E, bits); r.free()
        t1 = time.perf_counter()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 2
        rep = ctx.timing_report() if timing else {}
        print(f"K1a {bits} bits, events {'on ' if timing else 'off'}: {dt*1e3:7.2f} ms/call -> {n/dt:,.0f}/s   host returns after {(t1-t0)/2*1e3:.2f} ms/call", {k: (v[0], round(v[1]/2, 2)) for k, v in rep.items()})
