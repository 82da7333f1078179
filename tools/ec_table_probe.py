import time, sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
vmn = entry.load_package()
ctx = vmn.Context(0)
G = vmn.ECqPGroup(ctx, "P-256")
from oracle.pyref_ec import Curve
c = Curve("P-256")
for k in range(3):
    base = c.mul(1234567 + k, c.g)
    ctx.synchronize()
    t = time.perf_counter()
    G.precomputeFixed(base, 1000000, 16)
    ctx.synchronize()
    print("table for a new base: %.2f ms" % ((time.perf_counter() - t) * 1e3))
