"""Where the set-up of a long-lived fixed base goes: vmn_group_precompute_fixed(base, n, uses) timed piece by piece
(wall clock around a synchronised call, the library's own kernel-family clock for the table build), for the windows
the one-shot and the session pricing pick.  GPU box only.
    python tools/setup_probe.py [bits] [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
vmn = entry.load_package()
from verificatum_vmn_amd import stdgroups as sg

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
p, q, g = sg.modp_group(bits)
t0 = time.perf_counter()
ctx = vmn.Context(0)
ctx.synchronize()
print(f"context: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
y = pow(g, 0x1234567890ABCDEF1234567890ABCDEF, p)
z = pow(g, 0x1234567890ABCDEF1234567890ABCDE1, p)
for uses in (1, 16, 16, 1):
    for w_env in (None,):
        t0 = time.perf_counter()
        G = vmn.ModPGroup(ctx, p, q, g, nbytes=bits // 8)
        ctx.synchronize()
        t1 = time.perf_counter()
        ctx.timing_reset()
        ctx.timing_enable(True)
        G.precomputeFixed(g, n, uses)
        t2 = time.perf_counter()
        ctx.synchronize()
        t3 = time.perf_counter()
        G.precomputeFixed(y, n, uses)
        ctx.synchronize()
        t4 = time.perf_counter()
        G.precomputeFixed(z, n, uses)
        ctx.synchronize()
        t5 = time.perf_counter()
        ctx.timing_enable(False)
        fam = ctx.timing_report()
        print(f"{bits} bits n = {n} uses = {uses}: group {(t1 - t0) * 1e3:.1f} ms; table g: call returns after {(t2 - t1) * 1e3:.1f} ms, "
              f"done after {(t3 - t1) * 1e3:.1f} ms; table y {(t4 - t3) * 1e3:.1f} ms; table z {(t5 - t4) * 1e3:.1f} ms; "
              f"table bytes {G.tableBytes()}; kernels "
              + ", ".join(f"{k}: {v[0]} launches {v[1]:.1f} ms" for k, v in fam.items()), flush=True)
        t0 = time.perf_counter()
        G.close()
        ctx.synchronize()
        print(f"    close: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
