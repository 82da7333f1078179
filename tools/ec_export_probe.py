#!/usr/bin/env python3
"""Byte tree of a large array of curve points whose rows are Jacobian (results of device operations): the export with one Fermat
inversion per point against the export that normalises the rows first (VMN_EC_EXPORT_NORMALISE_MIN).  GPU box only.
usage: python tools/ec_export_probe.py [curve] [n]"""
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
X = G.exp(G.g, G.ringArray(rnd.ring_array(n)))
ctx.synchronize()
ctx.timing_enable(True)
out = {}
for label, bound in (("one inversion per point", "1000000000"), ("rows normalised first", "1")):
    os.environ["VMN_EC_EXPORT_NORMALISE_MIN"] = bound
    X.toByteTree()                                     # warm-up (buffers)
    ctx.timing_reset()
    t0 = time.perf_counter()
    bt = X.toByteTree()
    dt = time.perf_counter() - t0
    rep = ctx.timing_report()
    kern = {k: round(v[1], 2) for k, v in rep.items() if k in ("export", "normalize")}
    print(f"{name} n={n} {label:24s}: {dt * 1e3:8.1f} ms wall ({len(bt) / dt / 1e9:.2f} GB/s of byte tree), kernels {kern}")
    out[label] = bt
assert out["one inversion per point"] == out["rows normalised first"], "the two exports differ"
print("byte trees identical")
