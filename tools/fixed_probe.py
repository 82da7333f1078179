"""Fixed-base exponentiation time against the table window (VMN_FIXED_WINDOW), N = 10^6, 2048 bits.  GPU box only."""
import os, sys, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
vmn = entry.load_package()
spec = importlib.util.spec_from_file_location("mx", os.path.join(entry.PKG_DIR, "randomsource.py")); mx = importlib.util.module_from_spec(spec); spec.loader.exec_module(mx)
spec = importlib.util.spec_from_file_location("sg", os.path.join(entry.PKG_DIR, "stdgroups.py")); sg = importlib.util.module_from_spec(spec); spec.loader.exec_module(sg)
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
p, q, g = sg.modp_group(bits)
ctx = vmn.Context(0)
rnd = mx.InsecureBulkRandomSource(1, q, bits // 8)
for w in (14, 16, 17, 18, 19, 20):
    os.environ["VMN_FIXED_WINDOW"] = str(w)
    G = vmn.ModPGroup(ctx, p, q, g, nbytes=bits // 8)
    E = G.ringArray(rnd.ring_array(n))
    ctx.timing_enable(True)
    t0 = time.perf_counter(); X = G.exp(g, E); ctx.synchronize(); build_and_first = time.perf_counter() - t0
    ctx.timing_reset()
    for _ in range(3):
        Y = G.exp(g, E); Y.free()
    ctx.synchronize()
    cnt, ms = ctx.timing_get("fixed")
    print(f"{bits} bits, N = {n}, w = {w}: {ms / cnt:.1f} ms per call ({n / (ms / cnt) * 1e3 / 1e6:.1f} M exp/s), first call incl. table build {build_and_first * 1e3:.0f} ms", flush=True)
    X.free(); E.free(); G.close()
