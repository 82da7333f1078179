"""Subgroup membership of 10^6 elements (2048 bits): Jacobi-symbol kernel against the x^q = 1 test.  GPU box only."""
import os, sys, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
vmn = entry.load_package()
spec = importlib.util.spec_from_file_location("mx", os.path.join(entry.PKG_DIR, "randomsource.py")); mx = importlib.util.module_from_spec(spec); spec.loader.exec_module(mx)
spec = importlib.util.spec_from_file_location("sg", os.path.join(entry.PKG_DIR, "stdgroups.py")); sg = importlib.util.module_from_spec(spec); spec.loader.exec_module(sg)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p, q, g = sg.modp_group(2048)
ctx = vmn.Context(0)
G = vmn.ModPGroup(ctx, p, q, g)
X = G.exp(g, G.ringArray(mx.InsecureBulkRandomSource(3, q, 256).ring_array(n)))
for mode in ("jacobi", "power"):
    if mode == "power":
        os.environ["VMN_MEMBER_BY_POWER"] = "1"
    X.isMember()
    ctx.synchronize(); t0 = time.perf_counter()
    ok = X.isMember()
    ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"{mode}: {n} elements in {dt * 1e3:.1f} ms = {n / dt / 1e6:.2f} M elements/s, all members: {ok}", flush=True)
