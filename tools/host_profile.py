"""cProfile of one warm pass of a mix leg (bench.mix_prove / mix_ec / mix_ccpos) to see where host time goes.
usage: host_profile.py [prove|ec|ccpos] [n]      GPU box only."""
import cProfile, gc, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
import bench

vmn = entry.load_package()
ctx = vmn.Context(0)
leg = sys.argv[1] if len(sys.argv) > 1 else "prove"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (1_000_000 if leg == "prove" else 400_000)
sync = ctx.synchronize
if leg == "prove":
    p, q, g = bench.load_sub(entry, "stdgroups").modp_group(2048)
    grp = vmn.ModPGroup(ctx, p, q, g)
    run = lambda seed: bench.mix_prove(entry, vmn, ctx, grp, n, seed, sync)
elif leg == "ec":
    run = lambda seed: bench.mix_ec(entry, vmn, ctx, n, seed, sync)
else:
    run = lambda seed: bench.mix_ccpos(entry, vmn, ctx, n, seed, sync)
run(7)          # warm-up (tables, pool)
gc.collect()
gc.disable()
pr = cProfile.Profile()
pr.enable()
res = run(8)
pr.disable()
print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items() if k.endswith("_ms") or k.startswith("ciphertexts")})
print(res["kernel_ms_by_family"])
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
