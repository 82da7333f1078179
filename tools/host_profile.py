"""cProfile of one warm mix+prove pass (bench.mix_prove) to see where host time goes.  GPU box only."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
import bench

vmn = entry.load_package()
ctx = vmn.Context(0)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
g, _ = load_golden(2048)
grp = vmn.ModPGroup(ctx, g["p"], g["q"], g["g"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sync = ctx.synchronize
bench.mix_prove(entry, vmn, ctx, grp, n, 7, sync)          # warm-up (tables, pool)
pr = cProfile.Profile()
pr.enable()
res = bench.mix_prove(entry, vmn, ctx, grp, n, 8, sync)
pr.disable()
print({k: res[k] for k in ("reencrypt_ms", "prove_ms", "verify_ms", "total_ms")})
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
