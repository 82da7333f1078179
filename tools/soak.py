"""Soak: repeated mix + prove passes with a NEW set of generators per pass (so a new fixed base h_0 each time), the
device memory in use after every pass.  The array pool and the fixed-base table cache must level off.  GPU box only."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
import bench

vmn = entry.load_package()
ctx = vmn.Context(0)
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
p, q, g = bench.load_sub(entry, "stdgroups").modp_group(2048)
grp = vmn.ModPGroup(ctx, p, q, g)
used = []
for it in range(passes):
    res = bench.mix_prove(entry, vmn, ctx, grp, n, 1000 + it, ctx.synchronize)      # new instance (new h, new h_0) per pass
    assert res["accepted"]
    gc.collect()
    ctx.synchronize()
    free, total = torch.cuda.mem_get_info()
    used.append((total - free) / 2**30)
    ms = ctx.memory_stats()
    tb = vmn.lib().vmn_group_table_bytes(grp._h)
    print(f"pass {it:2d}: {res['total_ms']:7.1f} ms, device memory in use {used[-1]:6.2f} GiB "
          f"(live {ms['live_bytes'] / 2**30:.2f}, pool {ms['pool_bytes'] / 2**30:.2f} in {ms['pool_blocks']} blocks, tables {tb / 2**30:.2f})", flush=True)
tail = used[len(used) // 2:]
print("second half: min %.2f GiB, max %.2f GiB" % (min(tail), max(tail)))
assert max(tail) - min(tail) < 1.0, "device memory keeps growing"
print("soak ok")
