"""Soak: repeated mix + prove passes with a NEW set of generators per pass (so a new fixed base h_0 each time), the
device memory in use after every pass.  The array pool and the fixed-base table cache must level off.  GPU box only."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The memory a long session holds is bounded by three caps: the array pool, the cache of fixed-base tables of a group
# (VMN_FIXED_CACHE_BYTES, 64 GB) and the arena that evicted / freed tables return to (VMN_TABLE_ARENA_BYTES, 48 GB, round 4).
# At 3.2 GB of new table per pass the defaults take ~40 passes to fill; the soak runs with small caps so that its 30 passes
# reach the steady state, and then asks for a flat second half.
os.environ.setdefault("VMN_FIXED_CACHE_BYTES", str(16 << 30))
os.environ.setdefault("VMN_TABLE_ARENA_BYTES", str(8 << 30))
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
assert max(tail) - min(tail) < 4.0, "device memory keeps growing"      # (one 3.2 GB table in or out of the arena)
print("soak ok")
