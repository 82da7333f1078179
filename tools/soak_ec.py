"""Soak over the curve group and the 3072-bit CCPoS path: repeated passes, device memory after each.  GPU box only."""
import gc, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
import bench

vmn = entry.load_package()
ctx = vmn.Context(0)
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for leg, n in (("ec", 200_000), ("ccpos", 100_000)):
    used = []
    for it in range(passes):
        res = bench.mix_ec(entry, vmn, ctx, n, 50 + it, ctx.synchronize) if leg == "ec" else bench.mix_ccpos(entry, vmn, ctx, n, 70 + it, ctx.synchronize)
        assert res["accepted"]
        del res
        gc.collect()
        ctx.synchronize()
        free, total = torch.cuda.mem_get_info()
        ms = ctx.memory_stats()
        used.append((total - free) / 2**30)
        print(f"{leg} pass {it:2d}: device memory in use {used[-1]:6.2f} GiB (live {ms['live_bytes'] / 2**30:.3f}, pool {ms['pool_bytes'] / 2**30:.2f} in {ms['pool_blocks']} blocks)", flush=True)
    tail = used[len(used) // 2:]
    assert max(tail) - min(tail) < 0.5, f"{leg}: device memory keeps growing"
print("soak ok")
