"""Host-to-device import rate of big-endian blocks (pageable Python bytes) and a host-time breakdown of one
mix+prove pass.  Run on the GPU box: python tools/h2d_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry

vmn = entry.load_package()
ctx = vmn.Context(0)
sys.path.insert(0, os.path.join(entry.ROOT, "tests"))
from conftest import load_golden
grp, _ = load_golden(2048)
G = vmn.ModPGroup(ctx, grp["p"], grp["q"], grp["g"])
n = 1_000_000
rng = np.random.Generator(np.random.PCG64(1))
a = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
a[:, 0] &= 0x3F
buf = a.tobytes()
for rep in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    R = G.ringArray(buf)
    ctx.synchronize(); t1 = time.perf_counter()
    print(f"ringArray import of {len(buf)/1e6:.0f} MB: {(t1-t0)*1e3:.1f} ms = {len(buf)/(t1-t0)/1e9:.1f} GB/s")
    t0 = time.perf_counter()
    out = R.toBytes() if hasattr(R, "toBytes") else None
    ctx.synchronize(); t1 = time.perf_counter()
    if out is not None:
        print(f"export: {(t1-t0)*1e3:.1f} ms")
    R.free()
