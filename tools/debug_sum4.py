import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
vmn = entry.load_package()
from oracle import pyref
ctx = vmn.Context(0)
p, q, g = pyref.modp_group(3072)
G = vmn.ModPGroup(ctx, p, q, g)
R = pow(2, 28 * 110, q)
for xs in ([5, 7], [1, 0], [0, 0], [1 << 1540, 0], [(1 << 1540) + 3, 9]):
    s = G.ringArray(xs).sum()
    want = sum(xs) % q
    print([hex(x)[:12] for x in xs], "ok" if s == want else "BAD", hex(s)[:30], s.bit_length(), "ratio-check", (s * pow(want, -1, q)) % q if want else None)
# after some other activity
X = G.toElementArray([4, 16, 64]); _ = X.prod(); _ = X.mul(X).toInts()
print("after activity:", G.ringArray([5, 7]).sum())
print("prod ring:", G.ringArray([5, 7]).prod())
