import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
vmn = entry.load_package()
from oracle import pyref
ctx = vmn.Context(0)
p, q, g = pyref.modp_group(3072)
G = vmn.ModPGroup(ctx, p, q, g)
for xs in ([5, 7], [5, 7, 100], [5, 7, 100, 1000], [1, 10, 100, 1000, 10000], [3] * 9):
    print(xs, "->", G.ringArray(xs).sum(), "want", sum(xs))
print("inner [2,3]x[10,100] ->", G.ringArray([2, 3]).innerProduct(G.ringArray([10, 100])), "want", 320)
print("group prod [4,16] ->", G.toElementArray([4, 16]).prod())
