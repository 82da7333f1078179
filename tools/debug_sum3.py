import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
vmn = entry.load_package()
from oracle import pyref
from conftest import load_golden, ints
ctx = vmn.Context(0)
grp, cases = load_golden(3072)
p, q, g = grp["p"], grp["q"], grp["g"]
G = vmn.ModPGroup(ctx, p, q, g)
for c in cases:
    if c["op"] == "ring_sum":
        xs = ints(c["x"])
        print("golden n", c["n"], G.ringArray(xs).sum() == int(c["out"], 16))
import random
random.seed(1)
xs = [random.randrange(q) for _ in range(2)]
X = G.ringArray(xs)
print("first sum ", X.sum() == sum(xs) % q)
print("second sum", X.sum() == sum(xs) % q)
a = X.copyOfRange(0, 1).add(X.copyOfRange(1, 2)).toInts()[0]
print("add of halves", a == sum(xs) % q)
print("values < q:", all(x < q for x in xs), [x.bit_length() for x in xs])
xs2 = [q - 5, 7]
print("wrap case", G.ringArray(xs2).sum() == 2)
xs3 = [5, 7]
print("small case", G.ringArray(xs3).sum() == 12)
xs4 = [(1 << 1500) + 5, (1 << 1600) + 7]
print("mid case", G.ringArray(xs4).sum() == sum(xs4) % q)
xs5 = [(1 << 3000) + 5, (1 << 3001) + 7]
print("high case", G.ringArray(xs5).sum() == sum(xs5) % q)
xs6 = [(1 << 3069) + 5, (1 << 3069) + 7]
print("carry-to-top case", G.ringArray(xs6).sum() == sum(xs6) % q)
xs7 = [(1 << 3070) + 5, (1 << 3070) + 7]
print("ge case", G.ringArray(xs7).sum() == sum(xs7) % q, sum(xs7) >= q)
