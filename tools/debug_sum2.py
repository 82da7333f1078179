import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
vmn = entry.load_package()
from oracle import pyref
ctx = vmn.Context(0)
bits = 3072
p, q, g = pyref.modp_group(bits)
G = vmn.ModPGroup(ctx, p, q, g)
import random
random.seed(1)
for trial in range(6):
    n = [2, 3, 5, 129, 300, 1500][trial]
    top = [q, 1 << 3070, 1 << 3000, q, 1 << 3070, 1 << 3070][trial]
    xs = [random.randrange(top) for _ in range(n)]
    ys = [random.randrange(top) for _ in range(n)]
    X, Y = G.ringArray(xs), G.ringArray(ys)
    add_ok = X.add(Y).toInts() == [(a + b) % q for a, b in zip(xs, ys)]
    s = X.sum()
    print(n, hex(top)[:6], "add", add_ok, "sum", s == sum(xs) % q, flush=True)
    if s != sum(xs) % q and n <= 5:
        print(" got ", hex(s)[:40], hex(s)[-20:]); print(" want", hex(sum(xs) % q)[:40], hex(sum(xs) % q)[-20:])
        d = (s - sum(xs)) % q
        print(" diff", hex(d)[:50], d.bit_length())
