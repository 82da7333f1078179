import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
vmn = entry.load_package()
from oracle import pyref
from oracle.cbind import Oracle
spec = importlib.util.spec_from_file_location("mx", os.path.join(entry.PKG_DIR, "mixnet.py")); mx = importlib.util.module_from_spec(spec); spec.loader.exec_module(mx)
bits = int(sys.argv[1]); n = int(sys.argv[2])
p, q, g = pyref.modp_group(bits); nb = bits // 8
ctx = vmn.Context(0); G = vmn.ModPGroup(ctx, p, q, g, nbytes=nb)
rnd = mx.BulkRandomSource(5, q, nb)
rb, eb = rnd.ring_array(n), rnd.int_array(n, 256)
R, E = G.ringArray(rb), G.ringArray(eb)
ri = [int.from_bytes(rb[i*nb:(i+1)*nb], "big") for i in range(n)]
ei = [int.from_bytes(eb[i*nb:(i+1)*nb], "big") for i in range(n)]
print("roundtrip", R.toInts() == ri)
print("inner", R.innerProduct(E) == sum(a*b for a, b in zip(ri, ei)) % q)
print("sum", R.sum() == sum(ri) % q)
pi = rnd.permutation(n)
print("permute", E.permute(pi).toInts() == [ei[int(j)] for j in pi])
v = 0x1234567890abcdef1234567890abcdef
ma = E.mulAdd(v, R).toInts()
print("mulAdd", ma == [(a*v + b) % q for a, b in zip(ei, ri)])
X = G.exp(g, R)
orc = Oracle(p, q, nb)
m = min(n, 3000)
xs = X.copyOfRange(0, m).toInts()
print("fixed", xs == orc.exp_fixed(g, ri[:m]))
Xs = X.copyOfRange(0, m); Es = E.copyOfRange(0, m)
print("expprod256 small", Xs.expProd(Es, 256) == orc.exp_prod(xs, ei[:m], 256, pippenger_c=8))
# large expProd consistency: split in two halves
h = n // 2
a = X.copyOfRange(0, h).expProd(E.copyOfRange(0, h), 256); b = X.copyOfRange(h, n).expProd(E.copyOfRange(h, n), 256)
print("expprod split", X.expProd(E, 256) == a * b % p)
print("prod split", X.prod() == X.copyOfRange(0, h).prod() * X.copyOfRange(h, n).prod() % p)
