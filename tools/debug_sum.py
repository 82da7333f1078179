import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
vmn = entry.load_package()
from oracle import pyref
spec = importlib.util.spec_from_file_location("mx", os.path.join(entry.PKG_DIR, "mixnet.py")); mx = importlib.util.module_from_spec(spec); spec.loader.exec_module(mx)
ctx = vmn.Context(0)
for bits in (2048, 3072):
    p, q, g = pyref.modp_group(bits); nb = bits // 8
    G = vmn.ModPGroup(ctx, p, q, g, nbytes=nb)
    rnd = mx.BulkRandomSource(5, q, nb)
    for n in (1500, 40000, 65536, 65537, 70000, 131072, 131073, 200000):
        rb = rnd.ring_array(n)
        R = G.ringArray(rb)
        ri = [int.from_bytes(rb[i*nb:(i+1)*nb], "big") for i in range(n)]
        print(bits, n, "sum", R.sum() == sum(ri) % q, flush=True)
