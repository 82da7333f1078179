#!/usr/bin/env python3
"""Ad-hoc GPU check used while bringing kernels up (the real parity tests are in tests/)."""
import importlib.util, os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("verificatum_vmn_amd", os.path.join(ROOT, "verificatum-vmn_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "verificatum-vmn_amd")])
vmn = importlib.util.module_from_spec(spec); sys.modules["verificatum_vmn_amd"] = vmn; spec.loader.exec_module(vmn)

P14 = int("FFFFFFFFFFFFFFFFC90FDAA22168C234C4C6628B80DC1CD129024E088A67CC74020BBEA63B139B22514A08798E3404DD"
          "EF9519B3CD3A431B302B0A6DF25F14374FE1356D6D51C245E485B576625E7EC6F44C42E9A637ED6B0BFF5CB6F406B7ED"
          "EE386BFB5A899FA5AE9F24117C4B1FE649286651ECE45B3DC2007CB8A163BF0598DA48361C55D39A69163FA8FD24CF5F"
          "83655D23DCA3AD961C62F356208552BB9ED529077096966D670C354E4ABC9804F1746C08CA18217C32905E462E36CE3B"
          "E39E772C180E86039B2783A2EC07A28FB5C55DF06F4C52C9DE2BCBF6955817183995497CEA956AE515D2261898FA0510"
          "15728E5A8AACAA68FFFFFFFFFFFFFFFF", 16)

def stream(seed, n, mod):
    out, ctr = [], 0
    nb = (mod.bit_length() + 7) // 8 + 8
    while len(out) < n:
        buf = b""
        while len(buf) < nb:
            buf += hashlib.sha256(seed + ctr.to_bytes(8, "big")).digest(); ctr += 1
        out.append(int.from_bytes(buf[:nb], "big") % mod)
    return out

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    p, q, g = P14, (P14 - 1) // 2, 4
    ctx = vmn.Context(0)
    print(vmn.lib().vmn_version().decode(), "CUs", ctx.num_cus)
    grp = vmn.ModPGroup(ctx, p, q, g)
    xs = [1 + v % (p - 1) for v in stream(b"x", n, p)]
    ys = [1 + v % (p - 1) for v in stream(b"y", n, p)]
    es = stream(b"e", n, q)
    X = grp.toElementArray(xs); Y = grp.toElementArray(ys); E = grp.ringArray(es)
    assert X.toInts() == xs, "roundtrip"
    assert E.toInts() == es, "ring roundtrip"
    Z = X.mul(Y)
    assert Z.toInts() == [a * b % p for a, b in zip(xs, ys)], "mul"
    print("import/export/mul ok")
    t0 = time.time(); R = X.exp(E); got = R.toInts(); t1 = time.time()
    ncheck = min(n, 300)
    for i in list(range(ncheck // 2)) + list(range(n - ncheck // 2, n)):
        assert got[i] == pow(xs[i], es[i], p), f"modpow mismatch at {i}"
    print(f"modpow ok ({n} elements, {t1 - t0:.3f}s incl. export)")
    R2 = X.exp(0x2ABCDEF012345)
    assert R2.toInts()[:50] == [pow(x, 0x2ABCDEF012345, p) for x in xs[:50]], "scalar exp"
    e612 = [v % (1 << 612) for v in stream(b"k", n, 1 << 640)]
    R3 = X.expInts(e612, 612)
    assert R3.toInts()[:50] == [pow(x, e, p) for x, e in zip(xs[:50], e612[:50])], "int exp"
    print("scalar/int exp ok")
    if n >= 100000:
        ctx.timing_enable(True)
        for _ in range(2):
            R = X.exp(E)
        cnt, ms = ctx.timing_get("modpow")
        print(f"modpow kernel: {cnt} launches, {ms / cnt:.2f} ms each, {n / (ms / cnt) * 1e3:.0f} modexp/s")

main()
