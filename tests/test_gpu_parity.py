"""GPU suite: the HIP path, called through the C ABI, against the golden vectors, the GMP oracle on
seeded inputs, and size-independent properties at larger sizes."""
import pytest

from conftest import ints, load_golden
from oracle import pyref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def groups(vmn, gpu_ctx):
    out = {}
    for bits in (512, 1024, 2048, 3072, 4096):
        grp, cases = load_golden(bits)
        out[bits] = (vmn.ModPGroup(gpu_ctx, grp["p"], grp["q"], grp["g"]), grp, cases)
    return out


@pytest.mark.parametrize("bits", [512, 1024, 2048, 3072, 4096])
def test_golden_vectors_through_c_abi(bits, groups):
    G, grp, cases = groups[bits]
    for c in cases:
        op = c["op"]
        if op == "exp_array":
            assert G.toElementArray(ints(c["x"])).exp(G.ringArray(ints(c["e"]))).toInts() == ints(c["out"]), (op, c["n"])
        elif op == "exp_ints":
            assert G.toElementArray(ints(c["x"])).expInts(ints(c["e"]), c["ebits"]).toInts() == ints(c["out"]), (op, c["n"])
        elif op == "exp_scalar":
            assert G.toElementArray(ints(c["x"])).exp(int(c["e"], 16)).toInts() == ints(c["out"]), (op, c["n"])
        elif op == "exp_fixed":
            assert G.exp(int(c["base"], 16), G.ringArray(ints(c["e"]))).toInts() == ints(c["out"]), (op, c["n"])
        elif op == "exp_prod":
            assert G.toElementArray(ints(c["x"])).expProd(ints(c["e"]), c["ebits"]) == int(c["out"], 16), (op, c["n"])
        elif op == "exp_prod_ring":
            assert G.toElementArray(ints(c["x"])).expProd(G.ringArray(ints(c["e"]))) == int(c["out"], 16), (op, c["n"])
        elif op == "mul":
            assert G.toElementArray(ints(c["x"])).mul(G.toElementArray(ints(c["y"]))).toInts() == ints(c["out"])
        elif op == "prod":
            assert G.toElementArray(ints(c["x"])).prod() == int(c["out"], 16), (op, c["n"])
        elif op == "permute":
            assert G.toElementArray(ints(c["x"])).permute(c["perm"]).toInts() == ints(c["out"])
        elif op == "shift_push":
            assert G.toElementArray(ints(c["x"])).shiftPush(int(c["el"], 16)).toInts() == ints(c["out"])
        elif op == "rec_lin":
            x, d = G.ringArray(ints(c["b"])).recLin(G.ringArray(ints(c["e"])))
            assert x.toInts() == ints(c["out"]) and d == int(c["last"], 16), (op, c["n"])
        elif op == "prods":
            assert G.ringArray(ints(c["e"])).prods().toInts() == ints(c["out"]), (op, c["n"])
        elif op == "mul_add":
            got = G.ringArray(ints(c["x"])).mulAdd(int(c["v"], 16), G.ringArray(ints(c["y"]))).toInts()
            assert got == ints(c["out"])
        elif op == "ring_mul":
            assert G.ringArray(ints(c["x"])).mul(G.ringArray(ints(c["y"]))).toInts() == ints(c["out"])
        elif op == "ring_add":
            assert G.ringArray(ints(c["x"])).add(G.ringArray(ints(c["y"]))).toInts() == ints(c["out"])
        elif op == "inner_product":
            assert G.ringArray(ints(c["x"])).innerProduct(G.ringArray(ints(c["y"]))) == int(c["out"], 16)
        elif op == "ring_sum":
            assert G.ringArray(ints(c["x"])).sum() == int(c["out"], 16)
        elif op == "ring_prod":
            assert G.ringArray(ints(c["x"])).prod() == int(c["out"], 16)


def _inputs(tag, n, p, q):
    xs = [pow(1 + v % (p - 1), 2, p) for v in pyref.stream_ints(tag + b"/x", n, p)]
    es = pyref.stream_ints(tag + b"/e", n, q)
    return xs, es


@pytest.mark.parametrize("bits,n", [(2048, 257), (2048, 5000), (3072, 129), (3072, 1500), (4096, 131), (4096, 300)])
def test_seeded_arrays_against_gmp_oracle(bits, n, groups, oracle_for):
    """Ragged sizes (not multiples of the workgroup tile) against the GMP oracle, element for element;
    3072 bits exercises the two-lanes-per-element kernels, 4096 bits the four-lane ones."""
    G, grp, _ = groups[bits]
    p, q, g = grp["p"], grp["q"], grp["g"]
    orc = oracle_for(p, q)
    xs, es = _inputs(b"seeded%d" % n, n, p, q)
    fs = pyref.stream_ints(b"seeded/f", n, q)
    X, E, F = G.toElementArray(xs), G.ringArray(es), G.ringArray(fs)
    assert X.exp(E).toInts() == orc.exp_array(xs, es)
    assert G.exp(g, E).toInts() == orc.exp_fixed(g, es)
    ys = orc.exp_fixed(g, fs)
    assert X.mul(G.toElementArray(ys)).toInts() == orc.mul(xs, ys)
    assert X.prod() == orc.prod(xs)
    e256 = [v % (1 << 256) for v in es]
    assert X.expProd(e256, 256) == orc.exp_prod(xs, e256, 256, pippenger_c=8)
    assert X.expProd(E) == orc.exp_prod(xs, es, pippenger_c=8)
    x, d = E.recLin(F)
    want = orc.rec_lin(es, fs)
    assert x.toInts() == want and d == want[-1]
    assert F.prods().toInts() == orc.prods(fs)
    assert E.innerProduct(F) == orc.ring_reduce(es, fs, 0)
    assert E.sum() == orc.ring_reduce(es, None, 1)


def test_empty_and_single(groups):
    G, grp, _ = groups[512]
    p, q = grp["p"], grp["q"]
    X0, E0 = G.toElementArray([]), G.ringArray([])
    assert X0.size() == 0 and X0.toInts() == []
    assert X0.exp(E0).toInts() == []
    assert X0.prod() == 1
    assert X0.expProd(E0) == 1
    assert E0.sum() == 0 and E0.prod() == 1
    X1 = G.toElementArray([4])
    assert X1.exp(G.ringArray([q - 1])).toInts() == [pow(4, q - 1, p)]
    assert X1.exp(0).toInts() == [1]
    assert X1.prod() == 4


def test_out_of_range_import_is_reported_not_fatal(groups, vmn):
    G, grp, _ = groups[512]
    p = grp["p"]
    arr = G.toElementArray([5, p, 7], checked=False)
    assert arr.all_in_range is False
    assert arr.toInts() == [5, 1, 7]          # offending entry replaced by the trivial value
    with pytest.raises(ValueError):
        G.toElementArray([p + 1])


def test_equals_extract_range_get_membership(groups):
    G, grp, _ = groups[512]
    p, q = grp["p"], grp["q"]
    xs = [pow(3 + i, 2, p) for i in range(300)]
    X = G.toElementArray(xs)
    assert X.equals(G.toElementArray(xs))
    ys = list(xs)
    ys[299] = pow(2, 2, p)
    assert not X.equals(G.toElementArray(ys))
    keep = [i % 3 == 0 for i in range(300)]
    assert X.extract(keep).toInts() == [x for x, k in zip(xs, keep) if k]
    assert X.copyOfRange(10, 20).toInts() == xs[10:20]
    assert X.get(123) == xs[123]
    assert X.isMember()
    assert not G.toElementArray([pow(3, 2, p), p - 1]).isMember()     # p-1 has order 2


def test_wire_width_with_sign_byte(vmn, gpu_ctx):
    """VCR writes fixed-width two's-complement integers: 65 bytes for a 512-bit modulus
    (1937 bytes for a 15492-bit p).  Import/export must honour any width."""
    grp, _ = load_golden(512)
    p, q, g = grp["p"], grp["q"], grp["g"]
    G = vmn.ModPGroup(gpu_ctx, p, q, g, nbytes=65)
    xs = [pow(7 + i, 2, p) for i in range(70)]
    X = G.toElementArray(xs)
    raw = X.toBytes()
    assert len(raw) == 70 * 65 and all(raw[i * 65] == 0 for i in range(70))
    assert X.toInts() == xs
    assert X.exp(G.ringArray([q - 2] * 70)).toInts() == [pow(x, q - 2, p) for x in xs]


def test_properties_at_scale_2048(groups):
    """Size-independent properties at a size the CPU oracle would need minutes for:
    (x^e)^f = (x^f)^e ; x^e * x^f = x^(e+f) ; expProd = prod of exps ; g^e via fixed base = var base."""
    G, grp, _ = groups[2048]
    p, q, g = grp["p"], grp["q"], grp["g"]
    n = 40000
    xs, es = _inputs(b"scale", n, p, q)
    fs = pyref.stream_ints(b"scale/f", n, q)
    X, E, F = G.toElementArray(xs), G.ringArray(es), G.ringArray(fs)
    XE = X.exp(E)
    XF = X.exp(F)
    assert XE.exp(F).equals(XF.exp(E))
    assert XE.mul(XF).equals(X.exp(E.add(F)))
    assert X.expProd(E) == XE.prod()
    Gs = G.toElementArray([g] * n)
    assert G.exp(g, E).equals(Gs.exp(E))
    # spot-check a few entries against Python
    got = XE.toInts()
    for i in (0, 1, n // 2, n - 1):
        assert got[i] == pow(xs[i], es[i], p)


def test_properties_at_full_baseline_size(groups):
    """BASELINE.json configs[1] size (1 000 000 elements, 2048 bits): the CPU oracle would need hours, so the
    result is pinned through size-independent properties plus spot checks against Python."""
    import numpy as np
    G, grp, _ = groups[2048]
    p, q, g = grp["p"], grp["q"], grp["g"]
    n = 1_000_000
    rng = np.random.Generator(np.random.PCG64(99))
    def block(clear_top_bits):
        a = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
        a[:, 0] &= 0xFF >> clear_top_bits
        return a
    assert G.exp_bytes == 256                         # Java's width of the 2047-bit q: the blocks below are exponent rows
    eb, fb = block(2), block(2)                       # exponents < 2^2046: e + f < q, no wrap
    E, F = G.ringArray(eb.tobytes()), G.ringArray(fb.tobytes())
    X = G.exp(g, G.ringArray(block(1).tobytes()))     # random subgroup elements
    XE, XF = X.exp(E), X.exp(F)
    assert XE.mul(XF).equals(X.exp(E.add(F)))         # x^e x^f = x^(e+f), element-wise over 10^6 elements
    Gs = G.toElementArray(int(g).to_bytes(G.nbytes, "big") * n)
    assert G.exp(g, E).equals(Gs.exp(E))              # fixed-base table path = variable-base path
    assert X.expProd(E) == XE.prod()                  # Pippenger = product of the individual powers
    for i in (0, 1, 499_999, 999_999):
        x = X.get(i)
        e = int.from_bytes(eb[i].tobytes(), "big")
        assert XE.get(i) == pow(x, e, p)


def test_helper_thread_runs_concurrently_with_main_thread(groups):
    """The reference's one concurrency pattern (ShufflerElGamalSession.java:839-859): a helper thread does
    mul + permute on arrays while the protocol thread runs a verification.  ctypes drops the GIL during the
    foreign calls, so the two threads really enter the library together; results must equal the sequential ones."""
    import threading
    G, grp, _ = groups[2048]
    p, q = grp["p"], grp["q"]
    n = 700
    xs, es = _inputs(b"thr", n, p, q)
    ys, _ = _inputs(b"thr2", n, p, q)
    perm = sorted(range(n), key=lambda i: (xs[i], i))
    X, Y, E = G.toElementArray(xs), G.toElementArray(ys), G.ringArray(es)
    want_main = [pyref.exp_prod(xs, es, p), pyref.exp_prod(ys, es, p)]
    prod_xy = [a * b % p for a, b in zip(xs, ys)]
    want_helper = pyref.permute(prod_xy, perm)
    got_helper, errors = [], []

    def helper():
        try:
            for _ in range(12):
                t = X.mul(Y)
                r = t.permute(perm)
                t.free()
                got_helper.append(r.toInts())
                r.free()
        except Exception as exc:      # pragma: no cover
            errors.append(exc)

    th = threading.Thread(target=helper)
    th.start()
    got_main = []
    for _ in range(6):
        got_main.append([X.expProd(E), Y.expProd(E)])
    th.join()
    assert not errors
    assert all(g == want_main for g in got_main)
    assert len(got_helper) == 12 and all(g == want_helper for g in got_helper)


def test_helper_lane_overlaps_with_the_protocol_thread(vmn, gpu_ctx):
    """The helper thread announced with vmn_ctx_helper_begin gets its own stream, pool and lock: while the protocol
    thread has ~a second of fixed-base exponentiations queued (the prover's commit), the helper's mul + permute + byte-tree
    export (ShufflerElGamalSession.java:839-859; CCPoSW.java:114-123) finish long before that work does -- on one lane
    they would queue behind it.  Results equal the sequential ones; fixed-base tables built by one lane serve the other."""
    import threading
    import time
    import numpy as np
    grp, _ = load_golden(2048)
    p, q, g = grp["p"], grp["q"], grp["g"]
    G = vmn.ModPGroup(gpu_ctx, p, q, g, nbytes=256)
    n = 300_000
    rng = np.random.Generator(np.random.PCG64(5))
    def rows():
        a = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
        a[:, 0] &= 0x3F
        return a.tobytes()
    E = [G.ringArray(rows()) for _ in range(2)]
    X, Y = G.exp(g, E[0]), G.exp(g, E[1])
    perm = rng.permutation(n).astype(np.uint32)
    want = X.mul(Y).permute(perm)
    want_bt = want.toByteTree()
    gpu_ctx.synchronize()
    # the protocol thread's work alone
    t0 = time.perf_counter()
    outs = [G.exp(g, E[k % 2]) for k in range(24)]
    gpu_ctx.synchronize()
    t_main = time.perf_counter() - t0
    for o in outs:
        o.free()
    result = {}

    def helper():
        try:
            with gpu_ctx.helper():
                t1 = time.perf_counter()
                t = X.mul(Y)
                r = t.permute(perm)
                t.free()
                bt = r.toByteTree()                       # blocks until the helper's own stream has produced it
                result["t"] = time.perf_counter() - t1
                result["ok"] = bt == want_bt and r.equals(want)
                h = G.exp(g, E[0])                        # a table built on the main lane, used from the helper lane
                result["fixed"] = h.equals(X)
                h.free()
                r.free()
        except Exception as exc:      # pragma: no cover
            result["error"] = exc

    gpu_ctx.helper_mark()                                # X, Y, E are complete: what the helper relies on
    t0 = time.perf_counter()
    outs = [G.exp(g, E[k % 2]) for k in range(24)]      # queued asynchronously: ~t_main of GPU work ahead
    th = threading.Thread(target=helper)
    th.start()
    th.join()
    t_helper_done = time.perf_counter() - t0
    gpu_ctx.synchronize()
    t_both = time.perf_counter() - t0
    assert "error" not in result, result.get("error")
    assert result["ok"] and result["fixed"]
    assert all(o.equals(X if k % 2 == 0 else Y) for k, o in enumerate(outs))
    # overlap: the helper was done well before the protocol thread's queue drained, and the whole took about as long as
    # the protocol thread's work alone
    assert t_helper_done < 0.6 * t_main, (t_helper_done, t_main, result["t"])
    assert t_both < 1.35 * t_main, (t_both, t_main)


def test_fixed_base_table_cache_is_bounded(vmn, gpu_ctx, monkeypatch):
    """Every proof brings a new fixed base (h_0): the per-group table cache drops least-recently-used tables beyond
    its byte bound instead of growing for ever.  With a bound below one table, every call rebuilds; results stay
    bit-exact and a dropped base is rebuilt on demand."""
    grp, _ = load_golden(1024)
    p, q, g = grp["p"], grp["q"], grp["g"]
    monkeypatch.setenv("VMN_FIXED_CACHE_BYTES", str(3 << 20))          # < two tables at this size
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    n = 300
    es = pyref.stream_ints(b"cache/e", n, q)
    E = G.ringArray(es)
    bases = [pow(g, k, p) for k in pyref.stream_ints(b"cache/b", 4, q)]
    for rnd in range(2):
        for b in bases + [bases[0]]:
            assert G.exp(b, E).toInts() == pyref.exp_fixed(b, es, p), rnd


@pytest.mark.parametrize("bits,n", [(2048, 600), (3072, 300), (1024, 700), (4096, 150)])      # (three tiles and a ragged fourth)
def test_modpow_in_phases_is_the_same_power(bits, n, vmn, gpu_ctx, monkeypatch):
    """Arrays of more than one round of tiles run k_modpow_phased: a tile's power is cut into runs of windows that different
    workgroups take from a queue, the running value and the window table handed over through memory.  With the "device" shrunk
    to two workgroup slots (VMN_MODPOW_MAX_BLOCKS) small arrays take that kernel too: ragged sizes, full-length and short
    exponents, against Python's pow."""
    grp, _ = load_golden(bits) if bits in (1024, 2048) else (None, None)
    if grp is None:
        p, q, g = pyref.modp_group(bits)
    else:
        p, q, g = grp["p"], grp["q"], grp["g"]
    monkeypatch.setenv("VMN_MODPOW_MAX_BLOCKS", "2")
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    xs = [pow(g, k, p) for k in pyref.stream_ints(b"phased/x%d" % bits, n, q)]
    X = G.toElementArray(xs)
    for ebits in (q.bit_length(), 37):                     # (the expected values are Python pows: a third of a minute in all)
        es = [e % (1 << ebits) for e in pyref.stream_ints(b"phased/e%d" % ebits, n, 1 << ebits)]
        es[0], es[-1] = 0, (1 << ebits) - 1 if ebits < q.bit_length() else q - 1
        es = [e % q for e in es]
        got = X.exp(G.ringArray(es), 0 if ebits == q.bit_length() else ebits).toInts()
        assert got == [pow(x, e, p) for x, e in zip(xs, es)], (bits, ebits)


def test_released_table_leaves_the_cache_and_the_base_still_works(vmn, gpu_ctx):
    """vmn_group_release_fixed: the table of a base that will not come back (a prover's h_0, released when the proof object is
    freed) leaves the group's cache; a later use of the same base rebuilds it; an unknown base is no error."""
    grp, _ = load_golden(1024)
    p, q, g = grp["p"], grp["q"], grp["g"]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    es = pyref.stream_ints(b"release/e", 200, q)
    E = G.ringArray(es)
    b1, b2 = (pow(g, k, p) for k in pyref.stream_ints(b"release/b", 2, q))
    empty = G.tableBytes()
    assert G.exp(b1, E).toInts() == pyref.exp_fixed(b1, es, p)
    one = G.tableBytes()
    assert one > empty
    assert G.exp(b2, E).toInts() == pyref.exp_fixed(b2, es, p)
    assert G.tableBytes() > one
    G.releaseFixed(b2)
    assert G.tableBytes() == one
    G.releaseFixed(b2)                                       # unknown by now
    G.releaseFixed(pow(g, 12345, p))                         # never seen
    assert G.tableBytes() == one
    assert G.exp(b2, E).toInts() == pyref.exp_fixed(b2, es, p)
    G.releaseFixed(b1)
    G.releaseFixed(b2)
    assert G.tableBytes() == empty


def test_a_freed_prover_returns_the_table_of_its_base(vmn, gpu_ctx, entry):
    """The per-proof base h_0 of a PoS prover gets a fixed-base table (N exponents in commit); freeing the proof object
    releases it, so a session of proofs on fresh generators holds one such table at a time, not one per proof."""
    import mirror
    from tape import Tape
    nat = mirror.load(entry, ("native",))["native"]
    grp, _ = load_golden(1024)
    p, q, g = grp["p"], grp["q"], grp["g"]
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    n, NV, NE, NR = 64, 128, 128, 64
    seen = []
    y = pow(g, Tape(b"release-key", q).ring_element(), p)      # one key for the session: its table stays, like g's
    for k in range(3):
        t = Tape(b"release-prover-%d" % k, q)
        h = [pow(g, x, p) for x in t.ring_array(n)]
        w = [[pow(g, x, p) for x in t.ring_array(n)] for _ in range(2)]
        pi = t.permutation(n)
        s = [t.ring_array(n)]
        H, W, S = G.toElementArray(h), [G.toElementArray(c) for c in w], [G.ringArray(c) for c in s]
        pkey = [g, y]
        WP = nat.reencrypt_native(G, pkey, W, S, pi)
        pr = nat.PoSBasicTW(G, NV, NE, NR, rand=Tape(b"prover-%d" % k, q))
        pr.precompute(g, H, pi)
        pr.setInstance(pkey, W, WP, S)
        pr.setBatchVector(t.int_array(n, NE))
        pr.commit()
        during = G.tableBytes()
        pr.free()
        after = G.tableBytes()
        assert after < during, (k, during, after)
        seen.append(after)
    assert seen[0] == seen[1] == seen[2]                     # nothing of a freed prover is left behind


def test_array_pool_levels_off_and_nothing_stays_live(vmn, groups):
    """Temporaries whose size depends on the data (the level buffers of a multi-exponentiation) land in size classes,
    so repeated calls on different data reuse the cached blocks instead of adding new ones; when every array of a
    context has been freed, no allocation is live (tools/soak.py found the growth this guards against)."""
    G, grp, _ = groups[1024]
    p, q, g = grp["p"], grp["q"], grp["g"]
    ctx = G.ctx
    n = 20000
    base = ctx.memory_stats()["live_bytes"]
    blocks = []
    for it in range(14):
        es = pyref.stream_ints(b"pool%d" % it, n, q)
        E = G.ringArray(es)
        X = G.exp(g, E)
        _ = X.expProd(E)
        X.free()
        E.free()
        st = ctx.memory_stats()
        assert st["live_bytes"] == base
        blocks.append(st["pool_blocks"])
    # a data-dependent size may straddle a class boundary now and then; without classes every call adds a block
    assert blocks[-1] <= blocks[2] + 3, blocks
    # device-expanded draws wider than the modulus go through several temporaries of one object (a soak run showed
    # 13 MB per proof staying live: the first row buffer of every two-part draw was never handed back)
    seed = bytes(range(32))
    for bits in (q.bit_length() + 100, 3 * q.bit_length() + 7, 64):
        R = G.ringArrayFromPRG(seed, n, bits)
        R.free()
        assert ctx.memory_stats()["live_bytes"] == base, bits


@pytest.mark.parametrize("bits", [2048, 3072, 4096])
def test_worst_case_column_magnitudes(bits, vmn, gpu_ctx):
    """Lazy 64-bit columns must not overflow for the largest limbs: an all-ones odd modulus N = 2^bits - 1 (Montgomery
    arithmetic needs no primality), operands N - 1, N - 2 and all-ones patterns, products and powers against Python.
    4096 bits exercises the mid-product column relief of the four-lane geometry."""
    N = (1 << bits) - 1
    G = vmn.ModPGroup(gpu_ctx, N, N, 3, nbytes=bits // 8)
    vals = [N - 1, N - 2, (1 << (bits - 1)) - 1, N >> 1, (N // 3) | 1, 1, 2, N - (1 << 28)] + \
           [v | 1 for v in pyref.stream_ints(b"worst%d" % bits, 24, N)]
    X = G.toElementArray(vals)
    Y = G.toElementArray(list(reversed(vals)))
    assert X.mul(Y).toInts() == [a * b % N for a, b in zip(vals, reversed(vals))]
    assert X.mul(X).toInts() == [a * a % N for a in vals]
    es = [N - 1, N - 2, (1 << bits) - (1 << 64) - 1, 1, 0, 2, 3, (1 << (bits - 1)) + 1] + pyref.stream_ints(b"worst-e%d" % bits, 24, N)
    assert X.exp(G.ringArray(es)).toInts() == [pow(a, e, N) for a, e in zip(vals, es)]
    assert G.exp(N - 1, G.ringArray(es)).toInts() == [pow(N - 1, e, N) for e in es]
    E = G.ringArray(es)
    F = G.ringArray(list(reversed(es)))
    assert E.mul(F).toInts() == [a * b % N for a, b in zip(es, reversed(es))]
    assert X.prod() == pyref.prod(vals, N)


@pytest.mark.parametrize("bits", [512, 1024, 2048, 3072, 4096])
def test_subgroup_membership_by_jacobi_symbol(bits, groups):
    """K10: x is in the order-q subgroup of a safe-prime group iff (x / p) = 1.  The Jacobi kernels (one element per
    lane up to 2048 bits, the element's own two / four lanes at 3072 / 4096 bits) against Python on residues, non-residues
    and special values, one element at a time and inside large arrays."""
    G, grp, _ = groups[bits]
    p, q, g = grp["p"], grp["q"], grp["g"]
    rnd = pyref.stream_ints(b"jacobi%d" % bits, 40, p)
    vals = [1, 4, p - 1, p - 4, 2, p - 2, 3, (p - 1) // 2, (p + 1) // 2, 1 << 28, (1 << 56) + 1, (1 << (bits - 2)), 9] + [1 + v % (p - 1) for v in rnd]
    seen = set()
    for x in vals:
        want = pow(x, q, p) == 1
        seen.add(want)
        assert G.toElementArray([x]).isMember() is want, hex(x)
    assert seen == {True, False}
    members = [pow(v, 2, p) for v in pyref.stream_ints(b"jac-members%d" % bits, 3000, p) if v % p]
    assert G.toElementArray(members).isMember()
    nonres = next(x for x in range(2, 50) if pow(x, q, p) != 1)
    for pos in (0, 1234, len(members) - 1):
        tampered = list(members)
        tampered[pos] = tampered[pos] * nonres % p
        assert not G.toElementArray(tampered).isMember(), pos


@pytest.mark.parametrize("bits", [6144, 8192])
def test_rfc3526_groups_17_and_18(bits, vmn, gpu_ctx, oracle_for):
    """Moduli above 4096 bits (eight lanes per element): the RFC 3526 groups of 6144 and 8192 bits, array operations against
    the GMP oracle (CPython's pow takes 0.6 s per 8192-bit power), membership by the multi-lane Jacobi kernel."""
    p, q, g = pyref.modp_group(bits)
    orc = oracle_for(p, q)
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    n = 24
    es = pyref.stream_ints(b"rfc%d/e" % bits, n, q)
    fs = pyref.stream_ints(b"rfc%d/f" % bits, n, 1 << 613)
    X = G.exp(g, G.ringArray(es))
    xs = orc.exp_fixed(g, es)
    assert X.toInts() == xs and X.isMember()
    assert X.exp(G.ringArray(es[::-1])).toInts() == orc.exp_array(xs, es[::-1])
    assert X.exp(fs[0]).toInts() == orc.exp_scalar(xs, fs[0])
    assert X.mul(G.toElementArray(xs[::-1])).toInts() == orc.mul(xs, xs[::-1])
    assert X.expProd(G.ringArray(fs)) == orc.exp_prod(xs, fs, pippenger_c=6)
    assert X.expProd(G.ringArray(es)) == orc.exp_prod(xs, es, pippenger_c=6)
    assert X.prod() == orc.prod(xs)
    E, F = G.ringArray(es), G.ringArray(fs)
    x, d = E.recLin(F)
    want, last = pyref.rec_lin(es, fs, q)
    assert x.toInts() == want and d == last
    assert F.prods().toInts() == pyref.prods(fs, q)
    assert not G.toElementArray([xs[0], p - 1, xs[1]]).isMember()
    assert G.toElementArrayFromByteTree(X.toByteTree()).toInts() == xs


@pytest.mark.parametrize("bits", [8192, 16384])
def test_worst_case_column_magnitudes_above_4096_bits(bits, vmn, gpu_ctx, oracle_for):
    """The reliefs of the 8- and 16-lane products (every 74 rows; every two shares in a squaring): all-ones modulus and operands."""
    N = (1 << bits) - 1
    orc = oracle_for(N, N)
    G = vmn.ModPGroup(gpu_ctx, N, N, 3, nbytes=bits // 8)
    vals = [N - 1, N - 2, (1 << (bits - 1)) - 1, N >> 1, (N // 3) | 1, 1, 2, N - (1 << 28)] + [v | 1 for v in pyref.stream_ints(b"worst%d" % bits, 4, N)]
    X = G.toElementArray(vals)
    assert X.mul(G.toElementArray(vals[::-1])).toInts() == [a * b % N for a, b in zip(vals, vals[::-1])]
    assert X.mul(X).toInts() == [a * a % N for a in vals]
    es = [N - 1, N - 2, (1 << bits) - (1 << 64) - 1, 1, 0, 2, 3, (1 << (bits - 1)) + 1] + pyref.stream_ints(b"worst-e%d" % bits, 4, N)
    assert X.exp(G.ringArray(es)).toInts() == orc.exp_array(vals, es)
    assert G.exp(N - 1, G.ringArray(es)).toInts() == orc.exp_fixed(N - 1, es)


@pytest.mark.parametrize("bits", [1024, 2048, 3072, 4096])
def test_simultaneous_power_and_array_inverse(bits, groups, oracle_for):
    """vmn_garray_exp2 (x^e y^f with shared squarings) and vmn_garray_inv against the oracle: exponent pairs of unequal and
    equal length, zero digits and windows, e = 0, a shared exponent longer than the per-element ones."""
    G, grp, _ = groups[bits]
    p, q = grp["p"], grp["q"]
    orc = oracle_for(p, q)
    n = 70
    xs, fs = _inputs(b"exp2-%d" % bits, n, p, q)
    ys = [pow(x, 3, p) for x in xs[::-1]]
    X, Y = G.toElementArray(xs), G.toElementArray(ys)
    inv = X.inv().toInts()
    assert inv == [pow(x, -1, p) for x in xs]
    for e, fbits in ((pyref.stream_ints(b"exp2/e", 1, 1 << 256)[0], 613), (0, 613), (1, 1), ((1 << 255) + 1, 40), (q - 1, q.bit_length())):
        f = [v % (1 << fbits) for v in fs]
        f[0], f[1] = 0, (1 << (fbits - 1)) if fbits > 1 else 1
        got = X.exp2(e, Y, G.ringArray(f), fbits).toInts()
        want = orc.mul(orc.exp_scalar(xs, e), orc.exp_array(ys, f))
        assert got == want, (e.bit_length(), fbits)


@pytest.mark.parametrize("bits,n", [(2048, 600), (3072, 300)])
def test_simultaneous_power_in_phases_is_the_same_power(bits, n, vmn, gpu_ctx, oracle_for, monkeypatch):
    """k_modpow2_phased (arrays of more than one round of tiles; here a "device" of two workgroup slots): the simultaneous
    power handed from workgroup to workgroup between runs of windows, against the GMP oracle -- the shapes of a verifier's
    check (B): a shared 256-bit exponent with per-element 612-bit ones, and unequal window counts the other way round."""
    grp, _ = load_golden(bits) if bits == 2048 else (None, None)
    p, q, g = (grp["p"], grp["q"], grp["g"]) if grp else pyref.modp_group(bits)
    orc = oracle_for(p, q)
    monkeypatch.setenv("VMN_MODPOW_MAX_BLOCKS", "2")
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    xs, fs = _inputs(b"exp2-phased-%d" % bits, n, p, q)
    ys = [pow(x, 5, p) for x in xs[::-1]]
    X, Y = G.toElementArray(xs), G.toElementArray(ys)
    for e, fbits in ((pyref.stream_ints(b"exp2p/e", 1, 1 << 256)[0] | (1 << 255), 612), ((1 << 700) + 12345, 41), (0, 300)):
        f = [v % (1 << fbits) for v in fs]
        f[0], f[-1] = 0, (1 << fbits) - 1
        got = X.exp2(e, Y, G.ringArray(f), fbits).toInts()
        assert got == orc.mul(orc.exp_scalar(xs, e), orc.exp_array(ys, f)), (bits, e.bit_length(), fbits)


def test_inner_products_in_one_round_trip(vmn, groups):
    """vmn_rarray_inner_products against Python: products and plain sums mixed, arrays of different lengths (1 element, one
    beyond a reduction pass, empty)."""
    G, grp, _ = groups[2048]
    q = grp["q"]
    vals = pyref.stream_ints(b"inner-products", 3000, q)
    cases = [(vals[:1000], vals[1000:2000]), (vals[:1000], None), (vals[2000:2001], vals[5:6]), (vals[:777], vals[1:778]), ([], None), ([], [])]
    pairs = [(G.ringArray(x), G.ringArray(y) if y is not None else None) for x, y in cases]
    want = [sum(a * b for a, b in zip(x, y)) % q if y is not None else sum(x) % q for x, y in cases]
    assert vmn.innerProducts(pairs) == want
    assert vmn.innerProducts(pairs[:1]) == want[:1]


@pytest.mark.parametrize("kind", ["modp2048", "modp3072", "P-256"])
def test_multi_exponentiation_in_two_halves(kind, vmn, gpu_ctx, groups):
    """vmn_garray_expprod_multi_begin / vmn_pending_finish: the products are those of the one-call form (itself checked against
    the oracle above) -- with other device work queued in between, a second begin while one is in flight (computed at once),
    an abandoned handle, an empty array."""
    if kind == "P-256":
        from oracle.pyref_ec import Curve
        G, q = vmn.ECqPGroup(gpu_ctx, "P-256"), Curve("P-256").n
        base = Curve("P-256").g
    else:
        G, grp, _ = groups[int(kind[4:])]
        q, base = grp["q"], grp["g"]
    n = 333
    draws = pyref.stream_ints(b"halves-" + kind.encode(), 4 * n, q)
    es = [v >> (0 if i % 3 else 200) for i, v in enumerate(draws[:n])]
    E = G.ringArray(es)
    X = [G.exp(base, G.ringArray(draws[(a + 1) * n:(a + 2) * n])) for a in range(3)]
    want = vmn.expProdMulti(X, E)
    want1 = vmn.expProdMulti(X[:1], E)
    first = vmn.PendingExpProd(X, E)
    busy = X[0].mul(X[1]).exp(E)                        # device work behind the first half
    second = vmn.PendingExpProd(X[:1], E)               # a second one in flight: computed at once
    assert second.finish() == want1
    assert first.finish() == want
    assert busy.equals(X[0].exp(E).mul(X[1].exp(E)))
    dropped = vmn.PendingExpProd(X, E)                  # abandoned: its landing buffer is free again afterwards
    del dropped
    again = vmn.PendingExpProd(X, E)
    assert again.finish() == want
    empty = vmn.PendingExpProd([G.toElementArray([])], G.ringArray([]))
    assert empty.finish() == vmn.expProdMulti([G.toElementArray([])], G.ringArray([]))
    with pytest.raises(RuntimeError):
        again.finish()


@pytest.mark.parametrize("bits", [1024, 2048, 3072, 4096])
def test_two_powers_in_one_launch(bits, groups, oracle_for):
    """vmn_garray_exp_pair (k_modpow_jobs: x^e and y^f side by side in one grid) against the oracle: arrays of equal and of
    different length (either job the longer one, ragged last tiles), e = 0 and 1, short and full-length exponents."""
    G, grp, _ = groups[bits]
    p, q = grp["p"], grp["q"]
    orc = oracle_for(p, q)
    xs, fs = _inputs(b"pair-%d" % bits, 150, p, q)
    ys = [pow(x, 5, p) for x in xs[::-1]]
    e256 = pyref.stream_ints(b"pair/e", 1, 1 << 256)[0]
    for nx, ny, e, fbits in ((70, 70, e256, 613), (150, 3, e256, 613), (1, 150, e256, 40), (65, 129, 0, 613), (64, 64, 1, 1),
                             (33, 90, q - 1, q.bit_length())):
        f = [v % (1 << fbits) for v in fs[:ny]]
        f[0] = 0
        if ny > 1:
            f[1] = (1 << (fbits - 1)) if fbits > 1 else 1
        gx, gy = G.toElementArray(xs[:nx]).expPair(e, G.toElementArray(ys[:ny]), G.ringArray(f), fbits)
        assert gx.toInts() == orc.exp_scalar(xs[:nx], e), (nx, ny, e.bit_length(), fbits)
        assert gy.toInts() == orc.exp_array(ys[:ny], f), (nx, ny, e.bit_length(), fbits)
    if bits == 2048:
        # above the eight-lane threshold the two jobs run in DIFFERENT geometries (k_modpow_jobs_mixed: the longer chain eight
        # lanes per element, the shorter four): either job the longer one, ragged tiles in both
        big = 7001
        xs, fs = _inputs(b"pair-mixed", big, p, q)
        ys = xs[::-1]
        for nx, ny, e, fbits in ((6500, big, e256, 613), (big, 6300, q - 1, 100)):
            f = [v % (1 << fbits) for v in fs[:ny]]
            gx, gy = G.toElementArray(xs[:nx]).expPair(e, G.toElementArray(ys[:ny]), G.ringArray(f), fbits)
            assert gx.toInts() == orc.exp_scalar(xs[:nx], e), (nx, ny)
            assert gy.toInts() == orc.exp_array(ys[:ny], f), (nx, ny)


@pytest.mark.parametrize("bits", [2048, 3072, 4096])
def test_one_exponent_for_the_whole_array_sliding_window(bits, vmn, gpu_ctx, oracle_for, monkeypatch):
    """K1b (csrc/modp_shared_exp.h): X.exp(e) with ONE exponent walks a host-made sliding-window schedule.  Exponents that
    stress the schedule -- a single window, powers of two (trailing zeros only), all ones, alternating runs, the group order
    minus one, a full-length secret like a decryption share -- in the base and the wide geometries, against GMP and against
    the fixed-window kernel (VMN_SLIDING_WINDOW=0)."""
    p, q, g = pyref.modp_group(bits)
    orc = oracle_for(p, q)
    G = vmn.ModPGroup(gpu_ctx, p, q, g)
    n = 300
    xs = [pow(1 + v % (p - 1), 2, p) for v in pyref.stream_ints(b"slide/x%d" % bits, n, p)]
    xs[0], xs[1] = 1, p - 1
    X = G.toElementArray(xs, checked=False)
    full = pyref.stream_ints(b"slide/e%d" % bits, 1, q)[0] | (1 << (q.bit_length() - 2))
    exps = [1 << 33, (1 << 33) + 1, 1 << 200, (1 << 127) - 1, (1 << 129) - 1, int("10" * 150, 2), int("1100" * 90, 2) << 7,
            q - 1, q - 2, full, (1 << (bits - 2)) + 1]
    for e in exps:
        want = orc.exp_scalar(xs, e)
        assert X.exp(e).toInts() == want, hex(e)[:20]
    monkeypatch.setenv("VMN_SLIDING_WINDOW", "0")
    # (the knob is read per call) the fixed-window kernel gives the same array
    assert X.exp(full).toInts() == orc.exp_scalar(xs, full)
    monkeypatch.delenv("VMN_SLIDING_WINDOW")
    small = G.toElementArray(xs[:7], checked=False)            # a handful of elements: the widest geometry
    assert small.exp(full).toInts() == orc.exp_scalar(xs[:7], full)
    # arrays of more than one round of tiles walk the schedule in phases (k_modpow_shared_phased); on a "device" of ONE
    # workgroup slot this array is one: a full-length secret, a schedule of a few steps, a single window
    monkeypatch.setenv("VMN_MODPOW_MAX_BLOCKS", "1")
    for e in (full, q - 1, (1 << 129) - 1, 1 << 200, 1 << 33):
        assert X.exp(e).toInts() == orc.exp_scalar(xs, e), hex(e)[:20]
