"""CPU suite: the GMP oracle and the Python restatement against the committed golden vectors."""
import pytest

from conftest import ints, load_golden
from oracle import pyref


@pytest.mark.parametrize("bits", [512, 1024, 2048, 3072, 4096])
def test_groups_are_safe_prime_groups(bits):
    grp, _ = load_golden(bits)
    p, q, g = grp["p"], grp["q"], grp["g"]
    assert p == 2 * q + 1 and p.bit_length() == bits
    assert pyref.is_probable_prime(p) and pyref.is_probable_prime(q)
    assert pow(g, q, p) == 1 and g not in (0, 1)
    if bits != 512:
        assert p == pyref.rfc_modp_prime(bits)


def test_rfc3526_group14_constant_matches_pi_formula():
    assert pyref.rfc_modp_prime(2048) == pyref.RFC3526_14_P


@pytest.mark.parametrize("bits", [512, 1024, 2048, 3072, 4096])
def test_c_oracle_matches_golden(bits, oracle_for):
    grp, cases = load_golden(bits)
    p, q = grp["p"], grp["q"]
    orc = oracle_for(p, q)
    seen = set()
    for c in cases:
        op = c["op"]
        seen.add(op)
        if op == "exp_array":
            assert orc.exp_array(ints(c["x"]), ints(c["e"])) == ints(c["out"])
        elif op == "exp_ints":
            eb = (c["ebits"] + 7) // 8
            assert orc.exp_array(ints(c["x"]), ints(c["e"]), ebytes=eb) == ints(c["out"])
        elif op == "exp_scalar":
            assert orc.exp_scalar(ints(c["x"]), int(c["e"], 16)) == ints(c["out"])
        elif op == "exp_fixed":
            assert orc.exp_fixed(int(c["base"], 16), ints(c["e"])) == ints(c["out"])
        elif op == "exp_prod":
            want = int(c["out"], 16)
            assert orc.exp_prod(ints(c["x"]), ints(c["e"]), c["ebits"]) == want
            for cc in (1, 3, 8):
                assert orc.exp_prod(ints(c["x"]), ints(c["e"]), c["ebits"], pippenger_c=cc) == want
        elif op == "exp_prod_ring":
            assert orc.exp_prod(ints(c["x"]), ints(c["e"])) == int(c["out"], 16)
        elif op == "mul":
            assert orc.mul(ints(c["x"]), ints(c["y"])) == ints(c["out"])
        elif op == "prod":
            assert orc.prod(ints(c["x"])) == int(c["out"], 16)
        elif op == "rec_lin":
            got = orc.rec_lin(ints(c["b"]), ints(c["e"]))
            assert got == ints(c["out"]) and got[-1] == int(c["last"], 16)
        elif op == "prods":
            assert orc.prods(ints(c["e"])) == ints(c["out"])
        elif op == "mul_add":
            assert orc.mul_add(ints(c["x"]), int(c["v"], 16), ints(c["y"])) == ints(c["out"])
        elif op == "ring_mul":
            assert orc.ring_binary(ints(c["x"]), ints(c["y"]), 0) == ints(c["out"])
        elif op == "ring_add":
            assert orc.ring_binary(ints(c["x"]), ints(c["y"]), 1) == ints(c["out"])
        elif op == "inner_product":
            assert orc.ring_reduce(ints(c["x"]), ints(c["y"]), 0) == int(c["out"], 16)
        elif op == "ring_sum":
            assert orc.ring_reduce(ints(c["x"]), None, 1) == int(c["out"], 16)
        elif op == "ring_prod":
            assert orc.ring_reduce(ints(c["x"]), None, 2) == int(c["out"], 16)
    assert {"exp_array", "exp_fixed", "exp_prod", "mul", "prod", "rec_lin", "prods"} <= seen


def test_pyref_matches_golden_small():
    grp, cases = load_golden(512)
    p, q = grp["p"], grp["q"]
    for c in cases:
        if c["op"] == "exp_array":
            assert pyref.exp_array(ints(c["x"]), ints(c["e"]), p) == ints(c["out"])
        elif c["op"] == "rec_lin":
            assert pyref.rec_lin(ints(c["b"]), ints(c["e"]), q)[0] == ints(c["out"])
        elif c["op"] == "permute":
            assert pyref.permute(ints(c["x"]), c["perm"]) == ints(c["out"])
        elif c["op"] == "shift_push":
            assert pyref.shift_push(ints(c["x"]), int(c["el"], 16)) == ints(c["out"])


def test_oracle_edge_cases(oracle_for):
    grp, _ = load_golden(512)
    p, q = grp["p"], grp["q"]
    orc = oracle_for(p, q)
    assert orc.exp_array([], []) == []
    assert orc.prod([]) == 1
    assert orc.exp_prod([], []) == 1
    assert orc.exp_array([p - 1, 1, 5], [q, 0, 0]) == [pow(p - 1, q, p), 1, 1]
    assert orc.rec_lin([], []) == []


def test_table_driven_fixed_base_power_equals_mpz_powm(oracle_for):
    """orc_exp_fixed_table (bench.py's cpu_baseline of the mix + prove leg) against orc_exp_fixed, every window size."""
    from oracle import pyref
    p, q, g = pyref.modp_group(2048)
    orc = oracle_for(p, q)
    es = pyref.stream_ints(b"fixed-table", 40, q) + [0, 1, 2, q - 1, (1 << 2046) + 1]
    want = orc.exp_fixed(g, es)
    for w in (0, 1, 4, 7, 8, 12):
        assert orc.exp_fixed_table(g, es, w) == want, w
    assert orc.exp_fixed_table(g, []) == []


def _openssl_mod_exp(base: int, e: int, m: int) -> int:
    """BN_mod_exp of libcrypto: a third implementation, independent of GMP and of CPython's pow."""
    import ctypes
    import ctypes.util
    lib = ctypes.CDLL(ctypes.util.find_library("crypto") or "libcrypto.so.3")
    for f in ("BN_new", "BN_CTX_new", "BN_bin2bn"):
        getattr(lib, f).restype = ctypes.c_void_p
    ctx = ctypes.c_void_p(lib.BN_CTX_new())

    def bn(v):
        b = v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")
        return ctypes.c_void_p(lib.BN_bin2bn(b, len(b), None))
    r = ctypes.c_void_p(lib.BN_new())
    assert lib.BN_mod_exp(r, bn(base), bn(e), bn(m), ctx) == 1
    buf = ctypes.create_string_buffer((m.bit_length() + 7) // 8 + 8)
    n = lib.BN_bn2bin(r, buf)
    return int.from_bytes(buf.raw[:n], "big")


@pytest.mark.parametrize("bits", [2048, 3072, 4096])
def test_golden_modpow_vectors_against_openssl(bits):
    """The committed exp_array / exp_fixed vectors (the headline operation) recomputed by OpenSSL's BN_mod_exp: GMP, CPython
    and OpenSSL agree, so the vectors do not depend on one library's arithmetic."""
    grp, cases = load_golden(bits)
    p = grp["p"]
    checked = 0
    for c in cases:
        if c["op"] == "exp_array":
            for x, e, out in list(zip(ints(c["x"]), ints(c["e"]), ints(c["out"])))[:6]:
                assert _openssl_mod_exp(x, e, p) == out
                checked += 1
        elif c["op"] == "exp_fixed":
            for e, out in list(zip(ints(c["e"]), ints(c["out"])))[:4]:
                assert _openssl_mod_exp(int(c["base"], 16), e, p) == out
                checked += 1
    assert checked >= 10
