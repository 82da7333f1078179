#!/usr/bin/env python3
"""Generate the committed golden vectors (tests/golden/*.json) from Python integers only.

The reference holds no known-answer vectors for this path; these fixtures pin
the oracle (GMP) and the HIP kernels to values computed a third, independent way (CPython's
``pow``).  Inputs come from a SHA-256 counter stream with fixed seeds, plus the edge cases the
domain has: exponent 0 / 1 / q-1, base 1 / p-1, N = 1, all-equal exponents, maximum-length
exponents.  Re-run:  python tests/golden/gen_golden.py   (rewrites the JSON files in place).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyref  # noqa: E402


def hx(v):
    return format(v, "x")


def group_record(bits):
    if bits == 512:
        p = pyref.find_safe_prime(512, b"vmn-test-group-512")
    else:
        p = pyref.rfc_modp_prime(bits)
    q = (p - 1) // 2
    assert pyref.is_probable_prime(p) and pyref.is_probable_prime(q)
    g = 4
    assert pow(g, q, p) == 1 and g != 1
    return {"bits": bits, "p": hx(p), "q": hx(q), "g": hx(g)}


def subgroup_elems(seed, n, p):
    """Random elements of the order-q subgroup: squares of uniform residues."""
    return [pow(1 + v % (p - 1), 2, p) for v in pyref.stream_ints(seed, n, p)]


def cases_for(bits, sizes):
    grp = group_record(bits)
    p, q, g = int(grp["p"], 16), int(grp["q"], 16), int(grp["g"], 16)
    cases = []
    tag = f"g{bits}".encode()
    for n in sizes:
        s = tag + b"/%d/" % n
        xs = subgroup_elems(s + b"x", n, p)
        ys = subgroup_elems(s + b"y", n, p)
        es = pyref.stream_ints(s + b"e", n, q)
        fs = pyref.stream_ints(s + b"f", n, q)
        e256 = [v % (1 << 256) for v in pyref.stream_ints(s + b"e256", n, 1 << 256)]
        e612 = [v for v in pyref.stream_ints(s + b"e612", n, 1 << 612)]
        perm = sorted(range(n), key=lambda i: pyref.stream_ints(s + b"perm", n, 1 << 64)[i])
        v = pyref.stream_ints(s + b"v", 1, 1 << 256)[0]
        rho = pyref.stream_ints(s + b"rho", 1, 1 << 50)[0]
        base = pow(g, pyref.stream_ints(s + b"b", 1, q)[0], p)
        # edge cases folded into the first entries when there is room
        if n >= 5:
            es[0], es[1], es[2] = 0, 1, q - 1
            xs[3], xs[4] = 1, p - 1
            e256[0], e612[0] = 0, (1 << 612) - 1
        cases.append({"op": "exp_array", "n": n, "x": list(map(hx, xs)), "e": list(map(hx, es)),
                      "out": list(map(hx, pyref.exp_array(xs, es, p)))})
        cases.append({"op": "exp_ints", "n": n, "ebits": 612, "x": list(map(hx, xs)), "e": list(map(hx, e612)),
                      "out": list(map(hx, pyref.exp_array(xs, e612, p)))})
        cases.append({"op": "exp_scalar", "n": n, "x": list(map(hx, xs)), "e": hx(v),
                      "out": list(map(hx, pyref.exp_scalar(xs, v, p)))})
        cases.append({"op": "exp_scalar", "n": n, "x": list(map(hx, xs)), "e": hx(rho),
                      "out": list(map(hx, pyref.exp_scalar(xs, rho, p)))})
        cases.append({"op": "exp_fixed", "n": n, "base": hx(base), "e": list(map(hx, es)),
                      "out": list(map(hx, pyref.exp_fixed(base, es, p)))})
        cases.append({"op": "exp_prod", "n": n, "ebits": 256, "x": list(map(hx, xs)), "e": list(map(hx, e256)),
                      "out": hx(pyref.exp_prod(xs, e256, p))})
        cases.append({"op": "exp_prod", "n": n, "ebits": 612, "x": list(map(hx, xs)), "e": list(map(hx, e612)),
                      "out": hx(pyref.exp_prod(xs, e612, p))})
        cases.append({"op": "exp_prod_ring", "n": n, "x": list(map(hx, xs)), "e": list(map(hx, es)),
                      "out": hx(pyref.exp_prod(xs, es, p))})
        cases.append({"op": "mul", "n": n, "x": list(map(hx, xs)), "y": list(map(hx, ys)),
                      "out": list(map(hx, pyref.mul(xs, ys, p)))})
        cases.append({"op": "prod", "n": n, "x": list(map(hx, xs)), "out": hx(pyref.prod(xs, p))})
        cases.append({"op": "permute", "n": n, "x": list(map(hx, xs)), "perm": perm,
                      "out": list(map(hx, pyref.permute(xs, perm)))})
        cases.append({"op": "shift_push", "n": n, "x": list(map(hx, xs)), "el": hx(base),
                      "out": list(map(hx, pyref.shift_push(xs, base)))})
        rl, d = pyref.rec_lin(es, fs, q)
        cases.append({"op": "rec_lin", "n": n, "b": list(map(hx, es)), "e": list(map(hx, fs)),
                      "out": list(map(hx, rl)), "last": hx(d)})
        cases.append({"op": "prods", "n": n, "e": list(map(hx, fs)), "out": list(map(hx, pyref.prods(fs, q)))})
        cases.append({"op": "mul_add", "n": n, "x": list(map(hx, es)), "v": hx(v % q), "y": list(map(hx, fs)),
                      "out": list(map(hx, pyref.mul_add(es, v % q, fs, q)))})
        cases.append({"op": "ring_mul", "n": n, "x": list(map(hx, es)), "y": list(map(hx, fs)),
                      "out": list(map(hx, [a * b % q for a, b in zip(es, fs)]))})
        cases.append({"op": "ring_add", "n": n, "x": list(map(hx, es)), "y": list(map(hx, fs)),
                      "out": list(map(hx, [(a + b) % q for a, b in zip(es, fs)]))})
        cases.append({"op": "inner_product", "n": n, "x": list(map(hx, es)), "y": list(map(hx, fs)),
                      "out": hx(pyref.inner_product(es, fs, q))})
        cases.append({"op": "ring_sum", "n": n, "x": list(map(hx, es)), "out": hx(sum(es) % q)})
        cases.append({"op": "ring_prod", "n": n, "x": list(map(hx, fs)), "out": hx(pyref.prods(fs, q)[-1])})
    return {"group": grp, "cases": cases}


def main():
    plan = {512: [1, 2, 7, 65], 1024: [5, 64], 2048: [1, 2, 5, 63, 64], 3072: [1, 5, 33], 4096: [1, 5, 19]}
    for bits, sizes in plan.items():
        rec = cases_for(bits, sizes)
        path = os.path.join(HERE, f"modp{bits}.json")
        with open(path, "w") as f:
            json.dump(rec, f, separators=(",", ":"))
        print(path, len(rec["cases"]), "cases", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
