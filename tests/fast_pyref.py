"""Test infrastructure of the GPU suite: the four array exponentiations of oracle/pyref.py (exp_array, exp_scalar,
exp_fixed, exp_prod) routed through the C + GMP oracle (oracle/libvmnoracle.so, all host cores) instead of CPython's pow.

Same values -- the CPU suite pins the two oracles against each other and against the golden vectors
(tests/test_oracle_golden.py; this module is NOT active there) -- and about ten times faster, which keeps the transcript
tests of the 2048- to 4096-bit groups inside the round-end driver's time limit.  Everything else of the restatement (the
proofs' op sequences, scans, single elements) stays the Python text of oracle/pyref_proofs.py."""
from oracle import pyref
from oracle.cbind import Oracle

_PY = {name: getattr(pyref, name) for name in ("exp_array", "exp_scalar", "exp_fixed", "exp_prod")}
_cache = {}


def _orc(p):
    if p not in _cache:
        _cache[p] = Oracle(p, (p - 1) // 2)
    return _cache[p]


def _usable(p, n, es):
    return p.bit_length() >= 1024 and n >= 4 and all(0 <= e < p for e in es)


def replacements():
    def exp_array(xs, es, p):
        xs, es = list(xs), list(es)
        if len(xs) == len(es) and _usable(p, len(xs), es):
            return _orc(p).exp_array([x % p for x in xs], es)
        return _PY["exp_array"](xs, es, p)

    def exp_scalar(xs, e, p):
        xs = list(xs)
        return _orc(p).exp_scalar([x % p for x in xs], e) if _usable(p, len(xs), [e]) else _PY["exp_scalar"](xs, e, p)

    def exp_fixed(base, es, p):
        es = list(es)
        return _orc(p).exp_fixed(base % p, es) if _usable(p, len(es), es) else _PY["exp_fixed"](base, es, p)

    def exp_prod(xs, es, p):
        xs, es = list(xs), list(es)
        if len(xs) == len(es) and _usable(p, len(xs), es):
            return _orc(p).exp_prod([x % p for x in xs], es)
        return _PY["exp_prod"](xs, es, p)

    return {"exp_array": exp_array, "exp_scalar": exp_scalar, "exp_fixed": exp_fixed, "exp_prod": exp_prod}


def install():
    """For worker processes (tests/dist_worker.py on the GPU box): patch for the life of the process."""
    for name, fn in replacements().items():
        setattr(pyref, name, fn)
