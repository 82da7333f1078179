"""ctypes binding of oracle/libvmnoracle.so (C + GMP).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvmnoracle.so")


def build() -> str:
    """(Re)build the oracle with gcc + GMP; returns the library path."""
    subprocess.run(["make", "-C", _HERE, "libvmnoracle.so"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return LIB_PATH


class Oracle:
    """Array-level CPU oracle on big-endian fixed-width byte strings (the reference's wire format)."""

    def __init__(self, p: int, q: int, nbytes: int | None = None):
        if not os.path.exists(LIB_PATH):
            build()
        self.lib = C.CDLL(LIB_PATH)
        self.p, self.q = p, q
        self.nb = nbytes or (p.bit_length() + 7) // 8
        self.p_be = p.to_bytes(self.nb, "big")
        self.q_be = q.to_bytes(self.nb, "big")

    # -- helpers ---------------------------------------------------------------------------------
    def enc(self, xs: Sequence[int], nb: int | None = None) -> bytes:
        nb = nb or self.nb
        return b"".join(int(x).to_bytes(nb, "big") for x in xs)

    def dec(self, buf: bytes, nb: int | None = None) -> List[int]:
        nb = nb or self.nb
        return [int.from_bytes(buf[i:i + nb], "big") for i in range(0, len(buf), nb)]

    @property
    def threads(self) -> int:
        return self.lib.orc_num_threads()

    def set_threads(self, t: int) -> None:
        self.lib.orc_set_threads(C.c_int(t))

    # -- byte-level entry points (used by the benchmark: no Python integer conversion in the timed part)
    def exp_array_bytes(self, x: bytes, e: bytes, n: int, eb: int) -> bytes:
        out = C.create_string_buffer(max(1, n * self.nb))
        self.lib.orc_exp_array(out, x, e, C.c_size_t(n), C.c_size_t(self.nb), C.c_size_t(eb), self.p_be)
        return out.raw[: n * self.nb]

    def exp_fixed_bytes(self, base: bytes, e: bytes, n: int, eb: int) -> bytes:
        out = C.create_string_buffer(max(1, n * self.nb))
        self.lib.orc_exp_fixed(out, base, e, C.c_size_t(n), C.c_size_t(self.nb), C.c_size_t(eb), self.p_be)
        return out.raw[: n * self.nb]

    def exp_fixed_table_bytes(self, base: bytes, e: bytes, n: int, eb: int, ebits: int, w: int) -> bytes:
        out = C.create_string_buffer(max(1, n * self.nb))
        self.lib.orc_exp_fixed_table(out, base, e, C.c_size_t(n), C.c_size_t(self.nb), C.c_size_t(eb), C.c_int(ebits), C.c_int(w),
                                     self.p_be)
        return out.raw[: n * self.nb]

    def mul_bytes(self, x: bytes, y: bytes, n: int) -> bytes:
        out = C.create_string_buffer(max(1, n * self.nb))
        self.lib.orc_mul(out, x, y, C.c_size_t(n), C.c_size_t(self.nb), self.p_be)
        return out.raw[: n * self.nb]

    def expprod_pippenger_bytes(self, x: bytes, e: bytes, n: int, eb: int, ebits: int, c: int) -> bytes:
        out = C.create_string_buffer(self.nb)
        self.lib.orc_expprod_pippenger(out, x, e, C.c_size_t(n), C.c_size_t(self.nb), C.c_size_t(eb), C.c_int(ebits),
                                       C.c_int(c), self.p_be)
        return out.raw

    # -- integer-level entry points (tests) -------------------------------------------------------
    def exp_array(self, xs, es, ebytes: int | None = None) -> List[int]:
        eb = ebytes or self.nb
        return self.dec(self.exp_array_bytes(self.enc(xs), self.enc(es, eb), len(xs), eb))

    def exp_scalar(self, xs, e: int) -> List[int]:
        eb = max(1, (e.bit_length() + 7) // 8)
        out = C.create_string_buffer(max(1, len(xs) * self.nb))
        self.lib.orc_exp_scalar(out, self.enc(xs), e.to_bytes(eb, "big"), C.c_size_t(len(xs)), C.c_size_t(self.nb),
                                C.c_size_t(eb), self.p_be)
        return self.dec(out.raw[: len(xs) * self.nb])

    def exp_fixed(self, base: int, es) -> List[int]:
        return self.dec(self.exp_fixed_bytes(base.to_bytes(self.nb, "big"), self.enc(es), len(es), self.nb))

    def exp_fixed_table(self, base: int, es, w: int = 0) -> List[int]:
        """Fixed-base exponentiation through a precomputed table (orc_exp_fixed_table); w = 0 picks the window that
        minimises table build + use for this array size."""
        ebits = max(1, self.q.bit_length())
        if not w:
            n = max(1, len(es))
            w = min(range(1, 13), key=lambda c: ((1 << c) + n) * ((ebits + c - 1) // c))
        return self.dec(self.exp_fixed_table_bytes(base.to_bytes(self.nb, "big"), self.enc(es), len(es), self.nb, ebits, w))

    def exp_prod(self, xs, es, ebits: int = 0, pippenger_c: int = 0) -> int:
        eb = (ebits + 7) // 8 if ebits else self.nb
        if pippenger_c:
            return int.from_bytes(self.expprod_pippenger_bytes(self.enc(xs), self.enc(es, eb), len(xs), eb,
                                                               ebits or 8 * eb, pippenger_c), "big")
        out = C.create_string_buffer(self.nb)
        self.lib.orc_expprod_naive(out, self.enc(xs), self.enc(es, eb), C.c_size_t(len(xs)), C.c_size_t(self.nb),
                                   C.c_size_t(eb), self.p_be)
        return int.from_bytes(out.raw, "big")

    def mul(self, xs, ys) -> List[int]:
        return self.dec(self.mul_bytes(self.enc(xs), self.enc(ys), len(xs)))

    def prod(self, xs) -> int:
        out = C.create_string_buffer(self.nb)
        self.lib.orc_prod(out, self.enc(xs), C.c_size_t(len(xs)), C.c_size_t(self.nb), self.p_be)
        return int.from_bytes(out.raw, "big")

    def ring_binary(self, xs, ys, op: int) -> List[int]:
        out = C.create_string_buffer(max(1, len(xs) * self.nb))
        self.lib.orc_ring_binary(out, self.enc(xs), self.enc(ys if ys is not None else xs), C.c_size_t(len(xs)),
                                 C.c_size_t(self.nb), C.c_int(op), self.q_be)
        return self.dec(out.raw[: len(xs) * self.nb])

    def mul_add(self, xs, v: int, ys) -> List[int]:
        out = C.create_string_buffer(max(1, len(xs) * self.nb))
        self.lib.orc_ring_mul_add(out, self.enc(xs), v.to_bytes(self.nb, "big"), self.enc(ys), C.c_size_t(len(xs)),
                                  C.c_size_t(self.nb), self.q_be)
        return self.dec(out.raw[: len(xs) * self.nb])

    def rec_lin(self, bs, es) -> List[int]:
        out = C.create_string_buffer(max(1, len(bs) * self.nb))
        self.lib.orc_ring_rec_lin(out, self.enc(bs), self.enc(es), C.c_size_t(len(bs)), C.c_size_t(self.nb), self.q_be)
        return self.dec(out.raw[: len(bs) * self.nb])

    def prods(self, es) -> List[int]:
        out = C.create_string_buffer(max(1, len(es) * self.nb))
        self.lib.orc_ring_prods(out, self.enc(es), C.c_size_t(len(es)), C.c_size_t(self.nb), self.q_be)
        return self.dec(out.raw[: len(es) * self.nb])

    def ring_reduce(self, xs, ys, what: int) -> int:
        out = C.create_string_buffer(self.nb)
        self.lib.orc_ring_reduce(out, self.enc(xs), self.enc(ys if ys is not None else xs), C.c_size_t(len(xs)),
                                 C.c_size_t(self.nb), C.c_int(what), self.q_be)
        return int.from_bytes(out.raw, "big")


class GmpAdapter:
    """The group interface of oracle/pyref_proofs.py (GPoS / GCCPoS) on top of the C + GMP oracle: every array
    operation runs in ``libvmnoracle.so`` over all host cores (OpenMP), single elements stay Python integers.
    Used by the CPU tests and by bench.py's ``cpu_baseline`` of the mix + prove leg: the reference's op sequence
    with GMP underneath, i.e. what VCR + VMGJ bottoms out in, minus VCR's fixed-base tables (every exponentiation
    is an ``mpz_powm``)."""

    def __init__(self, orc: Oracle, pippenger_c: int = 8, fixed_tables: bool = False):
        self.o, self.p, self.q, self.one, self.c = orc, orc.p, orc.q, 1, pippenger_c
        self.fixed_tables = fixed_tables        # fixed-base exponentiations through precomputed tables (VCR / GMPMEE's fpowm)

    def mul(self, a, b):
        return a * b % self.p

    def exp(self, a, e):
        return pow(a, e % self.q, self.p)

    def inv(self, a):
        return pow(a, -1, self.p)

    def exp_fixed(self, base, es):
        return self.o.exp_fixed_table(base, es) if self.fixed_tables else self.o.exp_fixed(base, es)

    def exp_array(self, xs, es):
        return self.o.exp_array(xs, es)

    def exp_scalar(self, xs, e):
        return self.o.exp_scalar(xs, e)

    def exp_prod(self, xs, es):
        bits = max(1, max((int(e).bit_length() for e in es), default=1))
        return self.o.exp_prod(xs, es, ebits=8 * ((bits + 7) // 8), pippenger_c=self.c)

    def mul_arrays(self, xs, ys):
        return self.o.mul(xs, ys)

    def prod(self, xs):
        return self.o.prod(xs)
