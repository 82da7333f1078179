"""Integer-backed stand-in for the device arrays (same interface as verificatum-vmn_amd's
``ModPGroup`` / ``PGroupElementArray`` / ``PRingElementArray``), built on the oracle's pyref.  It lets
the CPU suite run the *host logic* of the multi-GPU path (sharding, carries, partial-product exchange)
on gloo ranks without a GPU.  Test infrastructure only."""
from oracle import pyref


class FakeGroup:
    def __init__(self, p, q, g, nbytes=None):
        self.p, self.q, self.g = p, q, g
        self.nbytes = nbytes or (p.bit_length() + 7) // 8
        self.exp_bytes = self.nbytes

    def _ints(self, values):
        if isinstance(values, (bytes, bytearray)):
            nb = self.nbytes
            return [int.from_bytes(values[i:i + nb], "big") for i in range(0, len(values), nb)]
        return [int(v) for v in values]

    def toElementArray(self, values, checked=True):
        return FakeG(self, self._ints(values))

    def ringArray(self, values, checked=True):
        return FakeR(self, self._ints(values))

    def exp(self, base, exponents):
        return FakeG(self, pyref.exp_fixed(base, exponents.v, self.p))

    def mulPartials(self, partials):
        return pyref.prod(partials, self.p)

    # single-element helpers (same names as the product's group classes)
    elem_bytes = property(lambda self: self.nbytes)
    ONE = 1

    def enc_el(self, el):
        return int(el).to_bytes(self.nbytes, "big")

    def dec_el(self, buf):
        return int.from_bytes(buf, "big")

    def k_mul(self, a, b):
        return a * b % self.p

    def k_exp(self, a, e):
        return pow(a, e % self.q, self.p)

    def k_inv(self, a):
        return pow(a, -1, self.p)


class _Arr:
    def __init__(self, group, v):
        self.group, self.v = group, list(v)

    def size(self):
        return len(self.v)

    def toInts(self):
        return list(self.v)

    def free(self):
        pass

    def get(self, i):
        return self.v[i]

    def copyOfRange(self, a, b):
        return type(self)(self.group, self.v[a:b])

    def permute(self, perm):
        return type(self)(self.group, [self.v[int(j)] for j in perm])

    def extract(self, keep):
        return type(self)(self.group, [x for x, k in zip(self.v, keep) if k])

    def shiftPush(self, el):
        return type(self)(self.group, pyref.shift_push(self.v, el) if self.v else [])

    def equals(self, other):
        return self.v == other.v


class FakeG(_Arr):
    def exp(self, e, ebits=0):
        p = self.group.p
        if isinstance(e, FakeR):
            return FakeG(self.group, pyref.exp_array(self.v, e.v, p))
        return FakeG(self.group, pyref.exp_scalar(self.v, int(e), p))

    def expProd(self, e, ebits=0):
        es = e.v if isinstance(e, FakeR) else list(e)
        return pyref.exp_prod(self.v, es, self.group.p)

    def mul(self, other):
        return FakeG(self.group, pyref.mul(self.v, other.v, self.group.p))

    def prod(self):
        return pyref.prod(self.v, self.group.p)


class FakeR(_Arr):
    def mul(self, other):
        q = self.group.q
        return FakeR(self.group, [a * b % q for a, b in zip(self.v, other.v)])

    def add(self, other):
        q = self.group.q
        return FakeR(self.group, [(a + b) % q for a, b in zip(self.v, other.v)])

    def mulAdd(self, v, other):
        q = self.group.q
        if other is None:
            return FakeR(self.group, [a * v % q for a in self.v])
        return FakeR(self.group, pyref.mul_add(self.v, v, other.v, q))

    def recLin(self, e):
        x, d = pyref.rec_lin(self.v, e.v, self.group.q)
        return FakeR(self.group, x), d

    def prods(self):
        return FakeR(self.group, pyref.prods(self.v, self.group.q))

    def innerProduct(self, other):
        return pyref.inner_product(self.v, other.v, self.group.q)

    def sum(self):
        return sum(self.v) % self.group.q

    def prod(self):
        acc = 1
        for t in self.v:
            acc = acc * t % self.group.q
        return acc

    def maxBits(self):
        return max((int(t).bit_length() for t in self.v), default=0)


class FakeECGroup(FakeGroup):
    """The same stand-in over an elliptic curve (oracle/pyref_ec.Curve); elements are affine points or None."""

    def __init__(self, curve):
        self.c = curve
        self.p, self.q, self.g = curve.p, curve.n, curve.g
        self.nbytes = curve.nbytes
        self.exp_bytes = self.nbytes

    elem_bytes = property(lambda self: 2 * self.nbytes)
    ONE = None

    def enc_el(self, el):
        return self.c.enc(el)

    def dec_el(self, buf):
        return self.c.dec(buf)

    def k_mul(self, a, b):
        return self.c.add(a, b)

    def k_exp(self, a, e):
        return self.c.mul(e % self.q, a)

    def k_inv(self, a):
        return self.c.neg(a)

    def toElementArray(self, values, checked=True):
        return FakeECArr(self, list(values))

    def exp(self, base, exponents):
        return FakeECArr(self, self.c.exp_fixed(base, exponents.v))

    def mulPartials(self, partials):
        return self.c.prod(partials)


class FakeECArr(_Arr):
    def exp(self, e, ebits=0):
        c = self.group.c
        if isinstance(e, FakeR):
            return FakeECArr(self.group, c.exp_array(self.v, e.v))
        return FakeECArr(self.group, [c.mul(int(e), P) for P in self.v])

    def expProd(self, e, ebits=0):
        es = e.v if isinstance(e, FakeR) else list(e)
        return self.group.c.exp_prod(self.v, es)

    def mul(self, other):
        return FakeECArr(self.group, self.group.c.mul_arrays(self.v, other.v))

    def prod(self):
        return self.group.c.prod(self.v)
