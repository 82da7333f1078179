"""Integer-backed stand-in for the device arrays (same interface as verificatum-vmn_amd's
``ModPGroup`` / ``PGroupElementArray`` / ``PRingElementArray``), built on the oracle's pyref.  It lets
the CPU suite run the *host logic* of the multi-GPU path (sharding, carries, partial-product exchange)
on gloo ranks without a GPU.  Test infrastructure only."""
from oracle import pyref


class FakeGroup:
    def __init__(self, p, q, g, nbytes=None):
        self.p, self.q, self.g = p, q, g
        self.nbytes = nbytes or (p.bit_length() + 7) // 8

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
