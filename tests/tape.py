"""Deterministic random tape shared by the oracle and the HIP path in the proof parity tests."""
import hashlib


class Tape:
    def __init__(self, seed: bytes, q: int):
        self.seed, self.q, self.ctr = seed, q, 0

    def _bytes(self, n):
        out = bytearray()
        while len(out) < n:
            out += hashlib.sha256(self.seed + self.ctr.to_bytes(8, "big")).digest()
            self.ctr += 1
        return bytes(out[:n])

    def int_array(self, n, bits):
        nb = (bits + 7) // 8
        buf = self._bytes(n * nb)
        mask = (1 << bits) - 1
        return [int.from_bytes(buf[i * nb:(i + 1) * nb], "big") & mask for i in range(n)]

    def ring_array(self, n):
        nb = (self.q.bit_length() + 7) // 8 + 8
        buf = self._bytes(n * nb)
        return [int.from_bytes(buf[i * nb:(i + 1) * nb], "big") % self.q for i in range(n)]

    def ring_element(self):
        return self.ring_array(1)[0]

    def permutation(self, n):
        keys = self.int_array(n, 64)
        return sorted(range(n), key=lambda i: (keys[i], i))


class SeedTape(Tape):
    """A tape whose N-sized draws are 32-byte seeds (``array_seed``): the product expands them on the GPU
    (vmn_random_source.array_seed, include/vmnproofs.h); ``expanding=True`` is the oracle's side, which expands the same
    seeds with the Python PRG: ring elements = (bits(q) + rbitlen)-bit integers of PRG(seed) mod q, integers of `bits`
    bits = the PRG integers (mod q)."""

    def __init__(self, seed: bytes, q: int, rbitlen: int, expanding: bool = False):
        Tape.__init__(self, seed, q)
        self.rbitlen, self.expanding = rbitlen, expanding

    def array_seed(self) -> bytes:
        return self._bytes(32)

    def ring_array(self, n):
        if not self.expanding or n == 1:
            return Tape.ring_array(self, n)
        from oracle import pyref_prg
        return [x % self.q for x in pyref_prg.random_integers(self.array_seed(), n, self.q.bit_length() + self.rbitlen)]

    def int_array(self, n, bits):
        if not self.expanding or n == 1:
            return Tape.int_array(self, n, bits)
        from oracle import pyref_prg
        return [x % self.q for x in pyref_prg.random_integers(self.array_seed(), n, bits)]
