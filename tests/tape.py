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
