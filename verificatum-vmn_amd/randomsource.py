"""randomsource.py — the random sources a prover is handed (``vmn_random_source`` of include/vmnproofs.h on the Python side).

``SecureRandomSource`` (os.urandom) is the default of the bindings; the fixed-seed sources are named ``Insecure*`` and serve
the tests and the benchmark.  VCR's own ``RandomSource`` classes are not part of the reference tree
(demo/mixnet/.checkbaseconf:124 selects ``RandomDevice /dev/urandom``)."""
from __future__ import annotations

import hashlib
from typing import List


def inv_perm(pi):
    """Table of the inverse permutation (what ``permute`` of the library, a gather, is handed: DESIGN.md §2)."""
    try:
        import numpy as np
        if isinstance(pi, np.ndarray) or len(pi) > 4096:
            a = np.asarray(pi, dtype=np.int64)
            inv = np.empty(len(a), dtype=np.uint32)
            inv[a] = np.arange(len(a), dtype=np.uint32)
            return inv
    except ImportError:      # pragma: no cover
        pass
    inv = [0] * len(pi)
    for i, j in enumerate(pi):
        inv[j] = i
    return inv


class SecureRandomSource:
    """The random source a prover uses by default: the operating system's CSPRNG (``os.urandom``), the counterpart of
    the reference's ``RandomDevice /dev/urandom`` (demo/mixnet/.checkbaseconf:124).  Single ring elements are drawn
    with ``rbitlen`` bits of slack and reduced (statistical distance 2^-rbitlen from uniform); the N-sized draws of a
    proof are expanded on the GPU from 32 fresh bytes each (``array_seed``, see ``vmn_random_source`` in
    include/vmnproofs.h), so no N-sized host buffer exists.  ``ring_array`` / ``int_array`` remain for drivers that
    want host rows (the Python mirror)."""

    def __init__(self, q: int, rbitlen: int = 100):
        import os
        self._urandom = os.urandom
        self.q, self.rbitlen = q, rbitlen

    def array_seed(self) -> bytes:
        return self._urandom(32)

    def ring_element(self) -> int:
        nb = (self.q.bit_length() + self.rbitlen + 7) // 8
        return int.from_bytes(self._urandom(nb), "big") % self.q

    def ring_array(self, n: int) -> List[int]:
        return [self.ring_element() for _ in range(n)]

    def int_array(self, n: int, bits: int) -> List[int]:
        nb = (bits + 7) // 8
        mask = (1 << bits) - 1
        return [int.from_bytes(self._urandom(nb), "big") & mask for _ in range(n)]

    def permutation(self, n: int) -> List[int]:
        import random
        pi = list(range(n))
        random.SystemRandom().shuffle(pi)
        return pi


class InsecureShaRandomSource:
    """NOT FOR PRODUCTION: a fixed-seed, reproducible stream (tests, benchmark).
    Deterministic random source for the benchmark and the tests (SHA-256 counter stream).  VCR's
    own ``randomElementArray`` sampling is not part of the reference tree; any source works for the
    arithmetic, and parity tests feed the same tape to the oracle."""

    def __init__(self, seed: bytes, q: int):
        self.seed, self.q, self.ctr = seed, q, 0

    def _bytes(self, n: int) -> bytes:
        out = bytearray()
        while len(out) < n:
            out += hashlib.sha256(self.seed + self.ctr.to_bytes(8, "big")).digest()
            self.ctr += 1
        return bytes(out[:n])

    def int_array(self, n: int, bits: int) -> List[int]:
        nb = (bits + 7) // 8
        buf = self._bytes(n * nb)
        mask = (1 << bits) - 1
        return [int.from_bytes(buf[i * nb:(i + 1) * nb], "big") & mask for i in range(n)]

    def ring_array(self, n: int) -> List[int]:
        nb = (self.q.bit_length() + 7) // 8 + 8
        buf = self._bytes(n * nb)
        return [int.from_bytes(buf[i * nb:(i + 1) * nb], "big") % self.q for i in range(n)]

    def ring_element(self) -> int:
        return self.ring_array(1)[0]

    def permutation(self, n: int) -> List[int]:
        keys = self.int_array(n, 64)
        return sorted(range(n), key=lambda i: (keys[i], i))


class InsecureBulkRandomSource:
    """NOT FOR PRODUCTION: numpy's PCG64 is not a cryptographic generator (benchmark inputs only).
    numpy-backed random source for large N (benchmark): hands out big-endian byte blocks of the
    group's wire width instead of Python integers, so that no per-element Python work happens.
    Ring elements are uniform below 2^(bits(q)-1) <= q (top bit cleared): statistically as good for
    a throughput run, and always in range."""

    def __init__(self, seed: int, q: int, nbytes: int):
        import numpy as np
        self.np = np
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.q, self.nbytes = q, nbytes
        self.qbits = q.bit_length()

    def _block(self, n: int, bits: int) -> bytes:
        np = self.np
        a = np.zeros((n, self.nbytes), dtype=np.uint8)
        nb = (bits + 7) // 8
        a[:, self.nbytes - nb:] = self.rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
        if bits % 8:
            a[:, self.nbytes - nb] &= (1 << (bits % 8)) - 1
        return a.tobytes()

    def int_array(self, n: int, bits: int) -> bytes:
        return self._block(n, min(bits, self.qbits - 1))

    def ring_array(self, n: int) -> bytes:
        return self._block(n, self.qbits - 1)

    def ring_element(self) -> int:
        return int.from_bytes(self._block(1, self.qbits - 1), "big")

    def permutation(self, n: int):
        return self.rng.permutation(n).astype(self.np.uint32)

    def array_seed(self) -> bytes:
        """32 bytes for a device-expanded array draw (vmn_random_source.array_seed)."""
        return self.rng.integers(0, 256, size=32, dtype=self.np.uint8).tobytes()
