"""fiatshamir.py — the Fiat-Shamir hashing around a proof, overlapped with the GPU work that does not depend on it.

Mirrors
  * ``ChallengerRO.challenge`` — RO_nout(globalPrefix || bytetree(data)),
    ref: src/java/com/verificatum/protocol/hvzk/ChallengerRO.java:96-116;
  * the two challenges of ``PoSTW`` / ``PoSCTW`` / ``CCPoSW``: the seed of the batching vector
    = challenge(node(g, h, u, pkey, w, w'), 8 * prg.minNoSeedBytes()) and the challenge
    v = challenge(node(leaf(seed), commitment), vbitlen), ref: hvzk/PoSTW.java:118-130, 146-151 (prover), 215-229,
    248-254 (verifier); hvzk/CCPoSW.java:75-158.

The digest is ONE sequential SHA-2 stream over all the byte trees (1.57 GB for the instance of a width-1 shuffle of
10^6 ciphertexts over a 2048-bit group: ~0.75 s on one host core), so it cannot be made faster -- only hidden: the
``InstanceHasher`` thread is the party's helper thread (``Context.helper()``, ``vmn_ctx_helper_begin``); it frames each
array on the GPU on the helper lane's own stream, downloads it into a page-locked buffer and feeds it to hashlib
(which releases the GIL), while the protocol thread keeps the GPU busy with the work that does not need the seed
(re-encryption, the permutation commitment, ``commitPrepare()``).
"""
from __future__ import annotations

import hashlib
import queue
import threading
from typing import Optional, Sequence


def _hdr(tag: int, n: int) -> bytes:
    return bytes([tag]) + int(n).to_bytes(4, "big")


def leaf(data: bytes) -> bytes:
    return _hdr(1, len(data)) + bytes(data)


def prg_bytes(seed: bytes, nbytes: int, hashname: str = "sha256") -> bytes:
    """PRGHeuristic: H(seed || uint32_be(0)) || H(seed || uint32_be(1)) || ..."""
    out = bytearray()
    ctr = 0
    while len(out) < nbytes:
        out += hashlib.new(hashname, seed + ctr.to_bytes(4, "big")).digest()
        ctr += 1
    return bytes(out[:nbytes])


class Challenger:
    """``ChallengerRO``: a random oracle with a global prefix (ProtocolElGamal.java:659-683 builds the prefix)."""

    def __init__(self, globalPrefix: bytes, hashname: str = "sha256"):
        self.prefix, self.hashname = bytes(globalPrefix), hashname

    def start(self, nout_bits: int):
        """The running digest of RandomOracle(H, nout).getDigest() after update(globalPrefix)."""
        return hashlib.new(self.hashname, int(nout_bits).to_bytes(4, "big") + self.prefix)

    def finish(self, digest, nout_bits: int) -> bytes:
        nb = (nout_bits + 7) // 8
        out = bytearray(prg_bytes(digest.digest(), nb, self.hashname))
        if nout_bits % 8:
            out[0] &= (1 << (nout_bits % 8)) - 1
        return bytes(out)

    def challenge(self, data: bytes, nout_bits: int) -> bytes:
        d = self.start(nout_bits)
        d.update(data)
        return self.finish(d, nout_bits)


def element_tree(group, els: Sequence) -> bytes:
    """Byte tree of a (product-)group element given as its 2*width components [a_1..a_w, b_1..b_w] (a wide key, a
    ciphertext-shaped commitment): node(leaf, leaf) at width 1, node(node(w leaves), node(w leaves)) otherwise; one
    element alone is a leaf."""
    enc = [leaf(group.enc_el(e)) for e in els]
    if len(enc) == 1:
        return enc[0]
    if len(enc) == 2:
        return _hdr(0, 2) + enc[0] + enc[1]
    half = len(enc) // 2
    return _hdr(0, 2) + _hdr(0, half) + b"".join(enc[:half]) + _hdr(0, half) + b"".join(enc[half:])


class InstanceHasher(threading.Thread):
    """Streams byte strings and the byte trees of device arrays into one digest, on the context's helper lane.

    Items are queued by the protocol thread in hashing order: ``put_bytes``, ``put_array``, ``put_ciphertexts`` (a
    ciphertext array = node(first components, second components)), ``mark()`` (the arrays queued from here on were
    produced by GPU work the protocol thread has queued up to now), ``finish()`` -> the digest object.
    """

    def __init__(self, ctx, digest, max_array_bytes: int):
        super().__init__(daemon=True)
        import torch
        self.ctx, self.digest = ctx, digest
        self.buf = torch.empty(max_array_bytes, dtype=torch.uint8).pin_memory()
        self.view = memoryview(self.buf.numpy())
        self.q: "queue.Queue" = queue.Queue()
        self.error: Optional[BaseException] = None
        self.bytes_hashed = 0
        self.busy_s = 0.0
        self.start()

    # ---- protocol thread ---------------------------------------------------------------------------------------
    def put_bytes(self, data: bytes):
        self.q.put(("bytes", bytes(data)))

    def put_array(self, arr):
        self.q.put(("array", arr))

    def put_ciphertexts(self, comps: Sequence):
        """A PPGroupElementArray of width w given as 2w component arrays: node(u-part, v-part), a part being the array's
        own tree at width 1 and node(w array trees) otherwise (component-array-wise)."""
        half = len(comps) // 2
        self.put_bytes(_hdr(0, 2))
        for part in (comps[:half], comps[half:]):
            if half > 1:
                self.put_bytes(_hdr(0, half))
            for a in part:
                self.put_array(a)

    def mark(self):
        self.ctx.helper_mark()
        self.q.put(("sync",))

    def finish(self):
        self.q.put(("end",))
        self.join()
        if self.error is not None:
            raise self.error
        return self.digest

    # ---- helper thread -----------------------------------------------------------------------------------------
    def run(self):
        import time
        try:
            with self.ctx.helper() as lane:
                while True:
                    item = self.q.get()
                    if item[0] == "end":
                        break
                    t0 = time.perf_counter()
                    if item[0] == "sync":
                        lane.sync()
                    elif item[0] == "bytes":
                        self.digest.update(item[1])
                        self.bytes_hashed += len(item[1])
                    else:
                        n = item[1].toByteTreeInto(self.buf)
                        self.digest.update(self.view[:n])
                        self.bytes_hashed += n
                    self.busy_s += time.perf_counter() - t0
        except BaseException as exc:      # pragma: no cover - re-raised by finish()
            self.error = exc
            while True:                    # drain, so that the protocol thread never blocks on a dead consumer
                try:
                    if self.q.get(timeout=0.1)[0] == "end":
                        break
                except queue.Empty:
                    break


def hash_instance(hasher: InstanceHasher, group, g, h, u, pkey, w, wp):
    """Queue node(g, h, u, pkey, w, w') in the reference's order (PoSTW.java:118-125).  u / wp may be None: the caller
    queues them later (after ``hasher.mark()``) with ``put_array`` / ``put_ciphertexts``."""
    hasher.put_bytes(_hdr(0, 6) + leaf(group.enc_el(g)))
    hasher.put_array(h)
    if u is not None:
        hasher.put_array(u)
        hasher.put_bytes(element_tree(group, pkey))
        hasher.put_ciphertexts(w)
        if wp is not None:
            hasher.put_ciphertexts(wp)
