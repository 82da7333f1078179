"""parallel.py — what the sharded proof drivers need from Python: the shard bounds and the small-object exchange.

One process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo``
in the CPU tests).  Distribution:

  * every position-indexed array (h, u, w, w', B, B', k_B, k_E, r, s …) is split into contiguous
    shards ``[lo, hi)``; rank k computes and keeps shard k;
  * the *public inputs* (generators h, input ciphertexts w, batching vector e) are replicated on
    every GPU (HBM is plentiful: 1 M x 2048-bit = 304 MB per array), so the permuted arrays
    ``u = permute(h g^r, pi)`` and ``w' = permute(w pk^s, pi^-1)`` are local gathers of replicated data —
    no all-to-all of 304-byte elements;
  * the prover's random tape is shared (same seed on every rank; each rank slices what it needs);
  * the only exchanges are tiny all-gathers: one partial product per ``expProd``/``prod`` (G x 256 B),
    one partial sum per inner product, the carries (E_tot, X_tot) of the two scans, each shard's last
    B element, and the verdict bits.  Modular multiplication is not an RCCL reduction operator, so
    "all-reduce the products" is all-gather + local multiplication.

The sharded drivers themselves are C++ (``vmn_pos_set_comm`` / ``vmn_ccpos_set_comm``, csrc/vmnproofs.cpp; Python side:
``native.PoSBasicTW.setComm``); they call back into ``Comm.all_gather_bytes`` for every exchange.
"""
from __future__ import annotations

import sys
from typing import List, Optional, Sequence


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous shard [lo, hi) of n positions for ``rank`` of ``world`` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class Comm:
    """Small-object exchange between the ranks of one node.

    ``device`` given (the GPU box): the all-gather is RCCL's (``torch.distributed`` backend ``nccl``) on this rank's GPU.
    The payloads are the few hundred bytes the proof drivers finish on the HOST (the Horner tails of the
    multi-exponentiations, inner products read back): they are staged through page-locked send / receive buffers that
    live as long as the communicator -- one async copy in, the collective, one async copy out, ONE stream
    synchronisation; no pageable copy, no allocation per call.  ``fallback`` (a gloo process group, optional): the
    constructor -- called by every rank together -- then runs ONE probe exchange over RCCL and the ranks AGREE over the
    gloo group whether it worked for all of them; if it failed anywhere every rank switches to gloo together and says
    so (``fell_back``).  After the probe there is no rank-local switching: an exchange that raises is re-raised, so that
    all ranks leave the sharded leg the same way instead of issuing collectives on different groups.
    ``device`` None: gloo on host tensors (CPU tests, rehearsals)."""

    def __init__(self, dist=None, device=None, fallback=None):
        self.dist = dist if (dist is not None and dist.is_available() and dist.is_initialized()) else None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1
        self.device = device                      # torch device for nccl; None for gloo / single process
        self.fallback = fallback
        self.fell_back = None                     # why the RCCL path was abandoned (None: it was not)
        self.backend_used = None                  # set by the first exchange: "nccl", "gloo", "gloo (fallback)"
        self._cap = 0
        self._send_pin = self._recv_pin = self._send_dev = self._recv_dev = None
        if self.dist and self.device is not None and self.fallback is not None:
            self._probe()

    def _probe(self):
        """One RCCL exchange, then a collective decision over the gloo group: RCCL for everybody or for nobody."""
        import torch
        why = None
        try:
            got = self._gather_device(bytes([self.rank & 0xFF]) * 8)
            if [got[k * 8] for k in range(self.world)] != [k & 0xFF for k in range(self.world)]:
                why = "probe exchange returned the wrong bytes"
        except Exception as exc:
            why = f"{type(exc).__name__}: {exc}"[:300]
        failed = torch.tensor([0 if why is None else 1], dtype=torch.int32)
        self.dist.all_reduce(failed, op=self.dist.ReduceOp.MAX, group=self.fallback)
        if int(failed[0]):
            self.fell_back = why or "the probe exchange failed on another rank"
            print(f"parallel.Comm: RCCL probe failed ({self.fell_back}); every rank continues over gloo", file=sys.stderr)

    def _stage(self, nbytes: int):
        import torch
        if nbytes > self._cap:
            cap = max(4096, 1 << (nbytes - 1).bit_length())
            self._send_pin = torch.empty(cap, dtype=torch.uint8).pin_memory()
            self._recv_pin = torch.empty(cap * self.world, dtype=torch.uint8).pin_memory()
            self._send_dev = torch.empty(cap, dtype=torch.uint8, device=self.device)
            self._recv_dev = torch.empty(cap * self.world, dtype=torch.uint8, device=self.device)
            self._cap = cap

    def _gather_device(self, data: bytes) -> bytes:
        import torch
        n = len(data)
        self._stage(n)
        self._send_pin[:n].copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
        self._send_dev[:n].copy_(self._send_pin[:n], non_blocking=True)
        self.dist.all_gather_into_tensor(self._recv_dev[:n * self.world], self._send_dev[:n])
        self._recv_pin[:n * self.world].copy_(self._recv_dev[:n * self.world], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return self._recv_pin[:n * self.world].numpy().tobytes()

    def _gather_host(self, data: bytes, group=None) -> bytes:
        import torch
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        out = torch.empty(self.world * len(data), dtype=torch.uint8)
        self.dist.all_gather_into_tensor(out, t, group=group)
        return out.numpy().tobytes()

    def all_gather_bytes(self, data: bytes) -> List[bytes]:
        """Every rank contributes ``len(data)`` bytes (the same length everywhere); the per-rank blocks in rank order.
        This is the one primitive the C++ proof drivers call back for (``vmn_comm.all_gather``, include/vmnproofs.h)."""
        if not self.dist:
            return [bytes(data)]
        n = len(data)
        if n == 0:
            return [b""] * self.world
        if self.device is not None and self.fell_back is None:
            raw = self._gather_device(data)       # raises on failure: no rank-local switch of the process group
            self.backend_used = "nccl"
        elif self.device is not None:
            raw = self._gather_host(data, self.fallback)
            self.backend_used = "gloo (fallback)"
        else:
            raw = self._gather_host(data)
            self.backend_used = "gloo"
        return [raw[k * n:(k + 1) * n] for k in range(self.world)]

    def all_gather_ints(self, values: Sequence[int], nbytes: int) -> List[List[int]]:
        """Every rank contributes ``len(values)`` non-negative integers (< 2^(8 nbytes)); returns the per-rank lists in
        rank order (one fixed-size all-gather)."""
        parts = self.all_gather_bytes(b"".join(int(v).to_bytes(nbytes, "big") for v in values))
        return [[int.from_bytes(raw[i * nbytes:(i + 1) * nbytes], "big") for i in range(len(values))] for raw in parts]

    def all_true(self, flag: bool) -> bool:
        return all(v[0] == 1 for v in self.all_gather_ints([1 if flag else 0], 1))

    def max_over_ranks(self, x: float) -> float:
        """The slowest rank's value (timings of a sharded leg)."""
        import struct
        return max(struct.unpack(">d", b)[0] for b in self.all_gather_bytes(struct.pack(">d", float(x))))
