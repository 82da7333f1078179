"""native.py — ctypes binding of the proof-level C ABI (``include/vmnproofs.h``, ``libvmnproofs.so``): the C++
drivers ``vmn_pos_* / vmn_posc_* / vmn_ccpos_*`` and the shuffler lines, under the method names of the reference
classes so that the same tests drive them and the Python mirror in ``hvzk.py`` / ``mixnet.py``.

ref: src/java/com/verificatum/protocol/hvzk/{PoSBasicTW,PoSCBasicTW,CCPoSBasicW}.java,
     mixnet/ShufflerElGamalSession.java:400-409, 273-278, mixnet/PermutationCommitment.java:189-215.

Messages are dicts with the keys of ``hvzk.py`` (arrays are views into the native message, scalars host values);
the native message travels along as the ``native`` attribute of the dict (``MsgDict``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

from . import (PGroupElementArray, PRingElementArray, VmnError, _check, _u32_array, host_block, int_to_be, lib)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvmnproofs.so")
_plib: Optional[C.CDLL] = None

GARRAY, RARRAY, ELEMENTS, RING = 1, 2, 3, 4
_ROWS_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p))
_INTS_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p))


_SEED_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8))


class _RandomSourceStruct(C.Structure):
    _fields_ = [("user", C.c_void_p), ("ring_elements", _ROWS_CB), ("integers", _INTS_CB), ("array_seed", _SEED_CB)]


_GATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class _CommStruct(C.Structure):
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("all_gather", _GATHER_CB)]


class NativeComm:
    """``vmn_comm`` over a ``parallel.Comm`` (torch.distributed: RCCL on the GPU box, gloo in the tests): the proof
    drivers call back for their few fixed-size all-gathers of scalars."""

    def __init__(self, comm):
        self.comm = comm
        self.error = None
        self.exchanges = 0
        self.bytes_sent = 0

        def gather_cb(_user, send, nbytes, recv):
            try:
                parts = comm.all_gather_bytes(C.string_at(send, nbytes))
                C.memmove(recv, b"".join(parts), nbytes * comm.world)
                self.exchanges += 1
                self.bytes_sent += nbytes
                return 0
            except Exception as exc:       # pragma: no cover - re-raised by the caller
                self.error = exc
                return 1

        self._cb = _GATHER_CB(gather_cb)
        self.struct = _CommStruct(None, comm.rank, comm.world, self._cb)


def shard_bounds_native(n: int, world: int, rank: int):
    lo, hi = C.c_size_t(), C.c_size_t()
    plib().vmn_shard_bounds(C.c_size_t(n), C.c_int(world), C.c_int(rank), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def plib() -> C.CDLL:
    """Load ``libvmnproofs.so`` (g++, built by ``__graft_entry__.build()``); it links against ``libvmnhip.so``."""
    global _plib
    if _plib is None:
        lib()                                   # the HIP library first: fails loudly when it is missing
        if not os.path.exists(LIB_PATH):
            raise VmnError(-2, f"{LIB_PATH} is missing: run __graft_entry__.build()")
        _plib = C.CDLL(LIB_PATH)
        for name in ("vmn_msg_items", "vmn_msg_bytetree_size", "vmn_pos_width", "vmn_ccpos_width"):
            getattr(_plib, name).restype = C.c_size_t
        for name in ("vmn_msg_item_garray", "vmn_msg_item_rarray", "vmn_pos_permutation_commitment"):
            getattr(_plib, name).restype = C.c_void_p
        _plib.vmn_shard_bounds.restype = None
    return _plib


class RandomSource:
    """Adapter from a tape object (``ring_array(n)``, ``ring_element()``, ``int_array(n, bits)``; ints or
    big-endian blocks) to ``vmn_random_source``.  Buffers stay alive until the next call, as the ABI requires."""

    def __init__(self, group, tape, device_arrays: bool = True):
        # the callbacks close over this small state object only (no reference back to the RandomSource or to the
        # group: a reference cycle would hand finalisation order to the garbage collector)
        class _State:
            keep = None
            error = None
        st = self._state = _State()
        st.tape = tape
        q, nbytes = group.q, group.exp_bytes

        def hand_over(vals, out):
            blk = host_block(vals)                     # a block of rows of the wire width (bulk sources): no copy
            if blk is None:
                blk = host_block(b"".join(int(x % q).to_bytes(nbytes, "big") for x in vals))
            st.keep = blk[2]                           # valid until the next call
            out[0] = blk[0].value if isinstance(blk[0], C.c_void_p) else C.cast(C.c_char_p(blk[0]), C.c_void_p).value

        def ring_cb(_user, n, out):
            try:
                hand_over([st.tape.ring_element()] if n == 1 else st.tape.ring_array(n), out)
                return 0
            except Exception as exc:       # pragma: no cover - re-raised by the caller
                st.error = exc
                return 1

        def ints_cb(_user, n, bits, out):
            try:
                hand_over(st.tape.int_array(n, bits), out)
                return 0
            except Exception as exc:       # pragma: no cover
                st.error = exc
                return 1

        def seed_cb(_user, out):
            try:
                seed = bytes(st.tape.array_seed())
                if len(seed) != 32:
                    raise ValueError("array_seed() must return 32 bytes")
                C.memmove(out, seed, 32)
                return 0
            except Exception as exc:       # pragma: no cover
                st.error = exc
                return 1

        # N-sized draws are expanded on the device when the tape offers 32-byte seeds (``array_seed()``) and
        # ``device_arrays`` is not switched off; otherwise they come as host rows
        use_seed = device_arrays and hasattr(tape, "array_seed")
        self._cbs = (_ROWS_CB(ring_cb), _INTS_CB(ints_cb), _SEED_CB(seed_cb) if use_seed else _SEED_CB())
        self.struct = _RandomSourceStruct(None, self._cbs[0], self._cbs[1], self._cbs[2])

    @property
    def error(self):
        return self._state.error


class Message:
    """Owner of a ``vmn_msg``; arrays handed out are borrowed views that keep the message alive."""

    def __init__(self, group, handle):
        self.group, self._h = group, handle

    def __del__(self):
        try:
            if self._h and self.group.alive:
                plib().vmn_msg_free(self._h)
            self._h = None
        except Exception:
            pass

    def items(self) -> int:
        return plib().vmn_msg_items(self._h)

    def item(self, i: int):
        kind = plib().vmn_msg_item_kind(self._h, C.c_size_t(i))
        if kind in (GARRAY, RARRAY):
            fn, cls = ("vmn_msg_item_garray", PGroupElementArray) if kind == GARRAY else ("vmn_msg_item_rarray", PRingElementArray)
            arr = cls(self.group, C.c_void_p(getattr(plib(), fn)(self._h, C.c_size_t(i))))
            arr._borrowed, arr._keep = True, self
            return arr
        data, count, width = C.c_void_p(), C.c_size_t(), C.c_size_t()
        _check(plib().vmn_msg_item_bytes(self._h, C.c_size_t(i), C.byref(data), C.byref(count), C.byref(width)))
        raw = C.string_at(data, count.value * width.value)
        rows = [raw[k * width.value:(k + 1) * width.value] for k in range(count.value)]
        if kind == ELEMENTS:
            return [self.group.dec_el(r) for r in rows]
        return [int.from_bytes(r, "big") for r in rows]

    def toByteTree(self) -> bytes:
        size = plib().vmn_msg_bytetree_size(self._h)
        out = C.create_string_buffer(size)
        _check(plib().vmn_msg_to_bytetree(self._h, out))
        return out.raw

    def byteTreeSize(self) -> int:
        return plib().vmn_msg_bytetree_size(self._h)

    def toByteTreeInto(self, host_buffer) -> int:
        """The byte tree written into a caller-owned host buffer (a pinned ``torch`` uint8 tensor / uint8 array): the
        arrays inside are framed on the GPU and downloaded straight into it."""
        blk = host_block(host_buffer)
        size = self.byteTreeSize()
        if blk is None or blk[1] < size:
            raise ValueError("host buffer too small for the byte tree")
        _check(plib().vmn_msg_to_bytetree(self._h, blk[0]))
        return size

    @staticmethod
    def fromByteTree(group, bt: bytes, layout: Sequence[int], counts: Sequence[int]) -> Optional["Message"]:
        """None when the bytes are not a message of that layout (the caller substitutes trivial values)."""
        h, ok = C.c_void_p(), C.c_int(0)
        lay = (C.c_int * len(layout))(*layout)
        cnt = (C.c_size_t * len(counts))(*counts)
        if isinstance(bt, tuple):                      # (host buffer object, length): no copy (pinned: PCIe speed)
            blk = host_block(bt[0])
            ptr, length = blk[0], bt[1]
        else:
            ptr, length = bytes(bt), len(bt)
        _check(plib().vmn_msg_from_bytetree(group._h, ptr, C.c_size_t(length), lay, cnt, C.c_size_t(len(layout)),
                                            C.byref(h), C.byref(ok)))
        return Message(group, h) if ok.value else None

    @staticmethod
    def build(group, parts) -> "Message":
        """parts: arrays (copied handles are NOT taken: the array is duplicated) or ('el', [..]) / ('ring', [..])."""
        h = C.c_void_p()
        _check(plib().vmn_msg_create(C.byref(h)))
        m = Message(group, h)
        for part in parts:
            if isinstance(part, PGroupElementArray):
                dup = part.copyOfRange(0, part.size())
                _check(plib().vmn_msg_push_garray(h, dup._h))
                dup._borrowed = True
            elif isinstance(part, PRingElementArray):
                dup = part.copyOfRange(0, part.size())
                _check(plib().vmn_msg_push_rarray(h, dup._h))
                dup._borrowed = True
            elif part[0] == "el":
                buf = group.enc_els(part[1])
                _check(plib().vmn_msg_push_elements(h, buf, C.c_size_t(len(part[1])), C.c_size_t(group.elem_bytes)))
            else:
                # a value that fits the wire width travels as it is (a reply may hold k + q: the verifier must say no)
                lim = 1 << (8 * group.exp_bytes)
                buf = b"".join(int_to_be(x if 0 <= x < lim else x % group.q, group.exp_bytes) for x in part[1])
                _check(plib().vmn_msg_push_ring(h, buf, C.c_size_t(len(part[1])), C.c_size_t(group.exp_bytes)))
        return m


class MsgDict(dict):
    """A message in the shape of hvzk.py's dicts, plus the native ``vmn_msg`` it was read from."""
    native: Optional[Message] = None


def _one(x):
    return x[0]


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a._h.value if isinstance(a._h, C.c_void_p) else a._h for a in arrs])


def _be(x: int) -> bytes:
    return int(x).to_bytes(max(1, (int(x).bit_length() + 7) // 8), "big")


class _NativeProof:
    _prefix = ""
    _com_keys: Sequence[str] = ()
    _rep_keys: Sequence[str] = ()
    _com_scalar = ()          # keys whose value is ONE element (not a list)
    _rep_scalar = ()

    def __init__(self, group, vbitlen: int, ebitlen: int, rbitlen: int, rand=None):
        self.G = group
        self.q = group.q
        self._rs = RandomSource(group, rand) if rand is not None else None
        self._h = C.c_void_p()
        rs = C.byref(self._rs.struct) if self._rs else None
        _check(self._fn("create")(group._h, C.c_int(vbitlen), C.c_int(ebitlen), C.c_int(rbitlen), rs, C.byref(self._h)))
        self._keep = []

    def _fn(self, name):
        return getattr(plib(), f"vmn_{self._prefix}_{name}")

    def setComm(self, ncomm: "NativeComm"):
        """Shard this proof over the ranks of ``ncomm`` (``vmn_*_set_comm``; before the instance is set): every rank
        makes the same calls, array items of the messages are this rank's shard (include/vmnproofs.h, vmn_comm)."""
        self._ncomm = ncomm
        _check(self._fn("set_comm")(self._h, C.byref(ncomm.struct)))

    def _call(self, name, *args):
        rc = self._fn(name)(self._h, *args)
        if rc != 0 and self._rs is not None and self._rs.error is not None:
            raise self._rs.error
        if rc != 0 and getattr(self, "_ncomm", None) is not None and self._ncomm.error is not None:
            raise self._ncomm.error
        _check(rc)

    def free(self):
        if self._h and self.G.alive:
            self._fn("free")(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def _msg_dict(self, h, keys, scalar_keys):
        m = Message(self.G, h)
        out = MsgDict()
        out.native = m
        for i, k in enumerate(keys):
            val = m.item(i)
            out[k] = _one(val) if k in scalar_keys else val
        return out

    def _as_msg(self, d, keys, scalar_keys, kinds) -> Message:
        """A dict produced here travels with its native message; a hand-made (e.g. tampered) one is rebuilt."""
        m = getattr(d, "native", None)
        if m is not None and all(self._same(d[k], m.item(i), k in scalar_keys) for i, k in enumerate(keys)):
            return m
        parts = []
        for k, kind in zip(keys, kinds):
            val = d[k]
            if kind in (GARRAY, RARRAY):
                parts.append(val)
            else:
                vals = [val] if k in scalar_keys else list(val)
                parts.append(("el" if kind == ELEMENTS else "ring", vals))
        return Message.build(self.G, parts)

    @staticmethod
    def _same(val, item, scalar):
        if hasattr(val, "_h"):
            return hasattr(item, "_h") and val._h.value == item._h.value
        return ([val] if scalar else list(val)) == item

    def setBatchVector(self, e_ints):
        blk = host_block(e_ints) or host_block(b"".join(int_to_be(x, self.G.exp_bytes) for x in e_ints))
        self._call("set_batch_vector", blk[0])

    def setBatchVectorSeed(self, seed: bytes):
        """``setBatchVector(byte[] prgSeed)``: e derived on the GPU from the seed (PoSBasicTW.java:533-538)."""
        self._call("set_batch_vector_seed", bytes(seed), C.c_size_t(len(seed)))

    def setChallenge(self, v: int):
        b = _be(v)
        self._call("set_challenge", b, C.c_size_t(len(b)))

    def commitPrepare(self):
        """The part of ``commit()`` that does not depend on the batching vector (``vmn_*_commit_prepare``): run it while
        the Fiat-Shamir seed is being hashed."""
        self._call("commit_prepare")

    def _commit(self):
        h = C.c_void_p()
        self._call("commit", C.byref(h))
        return self._msg_dict(h, self._com_keys, self._com_scalar)

    def reply(self, v: int):
        b = _be(v)
        h = C.c_void_p()
        self._call("reply", b, C.c_size_t(len(b)), C.byref(h))
        return self._msg_dict(h, self._rep_keys, self._rep_scalar)


class PoSBasicTW(_NativeProof):
    """``vmn_pos_*`` — ref: hvzk/PoSBasicTW.java."""
    _prefix = "pos"
    _com_keys = ("B", "Ap", "Bp", "Cp", "Dp", "Fp")
    _com_scalar = ("Ap", "Cp", "Dp")
    _com_kinds = (GARRAY, ELEMENTS, GARRAY, ELEMENTS, ELEMENTS, ELEMENTS)
    _rep_keys = ("k_A", "k_B", "k_C", "k_D", "k_E", "k_F")
    _rep_scalar = ("k_A", "k_C", "k_D")
    _rep_kinds = (RING, RARRAY, RING, RING, RARRAY, RING)

    def precompute(self, g, h, pi=None):
        self.h = h
        if pi is None:
            self._call("precompute", self.G.enc_el(g), h._h, None)
            return
        ptr, keep = _u32_array(pi)
        self._call("precompute", self.G.enc_el(g), h._h, ptr)
        # a view of the proof object's u: valid while this object lives (no back-reference: a cycle would let the
        # garbage collector finalise the group before the proof object)
        self.u = PGroupElementArray(self.G, C.c_void_p(plib().vmn_pos_permutation_commitment(self._h)))
        self.u._borrowed = True

    def setPermutationCommitment(self, u):
        self.u = u
        self._call("set_permutation_commitment", u._h)

    def setInstance(self, pkey, w, wp, s=None):
        width = len(pkey) // 2
        self._keep = [w, wp, s]
        self._call("set_instance", self.G.enc_els(pkey), C.c_size_t(width), _ptr_array(w), _ptr_array(wp),
                   _ptr_array(s) if s is not None else None)

    def commit(self):
        return self._commit()

    def computeAF(self):
        self._call("compute_af")

    def setCommitment(self, msg):
        self._com = msg if isinstance(msg, Message) else self._as_msg(msg, self._com_keys, self._com_scalar, self._com_kinds)
        self._call("set_commitment", self._com._h)

    def verifyPrepare(self, reply) -> None:
        """``vmn_pos_verify_prepare``: the part of verify() that needs the reply but not the challenge (a Message only:
        verify() recognises the same object)."""
        self._call("verify_prepare", reply._h)

    def verify(self, reply) -> bool:
        m = reply if isinstance(reply, Message) else self._as_msg(reply, self._rep_keys, self._rep_scalar, self._rep_kinds)
        verdict = C.c_int(0)
        five = (C.c_int * 5)()
        self._call("verify", m._h, C.byref(verdict), five)
        self.verdicts = tuple(bool(x) for x in five)
        return bool(verdict.value)

    def _get(self, name: str, count: int = 1):
        out = C.create_string_buffer(count * self.G.elem_bytes)
        self._call(f"get_{name}", out)
        els = self.G.dec_els(out.raw)
        return els[0] if count == 1 else els

    def getA(self):
        """``getA()`` (:716): A = u.expProd(e), after computeAF."""
        return self._get("A")

    def getF(self):
        """``getF()`` (:761): F = w.expProd(e) as its 2 * width components, after computeAF."""
        return list(self._get("F", 2 * int(plib().vmn_pos_width(self._h))))

    def getC(self):
        """``getC()`` (:949): prod u_i / prod h_i, after verify."""
        return self._get("C")

    def getD(self):
        """``getD()`` (:958): B_{N-1} / h_0^(prod e_i), after verify."""
        return self._get("D")

    def readCommitment(self, bt, n: int, width: int):
        """``setCommitment(ByteTreeReader)`` (:780-823): parse the published byte tree (framing, range and subgroup
        membership of every element, on the GPU); malformed input gives None -- the caller substitutes trivial values."""
        return Message.fromByteTree(self.G, bt, self._com_kinds, [n, 1, n, 1, 1, 2 * width])

    def readReply(self, bt, n: int, width: int):
        return Message.fromByteTree(self.G, bt, self._rep_kinds, [1, n, 1, 1, n, width])


class PoSCBasicTW(_NativeProof):
    """``vmn_posc_*`` — ref: hvzk/PoSCBasicTW.java."""
    _prefix = "posc"
    _com_keys = ("B", "Ap", "Bp", "Cp", "Dp")
    _com_scalar = ("Ap", "Cp", "Dp")
    _com_kinds = (GARRAY, ELEMENTS, GARRAY, ELEMENTS, ELEMENTS)
    _rep_keys = ("k_A", "k_B", "k_C", "k_D", "k_E")
    _rep_scalar = ("k_A", "k_C", "k_D")
    _rep_kinds = (RING, RARRAY, RING, RING, RARRAY)

    def setInstance(self, g, h, u, r=None, pi=None):
        self._keep = [h, u, r]
        ptr, keep = _u32_array(pi) if pi is not None else (None, None)
        self._call("set_instance", self.G.enc_el(g), h._h, u._h, r._h if r is not None else None, ptr)

    def commit(self):
        return self._commit()

    def setCommitment(self, msg):
        self._com = msg if isinstance(msg, Message) else self._as_msg(msg, self._com_keys, self._com_scalar, self._com_kinds)
        self._call("set_commitment", self._com._h)

    def verifyPrepare(self, reply) -> None:
        """``vmn_posc_verify_prepare``: the part of verify() that needs the reply but not the challenge."""
        self._call("verify_prepare", reply._h)

    def verify(self, reply) -> bool:
        m = reply if isinstance(reply, Message) else self._as_msg(reply, self._rep_keys, self._rep_scalar, self._rep_kinds)
        verdict = C.c_int(0)
        self._call("verify", m._h, C.byref(verdict))
        return bool(verdict.value)

    def _get(self, name: str):
        out = C.create_string_buffer(self.G.elem_bytes)
        self._call(f"get_{name}", out)
        return self.G.dec_el(out.raw)

    def getA(self):
        """A = u.expProd(e) (PoSCBasicTW.java:676), after verify."""
        return self._get("A")

    def getC(self):
        return self._get("C")

    def getD(self):
        return self._get("D")


class CCPoSBasicW(_NativeProof):
    """``vmn_ccpos_*`` — ref: hvzk/CCPoSBasicW.java."""
    _prefix = "ccpos"
    _com_keys = ("Ap", "Bp")
    _com_scalar = ("Ap",)
    _com_kinds = (ELEMENTS, ELEMENTS)
    _rep_keys = ("k_A", "k_B", "k_E")
    _rep_scalar = ("k_A",)
    _rep_kinds = (RING, RING, RARRAY)

    def setInstance(self, g, h, u, pkey, w, wp, r=None, pi=None, s=None):
        width = len(pkey) // 2
        self._keep = [h, u, w, wp, r, s]
        ptr, keep = _u32_array(pi) if pi is not None else (None, None)
        self._call("set_instance", self.G.enc_el(g), h._h, u._h, self.G.enc_els(pkey), C.c_size_t(width), _ptr_array(w),
                   _ptr_array(wp), r._h if r is not None else None, ptr, _ptr_array(s) if s is not None else None)

    def commit(self):
        return self._commit()

    def setCommitment(self, msg):
        self._com = msg if isinstance(msg, Message) else self._as_msg(msg, self._com_keys, self._com_scalar, self._com_kinds)
        self._call("set_commitment", self._com._h)

    def computeAB(self, raisedu=None):
        self._call("compute_ab", raisedu._h if raisedu is not None else None)

    def getAB(self):
        """The values of computeAB (CCPoSBasicW.java:493-506): plain form (A, [B components]), raised form (None, [AB components])."""
        width = int(plib().vmn_ccpos_width(self._h))
        out = C.create_string_buffer((1 + 2 * width) * self.G.elem_bytes)
        cnt = C.c_size_t(0)
        self._call("get_AB", out, C.byref(cnt))
        els = self.G.dec_els(out.raw[: cnt.value * self.G.elem_bytes])
        return (els[0], els[1:]) if cnt.value == 1 + 2 * width else (None, els)

    def verifyPrepare(self, reply, raisedh=None, raisedExponent: Optional[int] = None) -> None:
        """``vmn_ccpos_verify_prepare``: the reply side of verify() -- all its array work; same arguments as the verify() that follows."""
        if raisedExponent is None:
            self._call("verify_prepare", reply._h, None, None, C.c_size_t(0))
        else:
            rho = _be(raisedExponent)
            self._call("verify_prepare", reply._h, raisedh._h, rho, C.c_size_t(len(rho)))

    def verify(self, reply, raisedh=None, raisedExponent: Optional[int] = None) -> bool:
        m = reply if isinstance(reply, Message) else self._as_msg(reply, self._rep_keys, self._rep_scalar, self._rep_kinds)
        verdict = C.c_int(0)
        if raisedExponent is None:
            self._call("verify", m._h, None, None, C.c_size_t(0), C.byref(verdict))
        else:
            rho = _be(raisedExponent)
            self._call("verify", m._h, raisedh._h, rho, C.c_size_t(len(rho)), C.byref(verdict))
        return bool(verdict.value)


# ---- shuffler lines (the signatures of mixnet.py) -----------------------------------------------------------------
def reencrypt_native(group, pkey, w, s, pi):
    """``vmn_shuffle_reencrypt``: w' = permute(w * pk^s, pi^-1) in one call (factors are freed inside)."""
    width = len(pkey) // 2
    out = (C.c_void_p * (2 * width))()
    ptr, keep = _u32_array(pi)
    _check(plib().vmn_shuffle_reencrypt(group._h, group.enc_els(pkey), C.c_size_t(width), _ptr_array(w), _ptr_array(s), ptr, out))
    return [PGroupElementArray(group, C.c_void_p(h)) for h in out]


def reencryption_factors_native(group, pkey, s):
    """``vmn_shuffle_reencryption_factors``: pk^s, the offline half of a precomputed shuffle (ShufflerElGamalSession.java:645-661)."""
    width = len(pkey) // 2
    out = (C.c_void_p * (2 * width))()
    _check(plib().vmn_shuffle_reencryption_factors(group._h, group.enc_els(pkey), C.c_size_t(width), _ptr_array(s), out))
    return [PGroupElementArray(group, C.c_void_p(h)) for h in out]


def apply_factors_native(group, w, factors, pi):
    """``vmn_shuffle_apply_factors``: w' = permute(w * factors, pi^-1), the online half (ShufflerElGamalSession.java:789-792)."""
    width = len(w) // 2
    out = (C.c_void_p * (2 * width))()
    ptr, keep = _u32_array(pi)
    _check(plib().vmn_shuffle_apply_factors(group._h, C.c_size_t(width), _ptr_array(w), _ptr_array(factors), ptr, out))
    return [PGroupElementArray(group, C.c_void_p(h)) for h in out]


def reencrypt_shard_native(group, pkey, w_full, s_full, pi, lo: int, hi: int):
    """``vmn_shuffle_reencrypt_shard``: positions [lo, hi) of w' = permute(w * pk^s, pi^-1), out of the whole w and s."""
    width = len(pkey) // 2
    out = (C.c_void_p * (2 * width))()
    ptr, keep = _u32_array(pi)
    _check(plib().vmn_shuffle_reencrypt_shard(group._h, group.enc_els(pkey), C.c_size_t(width), _ptr_array(w_full), _ptr_array(s_full),
                                              ptr, C.c_size_t(lo), C.c_size_t(hi), out))
    return [PGroupElementArray(group, C.c_void_p(h)) for h in out]


def permutation_commitment_shard_native(group, g, h_full, r_full, pi, lo: int, hi: int):
    """``vmn_permutation_commitment_shard``: positions [lo, hi) of u = permute(h * g^r, pi)."""
    out = C.c_void_p()
    ptr, keep = _u32_array(pi)
    _check(plib().vmn_permutation_commitment_shard(group._h, group.enc_el(g), h_full._h, r_full._h, ptr, C.c_size_t(lo), C.c_size_t(hi),
                                                   C.byref(out)))
    return PGroupElementArray(group, out)


def reencrypt_shard_seeded_native(group, pkey, w_full, tape, rbitlen: int, pi, lo: int, hi: int):
    """``vmn_shuffle_reencrypt_shard_seeded``: positions [lo, hi) of w' and of the re-encryption exponents s, the exponents
    being PRG draws of the tape's 32-byte seeds (one per column) of which only the rows this rank reads are generated.
    Returns (w' shard: 2 width arrays, s shard: width arrays)."""
    width = len(pkey) // 2
    rs = RandomSource(group, tape)
    out = (C.c_void_p * (2 * width))()
    s_out = (C.c_void_p * width)()
    ptr, keep = _u32_array(pi)
    rc = plib().vmn_shuffle_reencrypt_shard_seeded(group._h, group.enc_els(pkey), C.c_size_t(width), _ptr_array(w_full), C.byref(rs.struct),
                                                   C.c_int(rbitlen), ptr, C.c_size_t(lo), C.c_size_t(hi), out, s_out)
    if rc != 0 and rs.error is not None:
        raise rs.error
    _check(rc)
    return ([PGroupElementArray(group, C.c_void_p(h)) for h in out], [PRingElementArray(group, C.c_void_p(h)) for h in s_out])


def permutation_commitment_shard_seeded_native(group, g, h_full, tape, rbitlen: int, pi, lo: int, hi: int):
    """``vmn_permutation_commitment_shard_seeded``: (u[lo, hi), r[lo, hi)) with r a PRG draw of the tape's next seed."""
    rs = RandomSource(group, tape)
    u, r = C.c_void_p(), C.c_void_p()
    ptr, keep = _u32_array(pi)
    rc = plib().vmn_permutation_commitment_shard_seeded(group._h, group.enc_el(g), h_full._h, C.byref(rs.struct), C.c_int(rbitlen), ptr,
                                                        C.c_size_t(lo), C.c_size_t(hi), C.byref(u), C.byref(r))
    if rc != 0 and rs.error is not None:
        raise rs.error
    _check(rc)
    return PGroupElementArray(group, u), PRingElementArray(group, r)


def permutation_commitment_native(group, g, h, r, pi):
    """``vmn_permutation_commitment``: u = permute(h * g^r, pi)."""
    out = C.c_void_p()
    ptr, keep = _u32_array(pi)
    _check(plib().vmn_permutation_commitment(group._h, group.enc_el(g), h._h, r._h, ptr, C.byref(out)))
    return PGroupElementArray(group, out)


def permutation_shrink_native(pi, n: int):
    """``vmn_permutation_shrink``: (keep list, compressed permutation) of PermutationCommitment.shrink."""
    n_max = len(pi)
    ptr, keepalive = _u32_array(pi)
    keep = C.create_string_buffer(max(1, n_max))
    out = (C.c_uint32 * max(1, n))()
    _check(plib().vmn_permutation_shrink(ptr, C.c_size_t(n_max), C.c_size_t(n), keep, out))
    return [b != 0 for b in keep.raw[:n_max]], list(out[:n])


def keep_list_sanitize_native(keep, n_max: int, n: int):
    """``vmn_keep_list_sanitize``: the keep list a verifier uses (the trivial one unless exactly n of n_max are set)."""
    buf = C.create_string_buffer(bytes(1 if k else 0 for k in keep) + b"\0" * max(0, n_max - len(keep)), max(n_max, len(keep), 1))
    replaced = C.c_int()
    _check(plib().vmn_keep_list_sanitize(buf, C.c_size_t(len(keep)), C.c_size_t(n_max), C.c_size_t(n), C.byref(replaced)))
    return [b != 0 for b in buf.raw[:n_max]], bool(replaced.value)


def random_ring_array_native(group, tape, n: int, rbitlen: int):
    """``vmn_rarray_random``: pRing.randomElementArray(n, randomSource, rbitlen) -- expanded on the device from 32 bytes of
    the tape when it offers ``array_seed()`` (the re-encryption exponents, ShufflerElGamalSession.java:408-409)."""
    rs = RandomSource(group, tape)
    out = C.c_void_p()
    rc = plib().vmn_rarray_random(group._h, C.byref(rs.struct), C.c_size_t(n), C.c_int(rbitlen), C.byref(out))
    if rc != 0 and rs.error is not None:
        raise rs.error
    _check(rc)
    return PRingElementArray(group, out)


# ---- verifiable threshold decryption (the interface of elgamal.py over the C++ drivers) -------------------------------
def _flags(correct):
    return bytes(1 if c else 0 for c in correct)


def prodFactor(q: int, k: int, group=None) -> int:
    """``vmn_prod_factor`` (needs the group whose order is q)."""
    out = C.create_string_buffer(group.exp_bytes)
    _check(plib().vmn_prod_factor(group._h, C.c_int(k), out))
    return int.from_bytes(out.raw, "big")


def modifiedLagrangeCoefficients(q: int, correct, k: int, threshold: int, group=None):
    """``vmn_lagrange_coefficients``: signed integers of smallest absolute value."""
    absb = C.create_string_buffer(threshold * group.exp_bytes)
    neg = (C.c_int * threshold)()
    _check(plib().vmn_lagrange_coefficients(group._h, _flags(correct), C.c_int(k), C.c_int(threshold), absb, neg))
    nb = group.exp_bytes
    return [(-1 if neg[t] else 1) * int.from_bytes(absb.raw[t * nb:(t + 1) * nb], "big") for t in range(threshold)]


def decryptionFactors(u, secretKey: int, q: int, k: int):
    out = C.c_void_p()
    _check(plib().vmn_decryption_factors(u.group._h, u._h, int_to_be(secretKey % q, u.group.exp_bytes), C.c_int(k), C.byref(out)))
    return PGroupElementArray(u.group, out)


def _opt_ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[(a._h.value if a is not None else None) for a in arrs])


def combineDecryptionFactors(decryptionFactors_, correct, k: int, threshold: int, q: int):
    group = next(a for a in decryptionFactors_ if a is not None).group
    out = C.c_void_p()
    _check(plib().vmn_combine_decryption_factors(group._h, _opt_ptr_array(decryptionFactors_), _flags(correct), C.c_int(k),
                                                 C.c_int(threshold), C.byref(out)))
    return PGroupElementArray(group, out)


def plaintexts(v, combinedFactors):
    return v.mul(combinedFactors)


class DistrElGamalSessionBasic:
    """``vmn_decproof_*`` — ref: elgamal/DistrElGamalSessionBasic.java (one instance per party j)."""

    def __init__(self, group, j: int, k: int, threshold: int, ebitlen: int, rand=None):
        self.G, self.j, self.k, self.q = group, j, k, group.q
        self._rs = RandomSource(group, rand) if rand is not None else None
        self._h = C.c_void_p()
        _check(plib().vmn_decproof_create(group._h, C.c_int(j), C.c_int(k), C.c_int(threshold), C.c_int(ebitlen),
                                          C.byref(self._rs.struct) if self._rs else None, C.byref(self._h)))
        self.k_x = {}
        self._keep = []

    def free(self):
        if self._h and self.G.alive:
            plib().vmn_decproof_free(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def _call(self, name, *args):
        _check(getattr(plib(), "vmn_decproof_" + name)(self._h, *args))

    def setInstance(self, u, y, f):
        G = self.G
        ybuf = b"".join(G.enc_el(el) if el is not None else bytes(G.elem_bytes) for el in y)
        self._keep = [u, f]
        self._call("set_instance", u._h, ybuf, _opt_ptr_array(f))

    def setBatchVector(self, e_ints):
        blk = host_block(e_ints) or host_block(b"".join(int_to_be(x, self.G.exp_bytes) for x in e_ints))
        self._call("set_batch_vector", blk[0])

    def setBatchVectorSeed(self, seed: bytes):
        self._call("set_batch_vector_seed", bytes(seed), C.c_size_t(len(seed)))

    def batchInput(self):
        self._call("batch_input")

    def commit(self, x: int):
        G = self.G
        yp, Bp = C.create_string_buffer(G.elem_bytes), C.create_string_buffer(G.elem_bytes)
        self._call("commit", int_to_be(x % self.q, G.exp_bytes), yp, Bp)
        return G.dec_el(yp.raw), G.dec_el(Bp.raw)

    def reply(self, v: int) -> int:
        b = _be(v)
        out = C.create_string_buffer(self.G.exp_bytes)
        self._call("reply", b, C.c_size_t(len(b)), out)
        self.k_x[self.j] = int.from_bytes(out.raw, "big")
        return self.k_x[self.j]

    def setCommitment(self, l: int, yp, Bp):
        self._call("set_commitment", C.c_int(l), self.G.enc_el(yp), self.G.enc_el(Bp))

    def setReply(self, l: int, k_x: int):
        # a value that fits the wire width is handed over as it is: one >= q is not a field element and costs the party
        # its verdict (DistrElGamalSessionBasic.java:606-613), it is not reduced
        raw = k_x if 0 <= k_x < 1 << (8 * self.G.exp_bytes) else k_x % self.q
        self.k_x[l] = k_x % self.q
        self._call("set_reply", C.c_int(l), int_to_be(raw, self.G.exp_bytes))

    def batch(self, l: int):
        self._call("batch", C.c_int(l))

    def verify(self, l: int, v: int) -> bool:
        b = _be(v)
        verdict = C.c_int(0)
        self._call("verify", C.c_int(l), b, C.c_size_t(len(b)), C.byref(verdict))
        return bool(verdict.value)

    def combine(self, correct, combinedy, combinedf):
        self._keep.append(combinedf)
        self._call("combine", _flags(correct), self.G.enc_el(combinedy), combinedf._h)

    def batchCombined(self):
        self._call("batch_combined")

    def verifyCombined(self, v: int) -> bool:
        b = _be(v)
        verdict = C.c_int(0)
        self._call("verify_combined", b, C.c_size_t(len(b)), C.byref(verdict))
        return bool(verdict.value)


class IndependentGeneratorsBasicI:
    """``vmn_igen_*`` — ref: distr/IndependentGeneratorsBasicI.java (the interface of hvzk.IndependentGeneratorsBasicI)."""

    def __init__(self, group, j: int, threshold: int, ebitlen: int, rand=None):
        self.G, self.j, self.threshold, self.q = group, j, threshold, group.q
        self._rs = RandomSource(group, rand) if rand is not None else None
        self._h = C.c_void_p()
        _check(plib().vmn_igen_create(group._h, C.c_int(j), C.c_int(threshold), C.c_int(ebitlen),
                                      C.byref(self._rs.struct) if self._rs else None, C.byref(self._h)))
        self._keep = []

    def free(self):
        if self._h and self.G.alive:
            plib().vmn_igen_free(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def _call(self, name, *args):
        _check(getattr(plib(), "vmn_igen_" + name)(self._h, *args))

    def setInstance(self, g, h, s, combinedh):
        self._keep = [h, s, combinedh]
        self._call("set_instance", self.G.enc_el(g), _opt_ptr_array(h), s._h if s is not None else None, combinedh._h)

    def setBatchVector(self, e_ints):
        blk = host_block(e_ints) or host_block(b"".join(int_to_be(x, self.G.exp_bytes) for x in e_ints))
        self._call("set_batch_vector", blk[0])

    def setBatchVectorSeed(self, seed: bytes):
        self._call("set_batch_vector_seed", bytes(seed), C.c_size_t(len(seed)))

    def commit(self):
        out = C.create_string_buffer(self.G.elem_bytes)
        self._call("commit", out)
        return self.G.dec_el(out.raw)

    def setCommitment(self, l: int, Ap):
        self._call("set_commitment", C.c_int(l), self.G.enc_el(Ap))

    def setChallenge(self, v: int):
        b = _be(v)
        self._call("set_challenge", b, C.c_size_t(len(b)))

    def reply(self) -> int:
        out = C.create_string_buffer(self.G.exp_bytes)
        self._call("reply", out)
        return int.from_bytes(out.raw, "big")

    def setReply(self, l: int, k_a: int):
        self._call("set_reply", C.c_int(l), int_to_be(k_a % (1 << (8 * self.G.exp_bytes)), self.G.exp_bytes))

    def verify(self, l: Optional[int] = None) -> bool:
        verdict = C.c_int(0)
        if l is None:
            self._call("verify_combined", C.byref(verdict))
        else:
            self._call("verify", C.c_int(l), C.byref(verdict))
        return bool(verdict.value)
