"""verificatum-vmn_amd — host-side mirror of the array interface the Verificatum Mix-Net calls.

The reference (verificatum-vmn, Java) performs all arithmetic through VCR's array classes
(``PGroupElementArray``, ``PRingElementArray`` …; SURVEY.md §2.3 / App. B list the call sites in
``/root/reference``).  This package binds the C ABI of ``include/vmnhip.h`` (``libvmnhip.so``,
hand-written HIP kernels for gfx950) with ``ctypes`` and exposes it under the reference's method
names (``exp``, ``expProd``, ``mul``, ``prod``, ``permute``, ``shiftPush``, ``recLin``, ``prods``,
``mulAdd``, ``innerProduct`` …) so that the proof drivers and the parity tests read like the
reference's own code.

There is no CPU implementation behind these classes: without the HIP library and a gfx950 GPU
every constructor raises ``VmnError``.

The directory name contains a hyphen (it is fixed by the build contract); import it with
``load_package()`` from ``__graft_entry__`` or ``importlib`` under the name
``verificatum_vmn_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

try:                                    # optional: large index arrays are passed without a Python-level copy
    import numpy as _np
except Exception:                       # pragma: no cover
    _np = None


def _u32_array(seq):
    """ctypes uint32 array (or a numpy-backed pointer) for an index list."""
    if _np is not None and isinstance(seq, _np.ndarray):
        arr = _np.ascontiguousarray(seq, dtype=_np.uint32)
        return arr.ctypes.data_as(C.POINTER(C.c_uint32)), arr
    if _np is not None and len(seq) > 4096:
        arr = _np.asarray(seq, dtype=_np.uint32)
        return arr.ctypes.data_as(C.POINTER(C.c_uint32)), arr
    arr = (C.c_uint32 * len(seq))(*seq)
    return arr, arr

def host_block(values):
    """(pointer argument, byte length, keep-alive) of a block of big-endian rows handed in as ``bytes`` or as a
    host buffer object (a pinned ``torch`` tensor or a numpy array of uint8): the latter cross without a copy and,
    when page-locked, upload asynchronously at PCIe speed."""
    if isinstance(values, (bytes, bytearray)):
        b = bytes(values)
        return b, len(b), b
    if hasattr(values, "data_ptr"):                          # torch tensor (host)
        return C.c_void_p(values.data_ptr()), values.numel() * values.element_size(), values
    if _np is not None and isinstance(values, _np.ndarray) and values.dtype == _np.uint8:
        arr = _np.ascontiguousarray(values)
        return C.c_void_p(arr.ctypes.data), arr.size, arr
    return None


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvmnhip.so")


class VmnError(RuntimeError):
    """Raised for every non-zero status of the C ABI (misuse, device errors)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"vmnhip status {status}: {message}")
        self.status = status


_lib: Optional[C.CDLL] = None
_plib_scalar = None        # libvmnproofs.so for the host arithmetic on single elements (False: not built)


def lib() -> C.CDLL:
    """Load ``libvmnhip.so`` (built in-tree by ``__graft_entry__.build()``).  Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VmnError(-2, f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.vmn_last_error.restype = C.c_char_p
        _lib.vmn_version.restype = C.c_char_p
        _lib.vmn_ctx_get_stream.restype = C.c_void_p
        _lib.vmn_pending_free.restype = None
        for name in ("vmn_group_elem_bytes", "vmn_group_exp_bytes", "vmn_garray_size", "vmn_rarray_size", "vmn_group_table_bytes",
                     "vmn_garray_bytetree_size", "vmn_rarray_bytetree_size", "vmn_pending_bytes"):
            getattr(_lib, name).restype = C.c_size_t
    return _lib


def _check(status: int) -> None:
    if status != 0:
        raise VmnError(status, lib().vmn_last_error().decode("utf-8", "replace"))


def int_to_be(x: int, nbytes: int) -> bytes:
    return int(x).to_bytes(nbytes, "big")


def ints_to_be(xs: Sequence[int], nbytes: int) -> bytes:
    return b"".join(int(x).to_bytes(nbytes, "big") for x in xs)


def be_to_ints(buf: bytes, nbytes: int) -> list:
    return [int.from_bytes(buf[i:i + nbytes], "big") for i in range(0, len(buf), nbytes)]


class Context:
    """One GPU, one HIP stream, scratch workspace (``vmn_ctx``)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().vmn_ctx_create(C.c_int(device), C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            lib().vmn_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream: Optional[int]) -> None:
        """Run on an externally owned ``hipStream_t`` (e.g. ``torch.cuda.current_stream().cuda_stream``)."""
        _check(lib().vmn_ctx_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    @property
    def stream(self) -> int:
        return lib().vmn_ctx_get_stream(self._h) or 0

    def synchronize(self) -> None:
        _check(lib().vmn_ctx_synchronize(self._h))

    @property
    def num_cus(self) -> int:
        return lib().vmn_ctx_num_cus(self._h)

    def set_small_array_threshold(self, items: int) -> None:
        """Launches over at most ``items`` elements use the wide (four lanes per element) geometry for 2048-bit moduli
        (``vmn_ctx_set_small_array_threshold``); 0 = never.  A tuning knob: results never depend on it."""
        _check(lib().vmn_ctx_set_small_array_threshold(self._h, C.c_size_t(min(int(items), 2 ** 64 - 1))))

    def set_tiny_array_threshold(self, items: int) -> None:
        """Launches over at most ``items`` elements of a 2048-bit modulus use eight lanes per element
        (``vmn_ctx_set_tiny_array_threshold``); 0 = never.  Takes precedence over ``set_small_array_threshold``."""
        _check(lib().vmn_ctx_set_tiny_array_threshold(self._h, C.c_size_t(min(int(items), 2 ** 64 - 1))))

    def helper_mark(self) -> None:
        """Protocol thread: what is queued up to here is what the helper may rely on (``vmn_ctx_helper_mark``)."""
        _check(lib().vmn_ctx_helper_mark(self._h))

    def helper(self):
        """``with ctx.helper(): ...`` in the ONE helper thread of a party (ShufflerElGamalSession.java:839-859): calls made
        inside run on the context's helper lane -- a second, high-priority stream with its own pool and lock
        (``vmn_ctx_helper_begin`` / ``vmn_ctx_helper_end``)."""
        ctx = self

        class _Helper:
            def __enter__(self_inner):
                _check(lib().vmn_ctx_helper_begin(ctx._h))
                return self_inner

            def sync(self_inner):
                """Order the helper's stream behind the protocol thread's latest ``helper_mark()``."""
                _check(lib().vmn_ctx_helper_sync(ctx._h))

            def __exit__(self_inner, *exc):
                _check(lib().vmn_ctx_helper_end(ctx._h))
                return False
        return _Helper()

    def memory_stats(self) -> dict:
        """Bytes / blocks of freed arrays cached for reuse and bytes of live allocations (arrays + temporaries)."""
        pb, nb, lb = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _check(lib().vmn_ctx_memory_stats(self._h, C.byref(pb), C.byref(nb), C.byref(lb)))
        return {"pool_bytes": pb.value, "pool_blocks": nb.value, "live_bytes": lb.value}

    def timing_enable(self, on: bool = True) -> None:
        _check(lib().vmn_ctx_timing_enable(self._h, C.c_int(1 if on else 0)))

    def timing_reset(self) -> None:
        _check(lib().vmn_ctx_timing_reset(self._h))

    def timing_report(self) -> dict:
        """{family: (launches, total_ms, executed multiply-adds, the same products in canonical 32-bit multiply-accumulates)}
        of every kernel family since the last reset."""
        buf = C.create_string_buffer(16384)
        _check(lib().vmn_ctx_timing_report(self._h, buf, C.c_size_t(len(buf))))
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms, mads, canon = line.split()
            # mads: v_mad_u64_u32 executed (28-bit limbs); canon: SURVEY.md §8d's M(s) = 2 s^2 + s per product, s = bits / 32
            out[name] = (int(cnt), float(ms), float(mads), float(canon))
        return out

    def timing_get(self, family: str):
        n = C.c_long()
        ms = C.c_double()
        _check(lib().vmn_ctx_timing_get(self._h, family.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value


class ModPGroup:
    """``com.verificatum.arithm.ModPGroup``: the order-q subgroup of Z_p^* with generator g."""

    def __init__(self, ctx: Context, p: int, q: int, g: int, nbytes: Optional[int] = None):
        """``nbytes`` = one explicit width for elements and exponents on the host side; ``None`` = the reference's own
        widths (Java ``BigInteger.toByteArray().length`` of p resp. q: 257 / 256 bytes for an RFC 3526 2048-bit
        group), which is what VCR writes into byte-tree leaves."""
        self.ctx = ctx
        self.p, self.q, self.g = int(p), int(q), int(g)
        nb = nbytes or (self.p.bit_length() + 7) // 8
        self._h = C.c_void_p()
        _check(lib().vmn_modp_group_create(ctx._h, int_to_be(p, nb), int_to_be(q, nb),
                                           int_to_be(g, nb), C.c_size_t(nb), C.byref(self._h)))
        if nbytes is None:
            _check(lib().vmn_group_set_wire_bytes(self._h, C.c_size_t(0), C.c_size_t(0)))
        self._read_widths()

    def _read_widths(self) -> None:
        self.nbytes = lib().vmn_group_elem_bytes(self._h)        # bytes of a group element
        self.exp_bytes = lib().vmn_group_exp_bytes(self._h)      # bytes of a ring element (exponent)

    def close(self) -> None:
        # the garbage collector may finalise a context before the groups / arrays that were created on it
        # (cycles, interpreter shutdown): native objects of a closed context are gone with it, never touch them
        if self._h and self.ctx._h:
            lib().vmn_group_destroy(self._h)
        self._h = C.c_void_p()

    @property
    def alive(self) -> bool:
        return bool(self._h) and bool(self.ctx._h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- element codec and single-element ("scalar") group operations ---------------------------
    @property
    def elem_bytes(self) -> int:
        return self.nbytes

    @property
    def ONE(self):
        return 1

    def enc_el(self, el) -> bytes:
        return int_to_be(el, self.nbytes)

    def dec_el(self, buf: bytes):
        return int.from_bytes(buf, "big")

    def enc_els(self, els) -> bytes:
        return ints_to_be(els, self.nbytes)

    def dec_els(self, buf: bytes) -> list:
        return be_to_ints(buf, self.nbytes)

    # Single elements: through the host arithmetic of libvmnproofs.so (vmn_element_*: 64-bit Montgomery / Jacobian
    # curve code, several times faster than Python integers / the affine Python curve code for the ~20
    # exponentiations of a proof) when that library is there, else plain Python.
    def _scalar_lib(self):
        global _plib_scalar
        if _plib_scalar is None:
            path = os.path.join(_HERE, "libvmnproofs.so")
            _plib_scalar = C.CDLL(path) if os.path.exists(path) else False
        return _plib_scalar

    def _k_native(self, fn: str, *args):
        """(True, element) through libvmnproofs.so, or (False, None) when that library is not built."""
        pl = self._scalar_lib()
        if not pl or not self.alive:
            return False, None
        out = C.create_string_buffer(self.elem_bytes)
        _check(getattr(pl, fn)(self._h, *args, out))
        return True, self.dec_el(out.raw)

    def k_mul(self, a, b):
        ok, r = self._k_native("vmn_element_mul", self.enc_el(a), self.enc_el(b))
        return r if ok else self._py_mul(a, b)

    def k_exp(self, a, e: int):
        eb = int(e % self.q).to_bytes(self.exp_bytes, "big")
        ok, r = self._k_native("vmn_element_exp", self.enc_el(a), eb, C.c_size_t(len(eb)))
        return r if ok else self._py_exp(a, e)

    def k_inv(self, a):
        ok, r = self._k_native("vmn_element_inv", self.enc_el(a))
        return r if ok else self._py_inv(a)

    def _py_mul(self, a, b):
        return a * b % self.p

    def _py_exp(self, a, e: int):
        return pow(a, e % self.q, self.p)

    def _py_inv(self, a):
        return pow(a, -1, self.p)

    # -- constructors mirroring pGroup.toElementArray / pRing.toElementArray --------------------
    def toElementArray(self, values, checked: bool = True) -> "PGroupElementArray":
        """values: sequence of ints, or a block of n*nbytes big-endian bytes (bytes / pinned tensor / uint8 array)."""
        blk = host_block(values) or host_block(ints_to_be(values, self.nbytes))
        n = blk[1] // self.nbytes
        h = C.c_void_p()
        ok = C.c_int(1)
        _check(lib().vmn_garray_from_be(self._h, blk[0], C.c_size_t(n), C.byref(h), C.byref(ok)))
        arr = PGroupElementArray(self, h)
        arr.all_in_range = bool(ok.value)
        if checked and not ok.value:
            arr.free()
            raise ValueError("ArithmFormatException: group element out of range")
        return arr

    def ringArray(self, values, checked: bool = True) -> "PRingElementArray":
        blk = host_block(values) or host_block(ints_to_be(values, self.exp_bytes))
        n = blk[1] // self.exp_bytes
        h = C.c_void_p()
        ok = C.c_int(1)
        _check(lib().vmn_rarray_from_be(self._h, blk[0], C.c_size_t(n), C.byref(h), C.byref(ok)))
        arr = PRingElementArray(self, h)
        if checked and not ok.value:
            arr.free()
            raise ValueError("ArithmFormatException: ring element out of range")
        return arr

    def _from_bytetree(self, fn: str, cls, bt: bytes, expected_n: int):
        h = C.c_void_p()
        fmt, rng = C.c_int(0), C.c_int(1)
        blk = host_block(bt) or host_block(bytes(bt))      # bytes, or a host buffer object (pinned tensor): no copy
        _check(getattr(lib(), fn)(self._h, blk[0], C.c_size_t(blk[1]), C.c_size_t(expected_n), C.byref(h),
                                   C.byref(fmt), C.byref(rng)))
        if not fmt.value:
            raise ValueError("EIOException: not a byte tree of %d-byte leaves" % self.nbytes)
        arr = cls(self, h)
        if not rng.value:
            arr.free()
            raise ValueError("ArithmFormatException: element out of range")
        return arr

    def toElementArrayFromByteTree(self, bt: bytes, size: int = 0) -> "PGroupElementArray":
        """``pGroup.toElementArray(size, byteTreeReader)`` (size 0 = any), the reference's wire format."""
        return self._from_bytetree("vmn_garray_from_bytetree", PGroupElementArray, bt, size)

    def ringArrayFromByteTree(self, bt: bytes, size: int = 0) -> "PRingElementArray":
        return self._from_bytetree("vmn_rarray_from_bytetree", PRingElementArray, bt, size)

    def ringArrayFromPRG(self, seed: bytes, n: int, bits: int) -> "PRingElementArray":
        """``prg.setSeed(seed); LargeIntegerArray.random(n, bits, prg)`` as field elements, generated on the GPU
        (the random vector of a proof, PoSBasicTW.java:533-538)."""
        h = C.c_void_p()
        _check(lib().vmn_rarray_from_prg(self._h, bytes(seed), C.c_size_t(len(seed)), C.c_size_t(n), C.c_int(bits), C.byref(h)))
        return PRingElementArray(self, h)

    def ringArrayFromPRGRange(self, seed: bytes, first: int, n: int, bits: int) -> "PRingElementArray":
        """Values [first, first + n) of ``ringArrayFromPRG`` without the rest (``vmn_rarray_from_prg_range``)."""
        h = C.c_void_p()
        _check(lib().vmn_rarray_from_prg_range(self._h, bytes(seed), C.c_size_t(len(seed)), C.c_size_t(first), C.c_size_t(n),
                                               C.c_int(bits), C.byref(h)))
        return PRingElementArray(self, h)

    def ringArrayFromPRGGather(self, seed: bytes, idx, bits: int) -> "PRingElementArray":
        """The values ``idx`` of ``ringArrayFromPRG`` (``vmn_rarray_from_prg_gather``): a permuted read of a PRG array."""
        arr, _keep = _u32_array(idx)
        h = C.c_void_p()
        _check(lib().vmn_rarray_from_prg_gather(self._h, bytes(seed), C.c_size_t(len(seed)), arr, C.c_size_t(len(idx)), C.c_int(bits),
                                                C.byref(h)))
        return PRingElementArray(self, h)

    def elementArrayFromPRG(self, seed: bytes, n: int, rbitlen: int) -> "PGroupElementArray":
        """``pGroup.randomElementArray(n, prg, rbitlen)``: independent generators derived on the GPU
        (IndependentGeneratorsRO.java:117-130; safe-prime ModPGroup)."""
        h = C.c_void_p()
        _check(lib().vmn_garray_from_prg(self._h, bytes(seed), C.c_size_t(len(seed)), C.c_size_t(n), C.c_int(rbitlen), C.byref(h)))
        return PGroupElementArray(self, h)

    def exp(self, base, exponents: "PRingElementArray") -> "PGroupElementArray":
        """``g.exp(PRingElementArray)``: fixed base, one exponent per element (K2)."""
        h = C.c_void_p()
        _check(lib().vmn_group_exp_fixed(self._h, self.enc_el(base), exponents._h, C.byref(h)))
        return PGroupElementArray(self, h)

    def precomputeFixed(self, base, n_hint: int, uses_hint: int = 16) -> None:
        """Session setup: build the fixed-base table of a long-lived base (generator, public key) sized for about
        ``uses_hint`` calls on arrays of about ``n_hint`` exponents (``vmn_group_precompute_fixed``)."""
        _check(lib().vmn_group_precompute_fixed(self._h, self.enc_el(base), C.c_size_t(n_hint), C.c_int(uses_hint)))

    def releaseFixed(self, base) -> None:
        """The table of a base that will not be used again leaves the cache; its memory serves the next table of that size
        (``vmn_group_release_fixed``; the proof drivers do this for a prover's h_0 when the proof object is freed)."""
        _check(lib().vmn_group_release_fixed(self._h, self.enc_el(base)))

    def tableBytes(self) -> int:
        """Bytes of HBM the cached fixed-base tables of this group hold (``vmn_group_table_bytes``)."""
        return int(lib().vmn_group_table_bytes(self._h))

    def mulPartials(self, partials):
        out = C.create_string_buffer(self.elem_bytes)
        _check(lib().vmn_group_mul_partials(self._h, self.enc_els(partials), C.c_size_t(len(partials)), out))
        return self.dec_el(out.raw)


class ECqPGroup(ModPGroup):
    """``com.verificatum.arithm.ECqPGroup`` over a NIST curve (the reference's default group is P-256,
    demo/mixnet/.conf:153).  Elements are affine points ``(x, y)`` (``None`` = infinity); on the wire
    x || y fixed width; ``mul`` is point addition, ``exp`` scalar multiplication; exponents live in Z_n."""

    def __init__(self, ctx: Context, name: str = "P-256", java_widths: bool = False):
        from . import ecscalar
        self._ec = ecscalar
        c = ecscalar.CURVES[name]
        self.ctx, self.name = ctx, name
        self.p, self.q, self.b = c["p"], c["n"], c["b"]
        self.g = (c["gx"], c["gy"])
        self._h = C.c_void_p()
        _check(lib().vmn_ec_group_create(ctx._h, name.encode(), C.byref(self._h)))
        if java_widths:
            _check(lib().vmn_group_set_wire_bytes(self._h, C.c_size_t(0), C.c_size_t(0)))
        self._read_widths()

    def _read_widths(self) -> None:
        self.nbytes = lib().vmn_group_elem_bytes(self._h) // 2   # bytes of one coordinate
        self.exp_bytes = lib().vmn_group_exp_bytes(self._h)

    @property
    def elem_bytes(self) -> int:
        return 2 * self.nbytes

    @property
    def ONE(self):
        return None

    def enc_el(self, el) -> bytes:
        if el is None:
            return b"\xff" * (2 * self.nbytes)
        return int_to_be(el[0], self.nbytes) + int_to_be(el[1], self.nbytes)

    def dec_el(self, buf: bytes):
        nb = self.nbytes
        if buf == b"\xff" * (2 * nb):
            return None
        return int.from_bytes(buf[:nb], "big"), int.from_bytes(buf[nb:], "big")

    def enc_els(self, els) -> bytes:
        return b"".join(self.enc_el(e) for e in els)

    def dec_els(self, buf: bytes) -> list:
        w = 2 * self.nbytes
        return [self.dec_el(buf[i:i + w]) for i in range(0, len(buf), w)]

    def _py_mul(self, a, b):
        return self._ec.add(a, b, self.p)

    def _py_exp(self, a, e: int):
        return self._ec.mul(e, a, self.p, self.q)

    def _py_inv(self, a):
        return self._ec.neg(a, self.p)

    def toElementArray(self, values, checked: bool = True) -> "PGroupElementArray":
        blk = host_block(values) or host_block(self.enc_els(values))
        n = blk[1] // self.elem_bytes
        h = C.c_void_p()
        ok = C.c_int(1)
        _check(lib().vmn_garray_from_be(self._h, blk[0], C.c_size_t(n), C.byref(h), C.byref(ok)))
        arr = PGroupElementArray(self, h)
        arr.all_in_range = bool(ok.value)
        if checked and not ok.value:
            arr.free()
            raise ValueError("ArithmFormatException: point not on the curve")
        return arr


class _ArrayBase:
    _free_fn = ""

    def __init__(self, group: ModPGroup, handle: C.c_void_p):
        self.group = group
        self._h = handle

    _borrowed = False      # True: the handle belongs to a vmn_msg / proof object (native.py); free() is a no-op

    def free(self) -> None:
        if self._h and not self._borrowed and self.group.alive:
            getattr(lib(), self._free_fn)(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PGroupElementArray(_ArrayBase):
    """Device-resident ``PGroupElementArray`` over a ``ModPGroup``."""

    _free_fn = "vmn_garray_free"
    all_in_range = True

    def size(self) -> int:
        return lib().vmn_garray_size(self._h)

    def toBytes(self) -> bytes:
        out = C.create_string_buffer(max(1, self.size() * self.group.elem_bytes))
        _check(lib().vmn_garray_to_be(self._h, out))
        return out.raw[: self.size() * self.group.elem_bytes]

    def toInts(self) -> list:
        """The elements as host values (ints for ModPGroup, affine points for ECqPGroup)."""
        return self.group.dec_els(self.toBytes())

    def toByteTree(self) -> bytes:
        """``array.toByteTree()``: node of N fixed-width leaves, framed on the GPU."""
        size = lib().vmn_garray_bytetree_size(self._h)
        out = C.create_string_buffer(size)
        _check(lib().vmn_garray_to_bytetree(self._h, out))
        return out.raw[:size]

    def byteTreeSize(self) -> int:
        return lib().vmn_garray_bytetree_size(self._h)

    def toByteTreeInto(self, host_buffer) -> int:
        """The byte tree written into a caller-owned host buffer (a pinned ``torch`` uint8 tensor or a uint8 array of
        at least ``byteTreeSize()`` bytes): no intermediate copies, the download runs at PCIe speed."""
        blk = host_block(host_buffer)
        size = self.byteTreeSize()
        if blk is None or blk[1] < size:
            raise ValueError("host buffer too small for the byte tree")
        _check(lib().vmn_garray_to_bytetree(self._h, blk[0]))
        return size

    def _new(self, h) -> "PGroupElementArray":
        return PGroupElementArray(self.group, h)

    def exp2(self, e: int, y: "PGroupElementArray", f: "PRingElementArray", fbits: int = 0) -> "PGroupElementArray":
        """``vmn_garray_exp2``: self[i]^e * y[i]^f[i] as one simultaneous power (modular groups)."""
        h = C.c_void_p()
        e = int(e)
        nb = max(1, (e.bit_length() + 7) // 8)
        _check(lib().vmn_garray_exp2(self._h, int_to_be(e, nb), C.c_size_t(nb), y._h, f._h, C.c_int(fbits), C.byref(h)))
        return self._new(h)

    def expPair(self, e: int, y: "PGroupElementArray", f: "PRingElementArray", fbits: int = 0):
        """``vmn_garray_exp_pair``: (self[i]^e, y[i]^f[i]) -- two powers, one launch when the arrays are small."""
        hx, hy = C.c_void_p(), C.c_void_p()
        e = int(e)
        nb = max(1, (e.bit_length() + 7) // 8)
        _check(lib().vmn_garray_exp_pair(self._h, int_to_be(e, nb), C.c_size_t(nb), y._h, f._h, C.c_int(fbits), C.byref(hx), C.byref(hy)))
        return self._new(hx), self._new(hy)

    # K1a / K1b
    def exp(self, e, ebits: int = 0) -> "PGroupElementArray":
        """``X.exp(PRingElementArray)`` (per-element exponents) or ``X.exp(int)`` (shared exponent)."""
        h = C.c_void_p()
        if isinstance(e, PRingElementArray):
            _check(lib().vmn_garray_exp_array(self._h, e._h, C.c_int(ebits), C.byref(h)))
        else:
            e = int(e)
            nb = max(1, (e.bit_length() + 7) // 8)
            _check(lib().vmn_garray_exp_scalar(self._h, int_to_be(e, nb), C.c_size_t(nb), C.byref(h)))
        return self._new(h)

    def expInts(self, exps: Sequence[int], ebits: int) -> "PGroupElementArray":
        """Exponents that are plain integers of ``ebits`` bits (not reduced mod q)."""
        nb = (ebits + 7) // 8
        h = C.c_void_p()
        _check(lib().vmn_garray_exp_ints(self._h, ints_to_be(exps, nb), C.c_size_t(nb), C.c_int(ebits), C.byref(h)))
        return self._new(h)

    # K3
    def expProd(self, e, ebits: int = 0):
        out = C.create_string_buffer(self.group.elem_bytes)
        if isinstance(e, PRingElementArray):
            _check(lib().vmn_garray_expprod(self._h, e._h, C.c_int(ebits), out))
        else:
            nb = (ebits + 7) // 8
            _check(lib().vmn_garray_expprod_ints(self._h, ints_to_be(e, nb), C.c_size_t(nb), C.c_int(ebits), out))
        return self.group.dec_el(out.raw)

    # K4 / K5 / K6
    def mul(self, other: "PGroupElementArray") -> "PGroupElementArray":
        h = C.c_void_p()
        _check(lib().vmn_garray_mul(self._h, other._h, C.byref(h)))
        return self._new(h)

    def prod(self):
        out = C.create_string_buffer(self.group.elem_bytes)
        _check(lib().vmn_garray_prod(self._h, out))
        return self.group.dec_el(out.raw)

    def inv(self) -> "PGroupElementArray":
        """Element-wise inverse (batch inversion)."""
        h = C.c_void_p()
        _check(lib().vmn_garray_inv(self._h, C.byref(h)))
        return self._new(h)

    def equals(self, other: "PGroupElementArray") -> bool:
        eq = C.c_int()
        _check(lib().vmn_garray_equals(self._h, other._h, C.byref(eq)))
        return bool(eq.value)

    # K7
    def permute(self, perm: Sequence[int]) -> "PGroupElementArray":
        arr, _keep = _u32_array(perm)          # any length: a shard of a permuted array is a gather
        h = C.c_void_p()
        _check(lib().vmn_garray_gather(self._h, arr, C.c_size_t(len(perm)), C.byref(h)))
        return self._new(h)

    def shiftPush(self, el) -> "PGroupElementArray":
        h = C.c_void_p()
        _check(lib().vmn_garray_shift_push(self._h, self.group.enc_el(el), C.byref(h)))
        return self._new(h)

    def copyOfRange(self, start: int, end: int) -> "PGroupElementArray":
        h = C.c_void_p()
        _check(lib().vmn_garray_copy_range(self._h, C.c_size_t(start), C.c_size_t(end), C.byref(h)))
        return self._new(h)

    def extract(self, keep: Sequence[bool]) -> "PGroupElementArray":
        buf = bytes(1 if k else 0 for k in keep)
        h = C.c_void_p()
        _check(lib().vmn_garray_extract(self._h, buf, C.byref(h)))
        return self._new(h)

    def get(self, i: int):
        out = C.create_string_buffer(self.group.elem_bytes)
        _check(lib().vmn_garray_get(self._h, C.c_size_t(i), out))
        return self.group.dec_el(out.raw)

    def isMember(self) -> bool:
        ok = C.c_int()
        _check(lib().vmn_garray_is_member(self._h, C.byref(ok)))
        return bool(ok.value)


def expProdMulti(arrays, e: "PRingElementArray", ebits: int = 0) -> list:
    """``expProd`` of several arrays under one exponent array (a ciphertext array's components): one sort."""
    grp = arrays[0].group
    k = len(arrays)
    hs = (C.c_void_p * k)(*[a._h for a in arrays])
    out = C.create_string_buffer(k * grp.elem_bytes)
    _check(lib().vmn_garray_expprod_multi(hs, C.c_size_t(k), e._h, C.c_int(ebits), out))
    eb = grp.elem_bytes
    return [grp.dec_el(out.raw[i * eb:(i + 1) * eb]) for i in range(k)]


def innerProducts(pairs) -> list:
    """``vmn_rarray_inner_products``: [<x, y> mod q for (x, y) in pairs] in one round trip; y = None gives the sum of x."""
    k = len(pairs)
    xs = (C.c_void_p * k)(*[x._h for x, _ in pairs])
    ys = (C.c_void_p * k)(*[(y._h if y is not None else None) for _, y in pairs])
    xb = pairs[0][0].group.exp_bytes
    out = C.create_string_buffer(k * xb)
    _check(lib().vmn_rarray_inner_products(xs, ys, C.c_size_t(k), out))
    return [int.from_bytes(out.raw[i * xb:(i + 1) * xb], "big") for i in range(k)]


class PendingExpProd:
    """``vmn_garray_expprod_multi_begin``: the device part of ``expProdMulti`` is queued; ``finish()`` waits for it and returns
    the k elements.  Device work queued in between runs while the host completes the products."""

    def __init__(self, arrays, e: "PRingElementArray", ebits: int = 0):
        self.group = arrays[0].group
        self.k = len(arrays)
        hs = (C.c_void_p * self.k)(*[a._h for a in arrays])
        self._h = C.c_void_p()
        _check(lib().vmn_garray_expprod_multi_begin(hs, C.c_size_t(self.k), e._h, C.c_int(ebits), C.byref(self._h)))

    def finish(self) -> list:
        if not self._h:
            raise RuntimeError("finish() was already called")
        h, self._h = self._h, None
        nbytes = lib().vmn_pending_bytes(h)
        out = C.create_string_buffer(nbytes)
        _check(lib().vmn_pending_finish(h, out))
        eb = self.group.elem_bytes
        return [self.group.dec_el(out.raw[i * eb:(i + 1) * eb]) for i in range(self.k)]

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vmn_pending_free(self._h)
            self._h = None


class PRingElementArray(_ArrayBase):
    """Device-resident ``PRingElementArray`` / ``PFieldElementArray`` over Z_q."""

    _free_fn = "vmn_rarray_free"

    def size(self) -> int:
        return lib().vmn_rarray_size(self._h)

    def toBytes(self) -> bytes:
        out = C.create_string_buffer(max(1, self.size() * self.group.exp_bytes))
        _check(lib().vmn_rarray_to_be(self._h, out))
        return out.raw[: self.size() * self.group.exp_bytes]

    def toInts(self) -> list:
        return be_to_ints(self.toBytes(), self.group.exp_bytes)

    def toByteTree(self) -> bytes:
        size = lib().vmn_rarray_bytetree_size(self._h)
        out = C.create_string_buffer(size)
        _check(lib().vmn_rarray_to_bytetree(self._h, out))
        return out.raw[:size]

    def _new(self, h) -> "PRingElementArray":
        return PRingElementArray(self.group, h)

    def _binary(self, fn: str, other: "PRingElementArray") -> "PRingElementArray":
        h = C.c_void_p()
        _check(getattr(lib(), fn)(self._h, other._h, C.byref(h)))
        return self._new(h)

    def mul(self, other):
        return self._binary("vmn_rarray_mul", other)

    def add(self, other):
        return self._binary("vmn_rarray_add", other)

    def neg(self):
        h = C.c_void_p()
        _check(lib().vmn_rarray_neg(self._h, C.byref(h)))
        return self._new(h)

    def mulAdd(self, v: int, other: Optional["PRingElementArray"]) -> "PRingElementArray":
        """x.mulAdd(v, y) = x*v + y ; other=None: x*v."""
        h = C.c_void_p()
        _check(lib().vmn_rarray_mul_add(self._h, int_to_be(v, self.group.exp_bytes), other._h if other is not None else None,
                                        C.byref(h)))
        return self._new(h)

    def get(self, i: int) -> int:
        out = C.create_string_buffer(self.group.exp_bytes)
        _check(lib().vmn_rarray_get(self._h, C.c_size_t(i), out))
        return int.from_bytes(out.raw, "big")

    def maxBits(self) -> int:
        """Largest bit length among the entries (``vmn_rarray_max_bits``): what a verifier uses for an exponent array
        it was sent, so that every bit of it counts."""
        out = C.c_int()
        _check(lib().vmn_rarray_max_bits(self._h, C.byref(out)))
        return out.value

    def copyOfRange(self, start: int, end: int) -> "PRingElementArray":
        h = C.c_void_p()
        _check(lib().vmn_rarray_copy_range(self._h, C.c_size_t(start), C.c_size_t(end), C.byref(h)))
        return self._new(h)

    def recLin(self, e: "PRingElementArray"):
        """``b.recLin(e)`` -> (x, d): x0 = b0, xi = x(i-1)*ei + bi, d = x(N-1)."""
        h = C.c_void_p()
        last = C.create_string_buffer(self.group.exp_bytes)
        _check(lib().vmn_rarray_rec_lin(self._h, e._h, C.byref(h), last))
        return self._new(h), int.from_bytes(last.raw, "big")

    def prods(self) -> "PRingElementArray":
        h = C.c_void_p()
        _check(lib().vmn_rarray_prods(self._h, C.byref(h)))
        return self._new(h)

    def _scalar(self, fn: str, *others) -> int:
        out = C.create_string_buffer(self.group.exp_bytes)
        _check(getattr(lib(), fn)(self._h, *[o._h for o in others], out))
        return int.from_bytes(out.raw, "big")

    def innerProduct(self, other) -> int:
        return self._scalar("vmn_rarray_inner_product", other)

    def sum(self) -> int:
        return self._scalar("vmn_rarray_sum")

    def prod(self) -> int:
        return self._scalar("vmn_rarray_prod")

    def permute(self, perm: Sequence[int]) -> "PRingElementArray":
        arr, _keep = _u32_array(perm)
        h = C.c_void_p()
        _check(lib().vmn_rarray_gather(self._h, arr, C.c_size_t(len(perm)), C.byref(h)))
        return self._new(h)

    def shiftPush(self, el: int) -> "PRingElementArray":
        h = C.c_void_p()
        _check(lib().vmn_rarray_shift_push(self._h, int_to_be(el, self.group.exp_bytes), C.byref(h)))
        return self._new(h)

    def equals(self, other) -> bool:
        eq = C.c_int()
        _check(lib().vmn_rarray_equals(self._h, other._h, C.byref(eq)))
        return bool(eq.value)
