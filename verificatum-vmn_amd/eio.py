"""eio.py — the byte-tree wire format of the reference (``com.verificatum.eio.ByteTree*``, VCR; the
reference tree shows it only through its data: SURVEY.md App. D, decoded from the marshalled group at
``demo/mixnet/benchmarks/bench_config:43``).

    node :  00 | uint32_be(#children) | children...
    leaf :  01 | uint32_be(#bytes)    | bytes

Integers are big-endian two's complement, fixed width within a group.  A ``PGroupElementArray`` /
``PRingElementArray`` is a node of N leaves of the group's byte width; that framing is produced and
parsed on the GPU (``vmn_garray_to_bytetree`` / ``vmn_garray_from_bytetree``) so the arrays cross the
boundary in the reference's own format.  This module handles the small host-side trees (scalars,
containers, marshalled objects).
"""
from __future__ import annotations

from typing import List, Tuple, Union

Tree = Union[bytes, list]      # leaf = bytes, node = list of trees


def encode(tree: Tree) -> bytes:
    if isinstance(tree, (bytes, bytearray)):
        return b"\x01" + len(tree).to_bytes(4, "big") + bytes(tree)
    out = bytearray(b"\x00" + len(tree).to_bytes(4, "big"))
    for child in tree:
        out += encode(child)
    return bytes(out)


def decode(buf: bytes, pos: int = 0) -> Tuple[Tree, int]:
    """Parse one tree starting at ``pos``; returns (tree, next position).  Raises ValueError on
    malformed input (the reference's EIOException)."""
    if pos + 5 > len(buf):
        raise ValueError("EIOException: truncated byte tree header")
    tag = buf[pos]
    count = int.from_bytes(buf[pos + 1:pos + 5], "big")
    pos += 5
    if tag == 1:
        if pos + count > len(buf):
            raise ValueError("EIOException: truncated leaf")
        return bytes(buf[pos:pos + count]), pos + count
    if tag != 0:
        raise ValueError("EIOException: unknown byte tree tag %d" % tag)
    children: List[Tree] = []
    for _ in range(count):
        child, pos = decode(buf, pos)
        children.append(child)
    return children, pos


def int_leaf(x: int, nbytes: int) -> bytes:
    return int(x).to_bytes(nbytes, "big", signed=True)


def leaf_int(b: bytes) -> int:
    return int.from_bytes(b, "big", signed=True)


def unmarshal_modpgroup(buf: bytes):
    """Marshalled ``ModPGroup``: node(leaf(class name), node(p, q, g, int32 encoding)).
    Returns (p, q, g, encoding, byte width)."""
    tree, end = decode(buf)
    if end != len(buf) or not isinstance(tree, list) or len(tree) != 2:
        raise ValueError("EIOException: not a marshalled object")
    name, fields = tree
    if name != b"com.verificatum.arithm.ModPGroup" or not isinstance(fields, list) or len(fields) != 4:
        raise ValueError("EIOException: not a ModPGroup")
    p, q, g = (leaf_int(x) for x in fields[:3])
    return p, q, g, leaf_int(fields[3]), len(fields[0])
