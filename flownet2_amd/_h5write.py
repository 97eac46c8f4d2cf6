"""Minimal HDF5 writer (classic layout) for weight files: nested groups of contiguous little-endian float32 datasets.

What libhdf5 1.8 / 1.10 produce with default libver bounds for Net::ToHDF5 (src/caffe/net.cpp:896-950) -- superblock version 0,
version-1 object headers, groups as Symbol Table message -> version-1 B-tree (one level-0 node) -> symbol-table nodes + local heap,
datasets as dataspace v1 + IEEE float32 datatype + contiguous layout v3.  Written from the HDF5 File Format Specification; checked by
tests/test_hdf5.py against libhdf5 itself where one is installed (the build container's /opt/conda/lib) and against this package's
reader everywhere.  Limits: <= 256 links per group (32 symbol-table nodes x 8 entries under one B-tree node), float32 only."""
from __future__ import annotations

import struct
from typing import Dict, Union

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 4, 16                   # libhdf5 defaults: 2 * LEAF_K entries per symbol-table node, 2 * INTERNAL_K children per B-tree node
Tree = Dict[str, Union["Tree", np.ndarray]]


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype: int, body: bytes) -> bytes:
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), 0) + body


def _object_header(msgs) -> bytes:
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body


class _File:
    def __init__(self):
        self.buf = bytearray(96)             # superblock (56 bytes) + root symbol-table entry (40)

    def alloc(self, data: bytes) -> int:
        self.buf += b"\0" * (-len(self.buf) % 8)
        at = len(self.buf)
        self.buf += data
        return at

    def dataset(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a, dtype="<f4")
        raw = a.tobytes()
        addr = self.alloc(raw) if raw else UNDEF
        space = struct.pack("<BBB5x", 1, a.ndim, 0) + b"".join(struct.pack("<Q", d) for d in a.shape)
        # class 1 (floating point) version 1; bit field: little-endian, mantissa normalisation = implied MSB, sign bit 31
        dtype = struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        layout = struct.pack("<BBQQ", 3, 1, addr, len(raw))
        return self.alloc(_object_header([_msg(0x01, space), _msg(0x03, dtype), _msg(0x08, layout)]))

    def group(self, members: Tree) -> tuple:
        """-> (object header address, B-tree address, local heap address)."""
        names = sorted(members, key=lambda s: s.encode("utf-8"))
        if len(names) > 2 * LEAF_K * 2 * INTERNAL_K:
            raise ValueError("more than %d links in one group" % (2 * LEAF_K * 2 * INTERNAL_K))
        child = {}
        for n in names:
            v = members[n]
            child[n] = self.group(v) if isinstance(v, dict) else (self.dataset(v), None, None)
        # local heap: the empty string at offset 0, then the names; no free blocks (free-list head = H5HL_FREE_NULL = 1)
        heap_data = bytearray(8)
        off = {}
        for n in names:
            off[n] = len(heap_data)
            heap_data += _pad8(n.encode("utf-8") + b"\0")
        heap_data_addr = self.alloc(bytes(heap_data))
        heap = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, heap_data_addr))
        # symbol-table nodes of up to 2 * LEAF_K entries, in name order
        keys, kids = [0], []
        for i in range(0, max(1, len(names)), 2 * LEAF_K):
            part = names[i:i + 2 * LEAF_K]
            node = bytearray(b"SNOD" + struct.pack("<BxH", 1, len(part)))
            for n in part:
                hdr, bt, hp = child[n]
                if bt is None:
                    node += struct.pack("<QQII16x", off[n], hdr, 0, 0)
                else:
                    node += struct.pack("<QQIIQQ", off[n], hdr, 1, 0, bt, hp)       # cached symbol-table info of a group
            node += b"\0" * (8 + 2 * LEAF_K * 40 - len(node))
            kids.append(self.alloc(bytes(node)))
            keys.append(off[part[-1]] if part else 0)
        tree = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, len(kids), UNDEF, UNDEF))
        for k, c in zip(keys, kids):
            tree += struct.pack("<QQ", k, c)
        tree += struct.pack("<Q", keys[len(kids)])
        tree += b"\0" * (24 + 2 * INTERNAL_K * 8 + (2 * INTERNAL_K + 1) * 8 - len(tree))
        bt = self.alloc(bytes(tree))
        hdr = self.alloc(_object_header([_msg(0x11, struct.pack("<QQ", bt, heap))]))
        return hdr, bt, heap


def write(path: str, tree: Tree) -> None:
    f = _File()
    hdr, bt, heap = f.group(tree)
    f.buf += b"\0" * (-len(f.buf) % 8)
    eof = len(f.buf)
    sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQIIQQ", 0, hdr, 1, 0, bt, heap)         # root group's symbol-table entry
    assert len(sb) == 96
    f.buf[:96] = sb
    with open(path, "wb") as out:
        out.write(bytes(f.buf))
