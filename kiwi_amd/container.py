"""Section container shared by the raw model, the baked (flat) model dump and test fixtures.

Layout (little endian):  8-byte kind tag, u32 n_sections, u32 reserved, then n entries of
{char name[32]; u64 offset; u64 nbytes}, then the section payloads, each 64-byte aligned.
The C++ reader is ``kiwi_amd/csrc/container.hpp``.
"""
from __future__ import annotations

import os
import struct

import numpy as np

_ENTRY = struct.Struct("<32sQQ")


def write_container(path: str, sections: dict, kind: bytes = b"KAMDSEC1") -> None:
    assert len(kind) == 8
    names = list(sections)
    off = 16 + _ENTRY.size * len(names)
    entries, blobs = [], []
    for n in names:
        a = np.ascontiguousarray(sections[n])
        off = (off + 63) & ~63
        entries.append((n.encode(), off, a.nbytes))
        blobs.append((off, a.tobytes()))
        off += a.nbytes
    # (written beside the target and renamed: a reader -- another test process, another rank -- sees the old file or the new one, never a part of it)
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "wb") as f:
        f.write(kind)
        f.write(struct.pack("<II", len(names), 0))
        for e in entries:
            f.write(_ENTRY.pack(*e))
        for o, b in blobs:
            f.seek(o)
            f.write(b)
        f.truncate(max(off, f.tell()))
    os.replace(tmp, path)


def read_container(path: str) -> tuple[bytes, dict]:
    with open(path, "rb") as f:
        data = f.read()
    kind = data[:8]
    n, _ = struct.unpack_from("<II", data, 8)
    out = {}
    for i in range(n):
        name, off, nb = _ENTRY.unpack_from(data, 16 + i * _ENTRY.size)
        out[name.rstrip(b"\0").decode()] = np.frombuffer(data, "u1", nb, off)
    return kind, out
