"""Test helper: an independent reader / writer for Mitsuba's `.serialized` mesh container (pure Python + zlib),
following TriMesh::TriMesh(Stream *, int) and TriMesh::serialize(Stream *) (src/librender/trimesh.cpp:156-236,
:748-790).  Used to cross-check the product's C++ loader (mtsgpu_load_serialized) and to write test files."""
import struct
import zlib
import numpy as np

HAS_NORMALS, HAS_TEXCOORDS, HAS_COLORS, FACE_NORMALS, SINGLE, DOUBLE = 0x1, 0x2, 0x8, 0x10, 0x1000, 0x2000


def shape_count(data):
    return struct.unpack_from("<I", data, len(data) - 4)[0]


def read(path, index=0):
    data = open(path, "rb").read()
    pos = 0
    if index != 0:
        count = shape_count(data)
        if index < 0 or index > count:
            raise ValueError("shape index out of range")
        pos = struct.unpack_from("<I", data, len(data) - 4 * (1 + count - index))[0]
    fmt, ver = struct.unpack_from("<HH", data, pos)
    if fmt != 0x041C or ver != 3:
        raise ValueError("bad header")
    raw = zlib.decompressobj().decompress(data[pos + 4:])
    flags, nv, nt = struct.unpack_from("<IQQ", raw, 0)
    off = 20
    ft = np.float64 if flags & DOUBLE else np.float32

    def take(n, dtype):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, count=n, offset=off)
        off += a.nbytes
        return a
    out = {"flags": flags}
    out["positions"] = take(3 * nv, ft).astype(np.float32).reshape(-1, 3)
    out["normals"] = take(3 * nv, ft).astype(np.float32).reshape(-1, 3) if flags & HAS_NORMALS else None
    out["texcoords"] = take(2 * nv, ft).astype(np.float32).reshape(-1, 2) if flags & HAS_TEXCOORDS else None
    out["colors"] = take(3 * nv, ft).astype(np.float32).reshape(-1, 3) if flags & HAS_COLORS else None
    out["triangles"] = take(3 * nt, np.uint32).reshape(-1, 3)
    return out


def write(path, meshes, double=False):
    """meshes: list of dicts with positions, triangles and optionally normals / texcoords / colors / face_normals;
    writes the offset table the way the importer does (one uint32 per shape, then the count)"""
    blob, offsets = b"", []
    ft = "<f8" if double else "<f4"
    for m in meshes:
        offsets.append(len(blob))
        flags = DOUBLE if double else SINGLE
        body = b""
        pos = np.asarray(m["positions"]); tri = np.asarray(m["triangles"], dtype="<u4")
        body += pos.astype(ft).tobytes()
        for key, bit in (("normals", HAS_NORMALS), ("texcoords", HAS_TEXCOORDS), ("colors", HAS_COLORS)):
            if m.get(key) is not None:
                flags |= bit
                body += np.asarray(m[key]).astype(ft).tobytes()
        if m.get("face_normals"):
            flags |= FACE_NORMALS
        body += tri.tobytes()
        head = struct.pack("<IQQ", flags, pos.reshape(-1, 3).shape[0], tri.reshape(-1, 3).shape[0])
        blob += struct.pack("<HH", 0x041C, 3) + zlib.compress(head + body)
    table = b"".join(struct.pack("<I", o) for o in offsets) + struct.pack("<I", len(offsets))
    open(path, "wb").write(blob + table)
