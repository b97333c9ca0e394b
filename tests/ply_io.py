"""Test helper: reader for the binary little-endian PLY meshes the reference's tests use (data/tests/bunny.ply,
loaded there through the `ply` shape plugin, src/tests/test_kd.cpp:86-91): float x/y/z vertices, triangle lists."""
import numpy as np


def read(path):
    data = open(path, "rb").read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    header = data[:end].decode("ascii").split("\n")
    assert header[0] == "ply" and header[1].startswith("format binary_little_endian")
    nv = nf = 0
    vprops = []
    cur = None
    for line in header:
        tok = line.split()
        if tok[:2] == ["element", "vertex"]:
            nv = int(tok[2]); cur = "v"
        elif tok[:2] == ["element", "face"]:
            nf = int(tok[2]); cur = "f"
        elif tok and tok[0] == "property" and cur == "v":
            assert tok[1] == "float"
            vprops.append(tok[2])
        elif tok and tok[0] == "property" and cur == "f":
            assert tok[1:4] == ["list", "uchar", "int"]
    verts = np.frombuffer(data, dtype="<f4", count=nv * len(vprops), offset=end).reshape(nv, len(vprops))
    pos = np.ascontiguousarray(verts[:, [vprops.index("x"), vprops.index("y"), vprops.index("z")]], dtype=np.float32)
    off = end + verts.nbytes
    faces = np.frombuffer(data, dtype=np.dtype([("n", "u1"), ("idx", "<i4", (3,))]), count=nf, offset=off)
    assert (faces["n"] == 3).all()
    return pos, np.ascontiguousarray(faces["idx"], dtype=np.uint32)
