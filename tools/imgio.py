"""tiny PNG writer (zlib only) for eyeballing films"""
import struct
import zlib
import numpy as np


def write_png(path, rgb):
    """rgb: float [H,W,3] linear radiance -> sRGB-ish 8 bit"""
    im = (np.clip(np.nan_to_num(rgb), 0, 1) ** (1 / 2.2) * 255).astype(np.uint8)
    h, w, _ = im.shape
    raw = b"".join(b"\x00" + im[y].tobytes() for y in range(h))

    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
