"""Output stage helpers: Reinhard tonemap (tonemap_composit.hlsl:37-44) and PFM / PNG writers.
Pure Python + numpy + zlib; the reference presents to a swap chain instead (PathTracer.cpp:789-824)."""
import struct
import zlib

import numpy as np


def tonemap_reinhard(hdr):
    """c / (c + 1) per channel on the RGB part of an (h, w, >=3) float image."""
    c = np.asarray(hdr, dtype=np.float32)[..., :3]
    return c / (c + np.float32(1.0))


def write_pfm(path, img):
    """(h, w, 3) float32, row 0 = top; PFM stores bottom row first, little endian (negative scale)."""
    a = np.ascontiguousarray(np.asarray(img, dtype=np.float32)[..., :3][::-1])
    h, w, _ = a.shape
    with open(path, "wb") as f:
        f.write(f"PF\n{w} {h}\n-1.0\n".encode())
        f.write(a.astype("<f4").tobytes())


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(x) for x in f.readline().split())
        scale = float(f.readline())
        a = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return a[::-1].astype(np.float32)


def write_png(path, ldr, gamma=2.2):
    """(h, w, 3) values in [0, 1] -> 8-bit RGB PNG (display gamma applied)."""
    a = np.clip(np.asarray(ldr, dtype=np.float32)[..., :3], 0.0, 1.0) ** np.float32(1.0 / gamma)
    a = (a * 255.0 + 0.5).astype(np.uint8)
    h, w, _ = a.shape
    raw = np.concatenate([np.zeros((h, 1), dtype=np.uint8), a.reshape(h, w * 3)], axis=1).tobytes()   # filter 0 per row

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))
