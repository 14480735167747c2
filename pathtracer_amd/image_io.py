"""Output stage helpers: Reinhard tonemap (tonemap_composit.hlsl:37-44) and PFM / PNG writers, and the
image READERS of the scene-ingestion side (8-bit PNG, binary PPM -> RGBA8 texels for pt_scene_set_albedo_texture; the
reference decodes its textures through FANCY's importer, absent, and never samples one: PathTracer.cpp:435-436).
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

    chunk = _png_chunk

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


def _png_chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png_rgba8(path, rgba8):
    """(h, w, 4) uint8 texels as they are (no gamma, no tonemap): the writer for albedo textures (scene_to_obj)."""
    a = np.ascontiguousarray(np.asarray(rgba8, dtype=np.uint8))
    assert a.ndim == 3 and a.shape[2] == 4
    h, w, _ = a.shape
    raw = np.concatenate([np.zeros((h, 1), dtype=np.uint8), a.reshape(h, w * 4)], axis=1).tobytes()
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)))
        f.write(_png_chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(_png_chunk(b"IEND", b""))


def read_png(path):
    """8-bit, non-interlaced PNG (grey, grey + alpha, RGB, palette incl. tRNS, RGBA; all five row filters) -> (h, w, 4)
    uint8, row 0 = top. Values are returned as stored (no gamma step: an albedo texture is sampled as data). Anything else
    (16-bit, interlaced, bit depths below 8) raises ValueError, as does a bad signature, CRC or size."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, ihdr, idat, plte, trns = 8, None, [], None, None
    while pos + 12 <= len(data):
        (n,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if len(body) != n or pos + 12 + n > len(data):
            raise ValueError(f"{path}: truncated {tag!r} chunk")
        if struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] != (zlib.crc32(tag + body) & 0xFFFFFFFF):
            raise ValueError(f"{path}: CRC mismatch in {tag!r} chunk")
        pos += 12 + n
        if tag == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            plte = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif tag == b"tRNS":
            trns = np.frombuffer(body, dtype=np.uint8)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    if ihdr is None or not idat:
        raise ValueError(f"{path}: no IHDR / IDAT")
    w, h, depth, ctype, comp, filt, interlace = ihdr
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype)
    if depth != 8 or channels is None or interlace != 0 or comp != 0 or filt != 0 or w == 0 or h == 0:
        raise ValueError(f"{path}: only 8-bit non-interlaced PNG is read (depth {depth}, colour type {ctype}, interlace {interlace})")
    bpp, stride = channels, w * channels
    raw = zlib.decompress(b"".join(idat))
    if len(raw) != h * (stride + 1):
        raise ValueError(f"{path}: {len(raw)} bytes of image data, expected {h * (stride + 1)}")
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = bytes(stride)
    for y in range(h):
        ft, line = int(rows[y, 0]), rows[y, 1:]
        if ft == 0:
            cur = line.tobytes()
        elif ft == 2:                                      # Up
            cur = (line + np.frombuffer(prev, dtype=np.uint8)).tobytes()            # uint8 addition wraps mod 256
        elif ft == 1:                                      # Sub: a running sum along the row, per channel, mod 256
            cur = np.cumsum(line.reshape(w, bpp), axis=0, dtype=np.uint32).astype(np.uint8).tobytes()
        elif ft in (3, 4):                                 # Average, Paeth: sequential in x — a plain byte loop
            src, c = line.tobytes(), bytearray(stride)
            if ft == 3:
                for i in range(stride):
                    a = c[i - bpp] if i >= bpp else 0
                    c[i] = (src[i] + ((a + prev[i]) >> 1)) & 255
            else:
                for i in range(stride):
                    if i >= bpp:
                        a, b, d = c[i - bpp], prev[i], prev[i - bpp]
                        p0 = a + b - d
                        pa, pb, pc = abs(p0 - a), abs(p0 - b), abs(p0 - d)
                        pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else d)
                    else:
                        pred = prev[i]
                    c[i] = (src[i] + pred) & 255
            cur = bytes(c)
        else:
            raise ValueError(f"{path}: row {y} has filter type {ft}")
        out[y] = np.frombuffer(cur, dtype=np.uint8)
        prev = cur
    px = out.reshape(h, w, channels)
    img = np.full((h, w, 4), 255, dtype=np.uint8)
    if ctype == 0:
        img[..., :3] = px
    elif ctype == 4:
        img[..., :3], img[..., 3] = px[..., :1], px[..., 1]
    elif ctype == 2:
        img[..., :3] = px
    elif ctype == 6:
        img[...] = px
    else:
        if plte is None or int(px.max()) >= len(plte):
            raise ValueError(f"{path}: palette index out of range")
        img[..., :3] = plte[px[..., 0]]
        if trns is not None:
            alpha = np.full(len(plte), 255, dtype=np.uint8)
            alpha[:len(trns)] = trns[:len(plte)]
            img[..., 3] = alpha[px[..., 0]]
    return img


def read_ppm(path):
    """Binary PPM (P6) / PGM (P5), maxval 255 -> (h, w, 4) uint8 with alpha 255."""
    with open(path, "rb") as f:
        data = f.read()
    tok, pos = [], 0
    while len(tok) < 4:                                    # magic, width, height, maxval; '#' comments allowed between
        while pos < len(data) and data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            while pos < len(data) and data[pos:pos + 1] != b"\n":
                pos += 1
            continue
        end = pos
        while end < len(data) and not data[end:end + 1].isspace():
            end += 1
        if end == pos:
            raise ValueError(f"{path}: truncated PNM header")
        tok.append(data[pos:end])
        pos = end
    pos += 1                                               # the single whitespace byte after maxval
    if tok[0] not in (b"P5", b"P6") or int(tok[3]) != 255:
        raise ValueError(f"{path}: only binary P5 / P6 with maxval 255 is read")
    w, h, ch = int(tok[1]), int(tok[2]), 3 if tok[0] == b"P6" else 1
    if w <= 0 or h <= 0 or len(data) - pos < w * h * ch:
        raise ValueError(f"{path}: {len(data) - pos} bytes of pixels, expected {w * h * ch}")
    px = np.frombuffer(data, dtype=np.uint8, count=w * h * ch, offset=pos).reshape(h, w, ch)
    img = np.full((h, w, 4), 255, dtype=np.uint8)
    img[..., :3] = px
    return img


def read_image_rgba8(path):
    """Texture file -> (h, w, 4) uint8 by signature (PNG, PPM / PGM)."""
    with open(path, "rb") as f:
        sig = f.read(8)
    if sig[:8] == b"\x89PNG\r\n\x1a\n":
        return read_png(path)
    if sig[:2] in (b"P5", b"P6"):
        return read_ppm(path)
    raise ValueError(f"{path}: unsupported image format (8-bit PNG and binary PPM / PGM are read)")
