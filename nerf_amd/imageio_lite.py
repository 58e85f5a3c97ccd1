"""Minimal PNG writer/reader (8-bit gray / RGB / RGBA, no interlace) so render_path can write
its frames without the reference's imageio dependency (run_nerf.py:155)."""
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def imwrite(path, img):
    img = np.asarray(img)
    if img.dtype != np.uint8:
        raise ValueError("imwrite: expected uint8 (use to8b)")
    if img.ndim == 2:
        img = img[..., None]
    h, w, c = img.shape
    color = {1: 0, 3: 2, 4: 6}[c]
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)) \
        + _chunk(b"IDAT", zlib.compress(raw, 6)) + _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def imread(path):
    """Reads back PNGs written by imwrite (filter type 0 rows only)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, color, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if depth != 8 or interlace != 0 or color not in (0, 2, 6):
                raise ValueError("unsupported PNG flavour")
            c = {0: 1, 2: 3, 6: 4}[color]
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = zlib.decompress(idat)
    stride = w * c + 1
    out = np.empty((h, w, c), dtype=np.uint8)
    for y in range(h):
        row = raw[y * stride:(y + 1) * stride]
        if row[0] != 0:
            raise ValueError("PNG row filter not supported by this reader")
        out[y] = np.frombuffer(row[1:], dtype=np.uint8).reshape(w, c)
    return out[..., 0] if c == 1 else out
