"""Minimal PNG writer/reader (8-bit gray / RGB / RGBA and 16-bit variants on read, no interlace) so
render_path can write its frames without the reference's imageio dependency (run_nerf.py:155) and
the loaders can read colour images and 16-bit depth maps (load_blender.py:112-121)."""
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
    """8- or 16-bit gray / gray+alpha / RGB / RGBA, non-interlaced PNGs with any scanline filter
    (what the blender and LLFF data sets ship; 16-bit gray is the depth-map format whose invalid
    readings are 0 and 65535) -> uint8 or uint16 [H,W] or [H,W,C].  Other files (JPEG, palette or
    interlaced PNGs) raise ValueError naming the flavour."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        kind = "a JPEG (not supported: convert the data set to PNG)" if data[:2] == b"\xff\xd8" else "not a PNG"
        raise ValueError(f"{path}: {kind}")
    pos, idat, w = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, color, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if depth not in (8, 16) or interlace != 0 or color not in (0, 2, 4, 6):
                raise ValueError(f"{path}: unsupported PNG flavour (depth {depth}, colour type {color}, interlace {interlace})")
            c = {0: 1, 2: 3, 4: 2, 6: 4}[color]
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    if w is None:
        raise ValueError(f"{path}: no IHDR chunk")
    raw = zlib.decompress(b"".join(idat))
    bpp = c * depth // 8                          # bytes per pixel: what the scanline filters step by
    stride = w * bpp
    if len(raw) != h * (stride + 1):
        raise ValueError(f"{path}: truncated image data")
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(h, stride + 1)
    if not rows[:, 0].any():                     # filter type 0 everywhere: nothing to undo
        out = rows[:, 1:].reshape(h, stride).copy()
    else:
        import ctypes
        from . import _native as nv
        out = np.empty((h, stride), dtype=np.uint8)
        src = np.ascontiguousarray(rows)
        nv.check(nv.lib().nerf_png_unfilter(src.ctypes.data_as(ctypes.c_void_p), h, stride, bpp,
                                            out.ctypes.data_as(ctypes.c_void_p)), "nerf_png_unfilter")
    if depth == 16:                              # samples are big-endian
        out = out.view(">u2").astype(np.uint16)
    out = out.reshape(h, w, c)
    return out[..., 0] if c == 1 else out
