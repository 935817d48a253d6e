"""Minimal PNG codec for the load_data / classify front end (rcn.rs:83, 394-398: `ImageReader::open(..).decode()
.grayscale()` then `get_pixel_matrix`, lib.rs:27-41).  The reference delegates to the `image` crate (^0.24.3, unpinned);
this restates the part of its behaviour the path relies on:

  * 8-bit grayscale (+alpha) decodes to Luma8 / LumaA8; `get_pixel_matrix` takes channel 0 and ignores alpha;
  * sub-byte grayscale is expanded to 8 bits, palette images to RGB, like the crate's default transformations;
  * RGB(A) `.grayscale()` -> luma = (2126 R + 7152 G + 722 B) / 10000 in integer arithmetic (image 0.24 `rgb_to_luma`);
  * 16-bit images become Luma16 / LumaA16, which `get_pixel_matrix` rejects (InvalidGrayscaleImageError, errors.rs:1-13).

Non-interlaced only (Adam7 raises).  No PIL in this image; zlib comes from the standard library."""
from __future__ import annotations

import struct
import zlib
from typing import Tuple

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"


class PngError(ValueError):
    pass


class InvalidGrayscaleImageError(PngError):
    """errors.rs:1-13: 'Image provided was not Luma8 (grayscaled image)'"""


def _paeth(a, b, c):
    p = a.astype(np.int32) + b.astype(np.int32) - c.astype(np.int32)
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c)).astype(np.uint8)


def decode(data: bytes) -> Tuple[np.ndarray, int, int]:
    """-> (pixels [H, W, channels] uint8 or uint16, color_type, bit_depth) with palette / sub-byte samples expanded."""
    if data[:8] != _SIG:
        raise PngError("not a PNG file")
    pos, idat, plte, ihdr = 8, [], None, None
    while pos + 8 <= len(data):
        ln, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + ln]
        if len(body) != ln:
            raise PngError("truncated chunk")
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            plte = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + ln
    if ihdr is None or not idat:
        raise PngError("missing IHDR / IDAT")
    W, H, depth, ctype, _, _, interlace = ihdr
    if interlace:
        raise PngError("interlaced PNG not supported")
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype)
    if ch is None or depth not in (1, 2, 4, 8, 16):
        raise PngError("bad colour type / bit depth")
    bpp = max(1, ch * depth // 8)                       # filter unit in bytes
    stride = (W * ch * depth + 7) // 8
    raw = zlib.decompress(b"".join(idat))
    if len(raw) < H * (stride + 1):
        raise PngError("IDAT too short")
    rows = np.zeros((H, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.uint8)
    for y in range(H):
        ft = raw[y * (stride + 1)]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=y * (stride + 1) + 1).copy()
        if ft == 1:
            for i in range(bpp, stride):
                line[i] = (int(line[i]) + int(line[i - bpp])) & 255
        elif ft == 2:
            line = (line.astype(np.uint16) + prev).astype(np.uint8)
        elif ft == 3:
            for i in range(stride):
                left = int(line[i - bpp]) if i >= bpp else 0
                line[i] = (int(line[i]) + ((left + int(prev[i])) >> 1)) & 255
        elif ft == 4:
            for i in range(stride):
                a = int(line[i - bpp]) if i >= bpp else 0
                b = int(prev[i])
                c = int(prev[i - bpp]) if i >= bpp else 0
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                line[i] = (int(line[i]) + pr) & 255
        elif ft != 0:
            raise PngError("bad filter type")
        rows[y] = line
        prev = line
    if depth == 16:
        px = rows.reshape(H, W * ch, 2)
        px = (px[..., 0].astype(np.uint16) << 8 | px[..., 1]).reshape(H, W, ch)
        return px, ctype, depth
    if depth < 8:
        bits = np.unpackbits(rows, axis=1)[:, : W * ch * depth].reshape(H, W * ch, depth)
        vals = bits.dot(1 << np.arange(depth - 1, -1, -1)).astype(np.uint8)
        if ctype == 0:
            vals = (vals.astype(np.uint16) * 255 // ((1 << depth) - 1)).astype(np.uint8)     # expand to 8 bits
        px = vals.reshape(H, W, ch)
    else:
        px = rows.reshape(H, W, ch)
    if ctype == 3:
        if plte is None:
            raise PngError("palette image without PLTE")
        px, ctype = plte[px[..., 0]], 2
    return px, ctype, 8


def to_pixel_matrix_u8(data: bytes) -> np.ndarray:
    """decode().grayscale() + get_pixel_matrix: a [H, W] uint8 array of channel-0 luma (row = y, column = x)."""
    px, ctype, depth = decode(data)
    if depth == 16:
        raise InvalidGrayscaleImageError("InvalidGrayscaleImageError: Image provided was not Luma8 (grayscaled image)")
    if ctype in (0, 4):
        return np.ascontiguousarray(px[..., 0])
    rgb = px[..., :3].astype(np.uint32)
    return ((2126 * rgb[..., 0] + 7152 * rgb[..., 1] + 722 * rgb[..., 2]) // 10000).astype(np.uint8)


def encode_gray(img: np.ndarray) -> bytes:
    """8-bit grayscale PNG (filter 0) -- for building test data sets."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(H))
    return _SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
