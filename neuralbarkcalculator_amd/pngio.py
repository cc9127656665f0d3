"""Minimal PNG writer for the folder driver (8-bit grey or RGB, non-interlaced).

The reference saves its images through scikit-image / PIL (models.py:203,355-356); what its
consumers read back are the PIXEL VALUES, which any conforming PNG reproduces.  PIL's encoder
spends 80-190 ms on a 1024x1024 RGB frame (row-filter heuristics plus deflate), far more than the
whole forward pass, so the driver writes its own files: filter type 0 on every row and one zlib
stream at a chosen level (0 = stored blocks assembled with array copies: 4-7 ms for such a frame; ``zlib``'s
checksums and numpy's copies release the GIL, so a thread pool scales).  Files decode with any PNG reader (checked against PIL in tests/test_driver.py).
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"


def _stored(raw: np.ndarray) -> np.ndarray:
    """The zlib stream of ``raw`` in stored (uncompressed) deflate blocks, assembled with array copies: what
    ``zlib.compress(raw, 0)`` produces semantically, at half its cost (it moves the data through its window)."""
    flat = raw.reshape(-1)
    n = flat.size
    nblk = (n + 65534) // 65535
    out = np.empty(2 + 5 * nblk + n + 4, dtype=np.uint8)
    out[0], out[1] = 0x78, 0x01                     # deflate, 32-KiB window, no preset dictionary, check bits
    pos = 2
    for b in range(nblk):
        lo = b * 65535
        ln = min(65535, n - lo)
        out[pos] = 1 if b == nblk - 1 else 0        # BFINAL, BTYPE = 00 (stored)
        out[pos + 1], out[pos + 2] = ln & 0xFF, ln >> 8
        out[pos + 3], out[pos + 4] = (~ln) & 0xFF, ((~ln) >> 8) & 0xFF
        out[pos + 5: pos + 5 + ln] = flat[lo: lo + ln]
        pos += 5 + ln
    out[pos: pos + 4] = np.frombuffer(struct.pack(">I", zlib.adler32(flat) & 0xFFFFFFFF), dtype=np.uint8)
    return out


def _pieces(a: np.ndarray, level: int):
    """The file as a list of byte strings (written one after the other: the 3-MB IDAT payload of a 1024x1024 RGB
    frame is never copied into a larger string; its CRC runs over tag and payload without joining them)."""
    if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
        raise ValueError("encode_png takes uint8 [H,W] or [H,W,3]")
    h, w = a.shape[:2]
    if h < 1 or w < 1:
        raise ValueError("empty image")
    ch = 1 if a.ndim == 2 else 3
    raw = np.empty((h, w * ch + 1), dtype=np.uint8)
    raw[:, 0] = 0                                   # filter type 0 (None) on every scanline
    raw[:, 1:] = a.reshape(h, w * ch)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 0 if ch == 1 else 2, 0, 0, 0)
    idat = _stored(raw) if level == 0 else zlib.compress(raw, level)      # the array's buffer: no intermediate bytes object
    out = [_SIG]
    for tag, data in ((b"IHDR", ihdr), (b"IDAT", idat), (b"IEND", b"")):
        out += [struct.pack(">I", len(data)), tag, data, struct.pack(">I", zlib.crc32(data, zlib.crc32(tag)) & 0xFFFFFFFF)]
    return out


def encode_png(a: np.ndarray, level: int = 1) -> bytes:
    """uint8 ``[H,W]`` (grey, PIL mode 'L') or ``[H,W,3]`` (RGB) -> PNG file bytes."""
    return b"".join(_pieces(a, level))


def write_png(path: str, a: np.ndarray, level: int = 1) -> None:
    with open(path, "wb") as f:
        f.writelines(_pieces(a, level))
