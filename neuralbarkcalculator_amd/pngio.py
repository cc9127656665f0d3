"""Minimal PNG writer for the folder driver (8-bit grey or RGB, non-interlaced).

The reference saves its images through scikit-image / PIL (models.py:203,355-356); what its
consumers read back are the PIXEL VALUES, which any conforming PNG reproduces.  PIL's encoder
spends 80-190 ms on a 1024x1024 RGB frame (row-filter heuristics plus deflate), far more than the
whole forward pass, so the driver writes its own files: filter type 0 on every row and one zlib
stream at a chosen level (0 = stored: 7 ms for such a frame; ``zlib`` releases the GIL, so a thread
pool scales).  Files decode with any PNG reader (checked against PIL in tests/test_driver.py).
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_png(a: np.ndarray, level: int = 1) -> bytes:
    """uint8 ``[H,W]`` (grey, PIL mode 'L') or ``[H,W,3]`` (RGB) -> PNG file bytes."""
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
    return _SIG + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + _chunk(b"IEND", b"")


def write_png(path: str, a: np.ndarray, level: int = 1) -> None:
    data = encode_png(a, level)
    with open(path, "wb") as f:
        f.write(data)
