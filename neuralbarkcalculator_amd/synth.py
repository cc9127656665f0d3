"""Deterministic synthetic weights and frames (there are no datasets or checkpoints offline).

Everything is a pure function of integer counters through a 64-bit mixing hash,
evaluated with IEEE float64 add/mul only (no libm), so the same numbers come out on
any machine and numpy version.  Shapes follow SURVEY.md section 8(d):

* frames: uint8 RGB ``H x W x 3`` with bark-like statistics (per-channel mean / std of
  ``models.py:208-209`` modulated by a smooth low-frequency field plus pixel noise), never
  containing a "dark row" so the reference's ``trim_black`` (``models.py:157-166``) would keep
  every row;
* weights: the 326-entry ``fcn_resnet50`` state_dict (``models.py:221-222``), either
  ``"random_init"`` (torchvision's initialisation: Kaiming-normal fan_out convs, BN identity)
  or ``"trained_like"`` (perturbed BN statistics and a calibrated classifier so that the three
  classes all appear in the label mask; random-init gives an all-"Nothing" mask, which would
  make a label-parity test vacuous).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .topology import conv_units, state_dict_spec

DEFAULT_MEAN = (0.7399, 0.6139, 0.4401)   # models.py:208
DEFAULT_STD = (0.1068, 0.1272, 0.1271)    # models.py:209

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


def uniform01(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n float64 in [0,1): element i is a function of (seed, stream, offset+i) only."""
    with np.errstate(over="ignore"):
        base = _mix64(np.uint64(seed) * _G + np.uint64(stream) * _M1 + np.uint64(0x1234567))
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        h = _mix64(base + idx * _G)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: int, n: int) -> np.ndarray:
    """Approximately N(0,1): Irwin-Hall sum of the four 16-bit fields of one hash word
    (integer arithmetic, then one exact scaling)."""
    with np.errstate(over="ignore"):
        base = _mix64(np.uint64(seed) * _G + np.uint64(stream) * _M1 + np.uint64(0x7654321))
        h = _mix64(base + np.arange(n, dtype=np.uint64) * _G)
    m = np.uint64(0xFFFF)
    s = ((h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + (h >> np.uint64(48)))
    # each field is uniform on {0..65535}: mean 32767.5, variance (65536^2 - 1) / 12
    return (s.astype(np.float64) - 131070.0) * (1.7320508075688772 / 65536.0)


# "trained_like" needs no calibration of the classifier: with the perturbed BN statistics below
# the CPU oracle gives all three classes a share of every synthetic frame (1024^2 frame 2:
# 81 % / 9 % / 10 %; 256^2 frame 0: 53 % / 31 % / 16 %), so classifier.4 keeps a zero bias.
TRAINED_LIKE_HEAD_BIAS = (0.0, 0.0, 0.0)


def make_state_dict(kind: str = "trained_like", seed: int = 7) -> Dict[str, np.ndarray]:
    """The 326-entry state_dict as numpy arrays (float32; ``num_batches_tracked`` int64)."""
    if kind not in ("trained_like", "random_init"):
        raise ValueError(f"unknown weight kind {kind!r}")
    units = {u.name: u for u in conv_units()}
    sd: Dict[str, np.ndarray] = {}
    for stream, (key, shape, dtype) in enumerate(state_dict_spec()):
        n = int(np.prod(shape)) if shape else 1
        if key.endswith("num_batches_tracked"):
            sd[key] = np.zeros((), dtype=np.int64)
            continue
        prefix, leaf = key.rsplit(".", 1)
        if prefix in units and leaf == "weight":
            u = units[prefix]
            if u.bn is None:  # classifier.4: nn.Conv2d default init ~ U(-1/sqrt(fan_in), +)
                bound = 1.0 / np.sqrt(u.cin * u.k * u.k)
                v = (uniform01(seed, stream, n) * 2.0 - 1.0) * bound
            else:             # torchvision: kaiming_normal_(mode="fan_out", nonlinearity="relu")
                v = normal(seed, stream, n) * np.sqrt(2.0 / (u.cout * u.k * u.k))
        elif prefix in units and leaf == "bias":
            u = units[prefix]
            if kind == "trained_like":
                v = np.asarray(TRAINED_LIKE_HEAD_BIAS, dtype=np.float64)
            else:
                bound = 1.0 / np.sqrt(u.cin * u.k * u.k)
                v = (uniform01(seed, stream, n) * 2.0 - 1.0) * bound
        else:  # BatchNorm leaf
            if kind == "random_init":
                v = np.ones(n) if leaf in ("weight", "running_var") else np.zeros(n)
            elif leaf == "weight":
                v = 0.5 + uniform01(seed, stream, n)
                if prefix.endswith(".bn3"):   # trained nets keep the residual branch small
                    v = v * 0.35
            elif leaf == "running_var":
                v = 0.5 + uniform01(seed, stream, n)
            else:  # bias, running_mean
                v = normal(seed, stream, n) * 0.1
        sd[key] = v.astype(np.float32).reshape(shape)
    return sd


def _smooth_field(seed: int, stream: int, h: int, w: int, cells: int) -> np.ndarray:
    """Bilinear interpolation of a (cells+1)^2 grid of N(0,1) values to h x w (float64)."""
    g = normal(seed, stream, (cells + 1) * (cells + 1)).reshape(cells + 1, cells + 1)
    ys = (np.arange(h, dtype=np.float64) + 0.5) * (cells / h)
    xs = (np.arange(w, dtype=np.float64) + 0.5) * (cells / w)
    y0 = np.minimum(ys.astype(np.int64), cells - 1)
    x0 = np.minimum(xs.astype(np.int64), cells - 1)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_frame(index: int, h: int = 1024, w: int = 1024) -> np.ndarray:
    """Synthetic processed image ``index`` as uint8 ``[h, w, 3]`` (RGB)."""
    seed = 1000 + index
    shared = (1.2 * _smooth_field(seed, 0, h, w, 4) + 0.9 * _smooth_field(seed, 1, h, w, 16)
              + 0.6 * _smooth_field(seed, 2, h, w, 64))
    img = np.empty((h, w, 3), dtype=np.uint8)
    for c in range(3):
        own = 0.5 * _smooth_field(seed, 3 + c, h, w, 32)
        noise = normal(seed, 8 + c, h * w).reshape(h, w) * 0.35
        v = DEFAULT_MEAN[c] + DEFAULT_STD[c] * (shared + own + noise)
        img[:, :, c] = np.clip(np.floor(255.0 * v + 0.5), 8, 255).astype(np.uint8)
    return img


def normalize_frame(img_u8: np.ndarray, mean=DEFAULT_MEAN, std=DEFAULT_STD) -> np.ndarray:
    """``ToTensor`` then ``Normalize`` (dataset.py:175-186): float32 ``[3, h, w]``."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    s = np.asarray(std, dtype=np.float32)[:, None, None]
    return np.ascontiguousarray(((x - m) / s).astype(np.float32))    # CHW in memory, like ToTensor


def make_input(index: int, h: int = 1024, w: int = 1024) -> np.ndarray:
    """Model input for synthetic frame ``index``: float32 ``[3, h, w]``."""
    return normalize_frame(make_frame(index, h, w))
