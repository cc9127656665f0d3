"""The preprocessor's resize on the GPU (nbc_resize_cubic_u8, models.py:191-198) against the numpy
restatement and the scikit-image 0.18.3 fixtures, bit for bit."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from neuralbarkcalculator_amd import predict as drv
from neuralbarkcalculator_amd.model import FCNResNet50

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def model(built_lib):
    return FCNResNet50("bf16").to(DEV)            # no weights needed for the preprocessor


@pytest.mark.parametrize("shape,out", [((2048, 2048), (1024, 1024)), ((1500, 1100), (1024, 1024)), ((1025, 1025), (1024, 1024)),
                                       ((3000, 512), (1024, 1024)), ((5, 7), (3, 4)), ((1, 9), (4, 4)), ((40, 1030), (64, 64)),
                                       ((333, 777), (100, 50))])
def test_resize_equals_numpy_restatement(model, shape, out):
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    img = rng.integers(0, 256, size=shape + (3,), dtype=np.uint8)
    img[: shape[0] // 3] //= 4                                  # a darker band: clipping bounds below 255
    want = drv.resize_bicubic_reflect(img.astype(np.float32) / np.float32(255), out[0], out[1])
    got = model.resize_cubic_u8(torch.from_numpy(img).to(DEV), out[0], out[1]).cpu().numpy()
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} values differ, max {np.abs(got - want).max()}"


def test_constant_and_extreme_images(model):
    for v in (0, 255, 17):
        img = np.full((70, 90, 3), v, np.uint8)
        got = model.resize_cubic_u8(torch.from_numpy(img).to(DEV), 32, 32).cpu().numpy()
        assert np.array_equal(got, drv.resize_bicubic_reflect(img.astype(np.float32) / np.float32(255), 32, 32))


def test_preprocess_image_with_and_without_device(model):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(1300, 1300, 3), dtype=np.uint8)
    img[:200] = 0                                               # black rows: trim_black has something to do
    a = drv.preprocess_image(img, 1024)
    b = drv.preprocess_image(img, 1024, model)
    assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "preprocess_*.npz"))))
def test_skimage_fixtures_through_the_device(model, path):
    g = np.load(path, allow_pickle=False)
    out = drv.preprocess_image(g["image"], int(g["target"]), model)
    assert out.shape == g["expected"].shape and out.dtype == np.uint8
    assert np.array_equal(out, g["expected"])                   # byte for byte what scikit-image 0.18.3 saved


def test_device_quantisation_and_lit_counts(model):
    """nbc_preprocess_u8: the bytes imsave would write and trim_black's per-row lit counts, from the device."""
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, size=(1300, 1100, 3), dtype=np.uint8)
    img[:150] = 0                                                # black rows
    img[150:300, ::3] = 0                                        # a third of the pixels black
    img[700, :, :2] = 0                                          # two channels out: still lit
    f = drv.resize_bicubic_reflect(img.astype(np.float32) / np.float32(255), 512, 512)
    out, lit = model.preprocess_u8(torch.from_numpy(img).to(DEV), 512, 512)
    assert np.array_equal(out.cpu().numpy(), drv._float_to_u8(f))
    assert np.array_equal(lit.cpu().numpy(), (np.sum(f, axis=-1) > 1e-3).sum(axis=1))
    # and the whole preprocessor: device route == numpy route, trimmed rows included
    a = drv.preprocess_image(img[:, :1100][:1100], 512)
    b = drv.preprocess_image(img[:, :1100][:1100], 512, model)
    assert a.shape == b.shape and a.shape[0] < 512 and np.array_equal(a, b)


def _write_bmp24(path, img, top_down):
    """A 24-bit uncompressed BMP by hand: rows padded to 4 bytes, BGR, bottom-up unless ``top_down`` (negative height)."""
    import struct
    h, w = img.shape[:2]
    stride = (w * 3 + 3) & ~3
    rows = np.zeros((h, stride), dtype=np.uint8)
    rows[:, : w * 3] = img[..., ::-1].reshape(h, w * 3)
    if not top_down:
        rows = rows[::-1]
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + stride * h, 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, -h if top_down else h, 1, 24, 0, stride * h, 2835, 2835, 0, 0))
        f.write(np.ascontiguousarray(rows).tobytes())


@pytest.mark.parametrize("shape,top_down", [((1500, 1301), False), ((1301, 1500), True), ((2048, 2048), False)])
def test_raw_bmp_scan_through_the_device_equals_the_host_decode(tmp_path, model, shape, top_down):
    """The raw-scan route of the folder driver (file -> pinned memory -> device, BGR / bottom-up / padded rows put in
    order there) writes the bytes the host-decode route writes, including a square scan that trim_black cuts."""
    import threading
    rng = np.random.default_rng(shape[0] + shape[1])
    img = rng.integers(0, 256, size=shape + (3,), dtype=np.uint8)
    img[: shape[0] // 5] = 0                                      # a black band (cut by trim_black when the scan is square)
    path = str(tmp_path / "scan.bmp")
    _write_bmp24(path, img, top_down)
    assert np.array_equal(drv._decode_rgb(path), img)             # the host decoder reads the hand-written file
    want = drv.preprocess_image(img, 1024, model)
    got = drv.preprocess_bmp_scan_on_device(path, 1024, model, threading.Lock())
    assert got is not None and got.dtype == np.uint8 and np.array_equal(got, want)
    if shape[0] == shape[1]:
        assert got.shape[0] < 1024                                # trimmed
    # small images and other formats take the host route
    small = str(tmp_path / "small.bmp")
    _write_bmp24(small, img[:600, :700], False)
    assert drv.preprocess_bmp_scan_on_device(small, 1024, model, threading.Lock()) is None
    from PIL import Image
    png = str(tmp_path / "scan.png")
    Image.fromarray(img).save(png)
    assert drv.preprocess_bmp_scan_on_device(png, 1024, model, threading.Lock()) is None
