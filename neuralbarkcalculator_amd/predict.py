"""Folder-level batch prediction: the caller side of the hot path, sharded over the GPUs of a node.

Reproduces the file contract of the reference's inference entry point
(/root/reference/src/bark_calculator/predict.py:10-58 and models.py:230-364) around the
accelerated model call:

* input  ``ROOT/samples/<wood_type>/*.{bmp,png,...}``; wood types and order of ``dataset.py:50-58``
  (``epinette_gelee, epinette_non_gelee, sapin``; file names sorted; ``"bmp" -> "png"`` in the
  output name, every occurrence, like ``str.replace`` there);
* ``ROOT/processed/samples/<wood_type>/<name>.png`` (``models.py:173-203``: the bicubic resize to
  1024 x 1024 of larger images, ``trim_black`` on square ones; host-side numpy, pinned by
  scikit-image 0.18.3 fixtures);
* ``ROOT/results/outputs/<wood_type>/<name>.png``: uint8 {0,127,255} mode 'L' (``models.py:349-356``);
* ``ROOT/results/final_stats.csv``: tab separated, the reference's 7-name header and 6-value rows
  (``models.py:252-255,315-332,360-364``: ``img_size`` is dropped by the re-initialisation at
  ``models.py:321``; reproduced verbatim).  The matplotlib figure of ``models.py:280-347``
  (about 10 s per image) is not produced.

Multi-GPU (one process per GPU, ``torch.distributed`` over RCCL; ``--gpus N`` starts the ranks): the
sorted list is cut into contiguous, pixel-balanced shards; rank 0 alone reads the checkpoint and
broadcasts the packed weights; every rank fills int64 rows ``(global_idx, H, W, count_1, count_2)``
which one ``all_gather`` brings to rank 0 for the CSV.  No collective sits in the per-image path.
"""
from __future__ import annotations

import argparse
import csv
import os
import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np

WOOD_TYPES = ["epinette_gelee", "epinette_non_gelee", "sapin"]            # dataset.py:50
IMG_EXTENSIONS = [".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", "webp"]  # dataset.py:77-79
MM2_PER_PIXEL = 3.6 * 3.6                                                  # models.py:210
CSV_HEADER = ["Name", "Type", "Image Size", "Output Bark %", "Bark area (mm^2)",
              "Output Node %", "Node area (mm^2)"]                         # models.py:252-255
ROW_WIDTH = 5                                                              # (global_idx, H, W, count_1, count_2)


def generate_folders(root: str, only_preprocess: bool = False) -> None:
    """predict.py:10-48."""
    present = os.listdir(os.path.join(root, "samples"))
    wood_types = [w for w in WOOD_TYPES if w in present]
    for w in wood_types:
        os.makedirs(os.path.join(root, "processed", "samples", w), exist_ok=True)
    if not only_preprocess:
        for level in ("combined_images", "outputs"):
            for w in wood_types:
                os.makedirs(os.path.join(root, "results", level, w), exist_ok=True)


def list_images(dir_: str) -> List[Tuple[str, str, str]]:
    """(sample_path, output_name, wood_type) in the order of dataset.py:41-68."""
    samples = os.path.join(dir_, "samples")
    if not os.path.isdir(samples):
        raise IOError("Root folder should have a 'samples' subfolder !")   # dataset.py:45-46
    out = []
    for wood in WOOD_TYPES:
        d = os.path.join(samples, wood)
        for _, _, fnames in sorted(os.walk(d)):
            for fname in sorted(fnames):
                if any(fname.lower().endswith(e) for e in IMG_EXTENSIONS):
                    out.append((os.path.join(d, fname), fname.replace("bmp", "png"), wood))
    return out


def trim_black(image: np.ndarray) -> np.ndarray:
    """models.py:157-166 on a float HWC image in [0,1]: drop leading/trailing rows in which 15 % or
    more of the pixels are black (channel sum <= 1e-3)."""
    lit = np.sum(image, axis=-1) > 1e-3
    clear = np.mean(lit, axis=-1) > 0.85
    first = int(np.argmax(clear))
    last = image.shape[0] - int(np.argmax(clear[::-1]))
    return image[first:last]


def _cubic(x, f0, f1, f2, f3):
    """scikit-image's cubic_interpolation (Catmull-Rom, a = -0.5): values at -1, 0, 1, 2; x in [0, 1].
    Evaluated in the image's own float type throughout, in the order its Cython source is written
    (``f1 + 0.5*x*(f2 - f0 + x*(2*f0 - 5*f1 + 4*f2 - f3 + x*(3*(f1 - f2) + f3 - f0)))``), one rounding per
    operation: this is what the compiled _warp_fast does for a float32 image (checked value for value
    against scikit-image 0.18.3, tests/golden/preprocess_*.npz ``float32``)."""
    t = x.dtype.type
    return f1 + t(0.5) * x * (f2 - f0 + x * (t(2.0) * f0 - t(5.0) * f1 + t(4.0) * f2 - f3 + x * (t(3.0) * (f1 - f2) + f3 - f0)))


def _reflect(i: np.ndarray, n: int) -> np.ndarray:
    """numpy.pad 'reflect' indexing (mirror without repeating the edge)."""
    if n == 1:
        return np.zeros_like(i)
    period = 2 * (n - 1)
    i = np.mod(i, period)
    return np.where(i >= n, period - i, i)


def resize_bicubic_reflect(image: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """``skimage.transform.resize(image, (out_h, out_w), order=3, mode='reflect',
    anti_aliasing=False)`` (models.py:194-198) for a float HWC image, as scikit-image 0.18.3 computes it:
    ``resize`` builds the metric transform ``[[fx, 0, fx/2 - 1/2], [0, fy, fy/2 - 1/2]]`` (f = in / out) in
    double, ``warp`` casts it to the image's float type and ``_warp_fast`` evaluates everything in that
    type: the sample position of output pixel i is ``f * i + (f/2 - 1/2)`` (one product, one sum, each
    rounded), its fractional part is taken in the same type, the 4 x 4 taps (reflected borders) go
    through ``_cubic`` row-wise then column-wise, and the result is clipped to the input range
    (``clip=True``).  Bit-identical to scikit-image's output on the committed fixtures, integer and
    non-integer zoom factors alike."""
    h, w = image.shape[:2]
    dt = image.dtype if image.dtype in (np.float32, np.float64) else np.float64
    t = np.dtype(dt).type
    img = image.astype(dt, copy=False)
    fy, fx = h / out_h, w / out_w                                    # Python doubles, like resize's `factors`
    ry = t(fy) * np.arange(out_h, dtype=dt) + t(fy * 0.5 - 0.5)
    rx = t(fx) * np.arange(out_w, dtype=dt) + t(fx * 0.5 - 0.5)
    y0 = np.floor(ry).astype(np.int64)
    x0 = np.floor(rx).astype(np.int64)
    ty = (ry - y0.astype(dt)).reshape(-1, 1, 1)
    tx = (rx - x0.astype(dt)).reshape(1, -1, 1)
    cols = [_reflect(x0 + k - 1, w) for k in range(4)]
    fr = []
    for k in range(4):
        rows = img[_reflect(y0 + k - 1, h)]
        fr.append(_cubic(tx, *[rows[:, c] for c in cols]))
    out = _cubic(ty, *fr)
    assert out.dtype == dt
    return np.clip(out, img.min(), img.max())


def _float_to_u8(image: np.ndarray) -> np.ndarray:
    """``skimage.io.imsave`` of a float image in [0, 1] (models.py:203) goes through imageio's
    ``image_as_uint``: ``uint8(float64(x) * 255 + 0.499999999)`` (exact halves round down)."""
    return np.clip(image.astype(np.float64) * 255.0 + 0.499999999, 0, 255).astype(np.uint8)


# uint8 -> ToTensor (float32 / 255) -> imsave's float -> uint8: a 256-entry table (it is the identity, which
# tests/test_driver.py asserts; the table keeps the code honest should a platform's float division differ)
_U8_ROUND_TRIP = _float_to_u8(np.arange(256, dtype=np.uint8).astype(np.float32) / np.float32(255))
_U8_ROUND_TRIP_IS_IDENTITY = bool(np.array_equal(_U8_ROUND_TRIP, np.arange(256, dtype=np.uint8)))


def _trim_rows(clear: np.ndarray) -> Tuple[int, int]:
    """models.py:162-166: first / one-past-last row of the ``clear`` (enough lit pixels) flags."""
    first = int(np.argmax(clear))
    last = clear.shape[0] - int(np.argmax(clear[::-1]))
    return first, last


def preprocess_image(img_u8: np.ndarray, target_size: int = 1024, model=None) -> np.ndarray:
    """models.py:191-203 for one decoded RGB image: ToTensor (u8 -> float32 / 255), resize to
    ``target_size`` x ``target_size`` when either side is larger, ``trim_black`` when square,
    float -> uint8 like ``skimage.io.imsave`` does through imageio.  Byte-identical to what scikit-image
    0.18.3 writes (tests/golden/preprocess_*.npz).  Three routes, same bytes:

    * no resize needed: the float round trip maps every byte to itself and a pixel is "lit" (float32 channel
      sum > 1e-3) exactly when one of its bytes is non-zero, so the image is trimmed as uint8;
    * resize with ``model`` (an ``FCNResNet50`` on a device): resize, float -> uint8 and the per-row lit
      counts on the device (``nbc_preprocess_u8``), the row trim here;
    * resize without a device: the numpy restatement (about 1 s for a 4096 x 4096 scan)."""
    if max(img_u8.shape[:2]) <= target_size:
        out = img_u8 if _U8_ROUND_TRIP_IS_IDENTITY else _U8_ROUND_TRIP[img_u8]
        if out.shape[0] == out.shape[1]:
            lit = (out[..., 0] | out[..., 1] | out[..., 2]) != 0     # == out.any(axis=-1), ten times faster
            first, last = _trim_rows(np.mean(lit, axis=-1) > 0.85)
            out = out[first:last]
        return np.ascontiguousarray(out)
    if model is not None:
        import torch
        dev_img = torch.from_numpy(np.ascontiguousarray(img_u8)).to(model.device)
        out_dev, lit_dev = model.preprocess_u8(dev_img, target_size, target_size)
        out, lit = out_dev.cpu().numpy(), lit_dev.cpu().numpy()
        first, last = _trim_rows(lit / np.float64(target_size) > 0.85)          # np.mean of booleans: float64 count / n
        return np.ascontiguousarray(out[first:last])
    image = resize_bicubic_reflect(img_u8.astype(np.float32) / np.float32(255), target_size, target_size)
    if image.shape[0] == image.shape[1]:
        image = trim_black(image)
    return _float_to_u8(image)


def _bmp24_layout(head: bytes, size: int):
    """(pixel offset, width, rows, row stride, bottom_up) of an uncompressed 24-bit BMP from its first 54 bytes and
    the file size; None for any other flavour."""
    import struct
    if len(head) < 54 or head[:2] != b"BM":
        return None
    off, = struct.unpack_from("<I", head, 10)
    hdr, w, h, planes, bpp, comp = struct.unpack_from("<IiiHHI", head, 14)
    if hdr < 40 or planes != 1 or bpp != 24 or comp != 0 or w <= 0 or h == 0:
        return None
    rows, stride = abs(h), (w * 3 + 3) & ~3
    if off + stride * rows > size:
        return None
    return off, w, rows, stride, h > 0


def _decode_bmp24(buf: bytes):
    """Uncompressed 24-bit BMP (what the scanner writes, predict.py:15-17) straight into an RGB array: three
    strided numpy copies (which drop the GIL) instead of PIL's decoder loop; None for any other flavour."""
    lay = _bmp24_layout(buf[:54], len(buf))
    if lay is None:
        return None
    off, w, rows, stride, bottom_up = lay
    a = np.frombuffer(buf, np.uint8, stride * rows, off).reshape(rows, stride)[:, : w * 3].reshape(rows, w, 3)
    if bottom_up:
        a = a[::-1]                                  # bottom-up rows
    out = np.empty((rows, w, 3), dtype=np.uint8)
    out[..., 0], out[..., 1], out[..., 2] = a[..., 2], a[..., 1], a[..., 0]     # BGR -> RGB
    return out


def _decode_rgb(path: str) -> np.ndarray:
    """pil_loader (dataset.py:82-90): the file as an RGB uint8 array (own, contiguous copy)."""
    import io
    from PIL import Image
    with open(path, "rb") as f:
        buf = f.read()
    img = _decode_bmp24(buf)
    if img is None:
        img = np.array(Image.open(io.BytesIO(buf)).convert("RGB"))
    return img


_pinned = threading.local()      # one pinned read buffer per pool thread (allocated on first use, grown on demand)


def preprocess_bmp_scan_on_device(path: str, target_size: int, model, lock) -> Optional[np.ndarray]:
    """models.py:173-203 for one raw scan that needs the resize, without touching its pixels on the host: the file
    is read straight into pinned memory, the pixel array goes to the device as the scanner stored it (BGR,
    bottom-up, padded rows), is put into RGB top-down order there and runs through ``nbc_preprocess_u8``; only the
    1024 x 1024 result comes back.  Same bytes as ``preprocess_image(_decode_rgb(path), target_size, model)``; a
    4096 x 4096 scan takes 20-30 ms of a pool thread instead of 150.  None when the file is not an uncompressed
    24-bit BMP or is small enough to need no resize (the caller then takes the host route)."""
    import torch
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        lay = _bmp24_layout(f.read(54), size)
        if lay is None or max(lay[1], lay[2]) <= target_size:
            return None
        off, w, rows, stride, bottom_up = lay
        buf = getattr(_pinned, "buf", None)
        if buf is None or buf.numel() < size:
            buf = _pinned.buf = torch.empty(size + (size >> 3), dtype=torch.uint8).pin_memory()
        f.seek(0)
        if f.readinto(buf.numpy()[:size]) != size:
            return None
    with lock:                                       # one context, one stream: device work of the pool is serialised
        raw = buf[off: off + stride * rows].to(model.device, non_blocking=True)
        img = raw.view(rows, stride)[:, : w * 3].reshape(rows, w, 3)
        if bottom_up:
            img = img.flip(0)
        img = img.flip(2).contiguous()               # BGR -> RGB
        out_dev, lit_dev = model.preprocess_u8(img, target_size, target_size)
        out, lit = out_dev.cpu().numpy(), lit_dev.cpu().numpy()      # .cpu() waits for the stream: buf is free again
    first, last = _trim_rows(lit / np.float64(target_size) > 0.85)
    return np.ascontiguousarray(out[first:last])


def _host_workers() -> int:
    return max(1, min(32, int(os.environ.get("NBC_HOST_WORKERS", "16"))))


def _png_level(kind: str) -> int:
    """zlib level of the PNGs the driver writes.  The files carry pixel values (models.py:203,349-356 pin
    values, not bytes): processed frames default to stored (level 0: noise-like photographs barely
    compress and deflate costs 100 ms per 1024x1024 frame), label maps to level 1 (a few ms, 20x smaller)."""
    return int(os.environ.get("NBC_PNG_LEVEL_" + kind.upper(), "0" if kind == "processed" else "1"))


def preprocess_images(root: str, target_size: int = 1024, model=None) -> None:
    """models.py:173-203 as a stand-alone pass (``--only_preprocess``): decode, resize / trim, save as PNG
    under processed/, on a thread pool; a device resize, when ``model`` is given, is serialised."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from .pngio import write_png
    lock = threading.Lock()

    def one(item):
        path, name, wood = item
        out = preprocess_bmp_scan_on_device(path, target_size, model, lock) if model is not None else None
        if out is None:
            img = _decode_rgb(path)
            if model is not None and max(img.shape[:2]) > target_size:
                with lock:
                    out = preprocess_image(img, target_size, model)
            else:
                out = preprocess_image(img, target_size)
        write_png(os.path.join(root, "processed", "samples", wood, name), out, _png_level("processed"))

    with ThreadPoolExecutor(max_workers=_host_workers()) as pool:
        list(pool.map(one, list_images(root)))


class NonFiniteLogits(RuntimeError):
    """A forward produced NaN / infinite logits: in f16x2 mode an activation beyond f16's range (or NaN/inf weights) -- or the
    packer reported weights the f16 pieces cannot carry at f32 grade (``FCNResNet50.pack_flags``), before any forward ran."""


class AbandonMarker:
    """How the ranks of one node tell each other that an f16x2 run is being abandoned: a file under ``results/`` that the
    rank that sees the non-finite word creates and every rank looks for once per window of images (``os.path.exists``: no
    collective, so ranks with different numbers of windows cannot wait for each other).  The folder driver's ranks share a
    node (``--gpus N`` starts them on this one) and the folder's file system with it.  Rank 0 clears a stale marker before
    the start barrier and the final one after the flag all-reduce."""

    def __init__(self, root: str):
        self.path = os.path.join(root, "results", ".f16x2_abandoned")

    def set(self):
        try:
            os.makedirs(os.path.dirname(self.path), exist_ok=True)
            open(self.path, "w").close()
        except OSError:
            pass                                         # the flag all-reduce at the end still tells every rank

    def is_set(self) -> bool:
        return os.path.exists(self.path)

    def clear(self):
        try:
            os.remove(self.path)
        except OSError:
            pass


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """Round-robin shard: images r, r+W, r+2W, ... (frames of equal size: bench.py)."""
    return list(range(rank, n, world))


def shard_by_pixels(pixels: Sequence[int], world: int) -> List[List[int]]:
    """Contiguous, pixel-balanced shards of the sorted image list (folders of height-trimmed
    or differently sized scans): image i goes to the rank whose share of the total pixel count contains the
    midpoint of i's own span.  Every rank gets a contiguous range; ranges are empty only when there are
    fewer images than ranks."""
    total = float(sum(pixels))
    shards: List[List[int]] = [[] for _ in range(world)]
    acc = 0.0
    for i, p in enumerate(pixels):
        mid = acc + 0.5 * p
        r = min(world - 1, int(mid * world / total)) if total > 0 else i % world
        shards[r].append(i)
        acc += p
    return shards


def stats_row(name: str, wood: str, h: int, w: int, count_1: int, count_2: int) -> List[str]:
    """One CSV row, float32 arithmetic and '{:.5f}' formatting of models.py:321-332."""
    row = [name, wood]
    pixels = np.float32(h * w)
    for c in (count_1, count_2):
        frac = np.float32(c) / pixels                 # (outputs == c).float().mean(): exact sum / N in f32
        row.append("{:.5f}".format(float(frac * np.float32(100))))
        row.append("{:.5f}".format(float(np.float32(c) * np.float32(MM2_PER_PIXEL))))
    return row


def write_stats_csv(path: str, rows: Sequence[Sequence[str]]) -> None:
    with open(path, "w") as f:                       # models.py:360-364 (no newline='' there either)
        csv.writer(f, delimiter="\t").writerows([CSV_HEADER] + [list(r) for r in rows])


def label_png(labels: np.ndarray) -> np.ndarray:
    """models.py:349-353: uint8 map with Bark = 127, Node = 255."""
    out = np.zeros(labels.shape, dtype=np.uint8)
    out[labels == 1] = 127
    out[labels == 2] = 255
    return out


def gather_rows(local_rows: np.ndarray, n_total: int, world: int, dist=None, device=None, cap: int = None) -> np.ndarray:
    """all_gather of fixed-size per-rank row buffers; returns the rows sorted by global index.
    ``local_rows``: int64 [k, ROW_WIDTH] with k <= ``cap`` (default ceil(n_total / world), the round-robin
    bound; pixel-balanced shards pass the largest shard's size)."""
    import torch
    if cap is None:
        cap = (n_total + world - 1) // world if n_total else 0
    buf = torch.full((max(cap, 1), ROW_WIDTH), -1, dtype=torch.int64)
    if len(local_rows):
        buf[: len(local_rows)] = torch.from_numpy(np.asarray(local_rows, dtype=np.int64))
    if dist is None or world == 1:
        allrows = buf
    else:
        if device is not None:
            buf = buf.to(device)
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        allrows = torch.cat(parts).cpu()
    allrows = allrows.numpy()
    allrows = allrows[allrows[:, 0] >= 0]
    return allrows[np.argsort(allrows[:, 0], kind="stable")]


def plan_items(root: str) -> List[dict]:
    """What the reference's two passes end up predicting, in its order: every file that will exist under
    processed/samples/<wood>/ once the preprocessor has run (models.py:173-189 writes
    ``fname.replace("bmp", "png")`` for each sample, a later sample of the same output name overwriting an
    earlier one; files already there stay), listed like dataset.py:41-68 lists them."""
    by_key = {}
    for path, name, wood in list_images(root):
        by_key[(wood, name)] = {"name": name, "wood": wood, "src": path}
    pdir = os.path.join(root, "processed")
    if os.path.isdir(os.path.join(pdir, "samples")):
        for path, name, wood in list_images(pdir):
            key = (wood, name.replace("bmp", "png"))
            if key not in by_key:                      # stale processed file without a sample: predicted as it is
                by_key[key] = {"name": key[1], "wood": wood, "src": None, "processed": path}
    order = {w: i for i, w in enumerate(WOOD_TYPES)}
    items = sorted(by_key.values(), key=lambda d: (order[d["wood"]], d["name"]))
    for d in items:
        d.setdefault("processed", os.path.join(pdir, "samples", d["wood"], d["name"]))
    return items


def predict_folder(root: str, model_path: str = "./best_model.pt", precision: str = "fp32",
                   exclude_nodes: bool = False, small_zones: bool = True, device_index: int = None,
                   batch: int = None, window: int = 64, target_size: int = 1024, autotune: bool = False, calibrate: bool = True,
                   streams: int = None) -> dict:
    """predict.py:51-58 + models.py:230-364 with the model call on the MI355X path.

    One pass per image instead of the reference's two (preprocess everything, then predict everything):
    a rank decodes and preprocesses its own images on a host thread pool (writing processed/ as the
    reference does), hands the uint8 frames to the GPU in windows of ``window`` images while the pool
    already works on the next window, runs equal-sized frames of a window as batches of up to ``batch``,
    and writes each label PNG from the pool as soon as its labels are on the host (pinned ring,
    asynchronous copies).  ``streams`` batches are in flight at once, each on its own HIP stream and model
    object (``clone_shared``: one copy of the weights): a scan of 520-730 rows leaves the last round of tiles of
    many layers a quarter full, and the next image's kernels fill it (measured at batch 1 in f32: 154 -> 222
    images/s at 528 rows, 146 -> 168 at 720, 118 -> 119 at 1024; default 4).  Every shape runs on the library's default per-layer tiles (a cost model that
    lands within 0.1-0.5 % of the measured best in f32); ``autotune=True`` measures them once per distinct
    full-batch shape instead, which costs 0.5-0.9 s per shape and pays only for many thousands of images of one
    shape.  In "f16x2" mode ``calibrate`` (default) runs the first image once with every activation kept and leaves with
    ``NonFiniteLogits`` -- before any batch -- when a stored tensor lies outside the range the f16 pieces hold at f32 grade
    (``FCNResNet50.activation_peaks``: the silent counterpart of the non-finite word, which still rides back with every batch).
    Returns timing / count statistics of this rank."""
    import time
    import torch
    from collections import defaultdict, deque
    from concurrent.futures import ThreadPoolExecutor
    import threading
    from PIL import Image
    from .model import FCNResNet50
    from .pngio import write_png
    t_start = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if device_index is None else device_index
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if batch is None:
        batch = 8 if precision == "bf16" else 2
    n_streams = 4 if streams is None else max(1, int(streams))

    model = FCNResNet50(precision)
    model.to(dev)
    pre_model = FCNResNet50(precision).to(dev)      # its own context: the pool's device resizes never touch the predictor's
    marker = AbandonMarker(root)
    if rank == 0:
        generate_folders(root)
        marker.clear()
    if dist is not None:
        dist.barrier()
    if rank == 0:                                    # only one rank touches the checkpoint
        model.load_state_dict(torch.load(model_path, map_location="cpu", weights_only=True))
    if dist is not None:
        model.broadcast_weights(src=0)
    if precision == "f16x2" and model.pack_flags:
        # what the packer had to give up rides in the blob's trailer: every rank reads the same bits and leaves here alike
        err = NonFiniteLogits("the packed weights carry NBC_PACK flags %d (a weight row beyond the reach of the f16x2 row "
                              "normalisation, or a BatchNorm scale outside f32's normal range under its powers of two): f16x2 "
                              "would not be f32 grade on this checkpoint; rerun with --precision fp32" % model.pack_flags)
        err.batches_run, err.images_this_rank = 0, 0
        raise err
    models = [model] + [model.clone_shared() for _ in range(n_streams - 1)]
    # side streams only: the default stream stays with the pool's device resizes (pre_model)
    gpu_streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    for m in models:                                 # the largest workspace once: a context's buffers only grow, and a folder of
        m.reserve(batch, target_size, target_size)   # rising heights would otherwise free and reallocate them shape after shape
        if small_zones:                              # likewise the remove_small_zones workspace (9 bytes per pixel)
            m.remove_small_zones(torch.zeros((batch, target_size, target_size), dtype=torch.uint8, device=dev))
    torch.cuda.synchronize(dev)                      # weights uploaded / received before any side stream reads them
    t_ready = time.perf_counter()

    items = plan_items(root)
    n_total = len(items)
    workers = _host_workers()
    pool = ThreadPoolExecutor(max_workers=workers)

    def header_pixels(d):
        with open(d["src"] or d["processed"], "rb") as f:
            w, h = Image.open(f).size                # header only: nothing is decoded
        return w * h
    pixels = list(pool.map(header_pixels, items))
    shards = shard_by_pixels(pixels, world)
    mine = shards[rank]
    rows = np.zeros((len(mine), ROW_WIDTH), dtype=np.int64)
    resize_lock = threading.Lock()
    lvl_proc, lvl_lab = _png_level("processed"), _png_level("labels")

    prof = defaultdict(float)                        # seconds per stage, summed over threads (NBC_FOLDER_PROFILE=1 prints them)
    clock = time.perf_counter

    def prepare(gi):
        """Pool: decode + preprocess + write processed/ -> the uint8 frame the model sees."""
        d = items[gi]
        t0 = clock()
        if d["src"] is None:
            return _decode_rgb(d["processed"])
        out = preprocess_bmp_scan_on_device(d["src"], target_size, pre_model, resize_lock)     # raw scans: no host decode
        t1 = t2 = clock()
        if out is None:
            img = _decode_rgb(d["src"])
            t1 = clock()
            if max(img.shape[:2]) > target_size:
                with resize_lock:
                    out = preprocess_image(img, target_size, pre_model)
            else:
                out = preprocess_image(img, target_size)
            t2 = clock()
        write_png(d["processed"], out, lvl_proc)
        t3 = clock()
        prof["pool.decode"] += t1 - t0; prof["pool.preprocess"] += t2 - t1; prof["pool.write_processed"] += t3 - t2
        return out

    label_paths = []                                 # label PNGs this rank has written (removed again if the run turns out invalid)

    def finish(k, gi, lab, c1, c2):
        """Pool: label PNG (models.py:349-356) + the image's row."""
        d = items[gi]
        t0 = clock()
        path = os.path.join(root, "results", "outputs", d["wood"], d["name"])
        label_paths.append(path)
        write_png(path, label_png(lab), lvl_lab)
        rows[k] = (gi, lab.shape[0], lab.shape[1], c1, c2)
        prof["pool.write_labels"] += clock() - t0

    # pinned rings, one slot more than batches in flight: frames going up, labels + counts coming back
    # (allocated once for the largest batch: pinning memory costs milliseconds, and a folder of rising heights would
    # otherwise re-pin at every new shape)
    depth = n_streams + 1
    full = batch * target_size * target_size
    ring = [(torch.empty(full, dtype=torch.uint8).pin_memory(), torch.empty((batch, 3), dtype=torch.int64).pin_memory())
            for _ in range(depth)]
    ring_ev = [torch.cuda.Event() for _ in range(depth)]
    # f16x2: the context's sticky non-finite word rides back with every batch (nbc_nonfinite_peek_async: no synchronisation)
    flag_host = torch.zeros(depth, dtype=torch.int32).pin_memory()
    check_flag = precision == "f16x2"
    bad_seen = [False]
    stage = [{"buf": torch.empty(full * 3, dtype=torch.uint8).pin_memory(), "ev": torch.cuda.Event()} for _ in range(depth)]
    pending = deque()                                # (slot, [(k, gi)], n, h, w), oldest first
    done = []
    tuned = set()
    shape_count = defaultdict(int)
    n_batches = 0

    def consume(p):
        slot, members, n, h, w = p
        ring_ev[slot].synchronize()
        if check_flag and int(flag_host[slot]) != 0:
            bad_seen[0] = True                       # this batch's labels (and every later one's) are not valid: nothing is written
            if world > 1:
                marker.set()                         # the other ranks of the node stop at their next window
            return
        lab_host, cnt_host = ring[slot]
        labs = lab_host[: n * h * w].numpy().reshape(n, h, w).copy()
        cnts = cnt_host[:n].numpy().copy()
        for j, (k, gi) in enumerate(members):
            done.append(pool.submit(finish, k, gi, labs[j], int(cnts[j, 1]), int(cnts[j, 2])))

    # f16x2, calibration guard: nbc_pack_weights places every tensor by its BatchNorm's promise (|beta| + 3 |gamma|); whether the
    # DATA keeps that promise shows on the first image: rank 0 runs it once with every activation kept and looks at each
    # stored tensor's largest value.  One below 2^-8 sits mostly under the f16 pieces' 2^-12 floor -- finite logits, nothing
    # for the non-finite flag to see --, one beyond 2^14 is a factor four from f16's range: either way the folder belongs on
    # the f32 MFMA, and every rank leaves here alike (one scalar broadcast), before any batch has run.
    if check_flag and calibrate:
        verdict = torch.zeros(1, dtype=torch.int32)
        offenders = {}
        if rank == 0 and len(mine) > 0:
            first = prepare(mine[0])                     # (decoded once more inside the loop: one image's host work)
            peaks = models[0].activation_peaks(torch.from_numpy(np.ascontiguousarray(first[None])).to(dev))
            ok, offenders = FCNResNet50.f16x2_range_ok(peaks)
            verdict[0] = 0 if ok else 1
        if dist is not None:
            vd = verdict.to(dev) if dist.get_backend() == "nccl" else verdict
            dist.broadcast(vd, src=0)
            verdict = vd.cpu()
        if int(verdict[0]) != 0:
            pool.shutdown(wait=True, cancel_futures=True)
            worst = ", ".join("%s %.3g" % kv for kv in sorted(offenders.items(), key=lambda kv: kv[1])[:4])
            err = NonFiniteLogits("calibration on the first image: an activation tensor lies outside the range the f16 pieces hold "
                                  "at f32 grade (stored peak below 2^-8 or beyond 2^14%s): f16x2 would lose bits silently on this "
                                  "checkpoint; rerun with --precision fp32" % ((": " + worst) if worst else ""))
            err.batches_run, err.images_this_rank = 0, len(mine)
            raise err

    # the GPU loop runs in this thread next to up to 32 busy pool threads: a short switch interval keeps it
    # from waiting 5 ms for the interpreter lock at every step (restored below)
    import sys
    switch = sys.getswitchinterval()
    sys.setswitchinterval(2e-4)
    try:
        windows = [list(range(a, min(a + window, len(mine)))) for a in range(0, len(mine), window)]
        futs = {k: pool.submit(prepare, mine[k]) for k in (windows[0] if windows else [])}
        t_loop = time.perf_counter()
        for wi, win in enumerate(windows):
            if check_flag and world > 1 and not bad_seen[0] and marker.is_set():
                bad_seen[0] = True                       # another rank of the node saw the word
            if bad_seen[0]:                              # f16x2 cannot carry these weights: the run is abandoned here
                break
            if wi + 1 < len(windows):                    # the pool starts on the next window before the GPU gets this one
                for k in windows[wi + 1]:
                    futs[k] = pool.submit(prepare, mine[k])
            t0 = clock()
            frames = {k: futs.pop(k).result() for k in win}
            prof["main.wait_for_frames"] += clock() - t0
            groups = defaultdict(list)
            for k in win:
                groups[frames[k].shape].append(k)
            for shape, ks in sorted(groups.items()):
                shape_count[shape] += len(ks)
                for a in range(0, len(ks), batch):
                    if bad_seen[0]:
                        break
                    part = ks[a:a + batch]
                    n, (h, w) = len(part), shape[:2]
                    t0 = clock()
                    slot = n_batches % depth                  # free: at most n_streams batches are pending, on other slots
                    sid = n_batches % n_streams
                    mdl = models[sid]
                    st = stage[slot]                          # frames are packed while the GPU runs the batches before
                    if st["buf"].numel() < n * h * w * 3:       # cannot happen after the preprocessor (h, w <= target_size)
                        st["buf"] = torch.empty(n * h * w * 3, dtype=torch.uint8).pin_memory()
                    st["ev"].synchronize()                    # the copy that last read this buffer has finished
                    xb = st["buf"][: n * h * w * 3].view(n, h, w, 3)
                    xnp = xb.numpy()
                    for j, k in enumerate(part):
                        xnp[j] = frames[k]
                    need = n * h * w
                    if ring[slot][0].numel() < need or ring[slot][1].shape[0] < n:
                        ring[slot] = (torch.empty(need, dtype=torch.uint8).pin_memory(), torch.empty((n, 3), dtype=torch.int64).pin_memory())
                    t1 = clock()
                    with torch.cuda.stream(gpu_streams[sid]):
                        x = xb.to(dev, non_blocking=True)     # uint8 NHWC; normalised on the device
                        st["ev"].record()
                        key = (n, h, w)
                        if autotune and (sid, key) not in tuned and n == batch and shape_count[shape] >= 2 * batch:
                            mdl.autotune(x)                  # once per distinct full-batch shape and model object
                            tuned.add((sid, key))
                        labels, counts = mdl.predict_labels(x, exclude_nodes=exclude_nodes, labels_dtype=torch.uint8,
                                                            small_zones=small_zones)   # models.py:269-276 on the device
                        ring[slot][0][:need].copy_(labels.reshape(-1), non_blocking=True)
                        ring[slot][1][:n].copy_(counts, non_blocking=True)
                        if check_flag:
                            mdl.nonfinite_peek_async(flag_host[slot:slot + 1])
                        ring_ev[slot].record()
                    t2 = clock()
                    pending.append((slot, [(k, mine[k]) for k in part], n, h, w))
                    while len(pending) > n_streams:           # the oldest batch's labels, while the newer ones run
                        consume(pending.popleft())
                    prof["main.pack"] += t1 - t0; prof["main.h2d_and_launch"] += t2 - t1; prof["main.consume"] += clock() - t2
                    n_batches += 1
            frames.clear()
        while pending:
            consume(pending.popleft())
        for f in done:
            f.result()
    finally:                                         # also on an exception from a worker: no stray threads, switch interval restored
        pool.shutdown(wait=True, cancel_futures=True)
        sys.setswitchinterval(switch)
    torch.cuda.synchronize()
    t_done = time.perf_counter()
    if precision == "f16x2":
        # f16x2 keeps every value as two f16 pieces: an activation beyond +-65504 cannot be represented and turns into NaN
        # (never into a silently wrong number).  Unknown weights that do this belong in the f32 MFMA mode.  Every rank
        # learns of it (one more tiny collective) so that all of them leave before the row gather, none waits in it.
        # The word rides back with every batch (consume), so a rank that sees it stops at that batch instead of finishing its
        # shard; the contexts are all read (and reset) here once more, whatever the first one says.
        bad = any([m.nonfinite_seen() for m in models]) or bad_seen[0]
        if dist is not None:
            flag = torch.tensor([int(bad)], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            bad = bool(int(flag.item()))
            if rank == 0:
                marker.clear()                       # every rank is past its loop (the all-reduce above)
        if bad:
            for path in label_paths:                 # this rank's label PNGs of the invalid run: a crash before the rerun
                try:                                 # must not leave them behind (the processed/ images do not depend on
                    os.remove(path)                  # the arithmetic and stay)
                except OSError:
                    pass
            err = NonFiniteLogits("a forward produced non-finite logits in f16x2 mode (an activation beyond f16's range, or NaN/inf "
                                  "in the weights): the run was abandoned after %d of this rank's %d images and its label PNGs "
                                  "removed; rerun with --precision fp32" % (min(n_batches * batch, len(mine)), len(mine)))
            err.batches_run, err.images_this_rank = n_batches, len(mine)
            raise err

    if os.environ.get("NBC_FOLDER_PROFILE"):
        print("rank %d stage seconds (pool stages summed over %d threads): %s; loop wall %.2f s" %
              (rank, workers, ", ".join("%s %.2f" % kv for kv in sorted(prof.items())), t_done - t_loop), flush=True)
    cap = max(len(s) for s in shards) if shards else 0
    # RCCL gathers device tensors; a gloo group (one-GPU rehearsals of the multi-rank path) gathers on the host
    gather_dev = dev if dist is not None and dist.get_backend() == "nccl" else None
    allrows = gather_rows(rows, n_total, world, dist, gather_dev, cap=cap)
    if rank == 0:
        write_stats_csv(os.path.join(root, "results", "final_stats.csv"),
                        [stats_row(items[int(r[0])]["name"], items[int(r[0])]["wood"], int(r[1]), int(r[2]), int(r[3]), int(r[4]))
                         for r in allrows])
    if dist is not None:
        dist.barrier()
    t_end = time.perf_counter()
    return {"rank": rank, "world": world, "images_total": n_total, "images_this_rank": len(mine), "batches": n_batches,
            "batch": batch, "host_workers": workers, "setup_s": t_ready - t_start, "loop_s": t_done - t_loop,
            "total_s": t_end - t_start, "images_per_s_loop": len(mine) / max(t_done - t_loop, 1e-9),
            "distinct_shapes": len(shape_count), "autotuned_shapes": len({k for _, k in tuned}), "streams": n_streams}


def launch_ranks(n: int, argv: Sequence[str]) -> int:
    """``--gpus N`` without a torchrun environment: start N ranks (one per GPU) as a child process."""
    import socket
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < n:                # counts devices without initialising HIP
        print("predict: --gpus %d but this node shows %d GPU(s)" % (n, torch.cuda.device_count()), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "neuralbarkcalculator_amd.predict"] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    import sys
    ap = argparse.ArgumentParser(description="MI355X folder prediction (mirrors bark_calculator/predict.py)")
    ap.add_argument("root_path", metavar="DIR")
    ap.add_argument("--device", default="cuda:0", help="cuda:N (the CPU path is the reference itself)")
    ap.add_argument("--exclude_nodes", action="store_true")
    ap.add_argument("--only_preprocess", action="store_true")
    ap.add_argument("--model_path", default="./best_model.pt")       # predict.py:57
    ap.add_argument("--precision", choices=["auto", "fp32", "f16x2", "bf16"], default="auto",
                    help="auto (default): the f32-grade f16x2 mode, 2.4x faster than the f32 MFMA at the same tolerances, and a "
                         "second run in fp32 if the weights drive an activation beyond f16's range (the library says so); fp32: "
                         "f32 MFMA; bf16: throughput mode, not f32 grade")
    ap.add_argument("--no_small_zones", action="store_true")
    ap.add_argument("--gpus", type=int, default=1, help="shard the folder over N GPUs of this node (one process each, RCCL)")
    ap.add_argument("--batch", type=int, default=None, help="frames of equal size per forward (default 2 in fp32, 8 in bf16)")
    ap.add_argument("--streams", type=int, default=None, help="batches in flight, each on its own HIP stream (default 4)")
    ap.add_argument("--autotune", action="store_true",
                    help="measure the conv tile shapes once per distinct full-batch image shape (0.5-0.9 s each) instead of the default choice")
    raw = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(raw)
    if not args.device.startswith("cuda"):
        raise SystemExit("this package is the MI355X path; run the reference for --device=cpu")
    if args.only_preprocess:                             # predict.py:53-55: the resize runs on the device here too
        from .model import FCNResNet50
        generate_folders(args.root_path, True)
        preprocess_images(args.root_path, model=FCNResNet50("fp32").to(args.device))   # a context for the resize kernel: no weights, no arithmetic mode involved
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, raw))
    idx = None
    if "WORLD_SIZE" not in os.environ and ":" in args.device:
        idx = int(args.device.split(":")[1])
    kw = dict(batch=args.batch, autotune=args.autotune, streams=args.streams)
    if args.precision == "auto":
        stats = None
        try:
            stats = predict_folder(args.root_path, args.model_path, "f16x2", args.exclude_nodes, not args.no_small_zones, idx, **kw)
        except NonFiniteLogits as e:                 # raised on every rank alike
            if int(os.environ.get("RANK", "0")) == 0:
                print("predict: %s -- running the folder again on the f32 MFMA" % e, flush=True)
        if stats is None:
            # outside the except block: the exception's traceback holds the first run's frame (four model contexts with
            # their workspaces, the pinned rings, the streams) for as long as the block lasts
            import gc
            import torch
            gc.collect()
            torch.cuda.empty_cache()
            stats = predict_folder(args.root_path, args.model_path, "fp32", args.exclude_nodes, not args.no_small_zones, idx, **kw)
    else:
        stats = predict_folder(args.root_path, args.model_path, args.precision, args.exclude_nodes,
                               not args.no_small_zones, idx, **kw)
    if stats["rank"] == 0:
        print("predicted %(images_total)d images (%(images_this_rank)d on rank 0, %(batches)d batches): %(total_s).2f s, "
              "%(images_per_s_loop).1f images/s in the loop on this rank" % stats)


if __name__ == "__main__":
    main()
