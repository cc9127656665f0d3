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

Multi-GPU (one process per GPU, ``torch.distributed`` over RCCL): rank r takes images
``r, r+W, ...`` of the sorted list; rank 0 alone reads the checkpoint and broadcasts the packed
weights; every rank fills int64 rows ``(global_idx, H, W, count_1, count_2)`` which one
``all_gather`` brings to rank 0 for the CSV.  No collective sits in the per-image path.
"""
from __future__ import annotations

import argparse
import csv
import os
from typing import List, Sequence, Tuple

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


def _cubic(x, f0, f1, f2, f3, dt=np.float64):
    """scikit-image's cubic_interpolation (Catmull-Rom, a = -0.5): values at -1, 0, 1, 2; x in [0, 1].
    C evaluation rules of its Cython source for an image of float type ``dt``: the two differences of
    image values are taken in ``dt``, everything that meets a (double) literal in double."""
    d20 = (f2.astype(dt) - f0.astype(dt)).astype(np.float64)
    d12 = (f1.astype(dt) - f2.astype(dt)).astype(np.float64)
    f0, f1, f2, f3 = (f.astype(np.float64) for f in (f0, f1, f2, f3))
    return f1 + 0.5 * x * (d20 + x * (2.0 * f0 - 5.0 * f1 + 4.0 * f2 - f3 + x * (3.0 * d12 + f3 - f0)))


def _reflect(i: np.ndarray, n: int) -> np.ndarray:
    """numpy.pad 'reflect' indexing (mirror without repeating the edge)."""
    if n == 1:
        return np.zeros_like(i)
    period = 2 * (n - 1)
    i = np.mod(i, period)
    return np.where(i >= n, period - i, i)


def resize_bicubic_reflect(image: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """``skimage.transform.resize(image, (out_h, out_w), order=3, mode='reflect',
    anti_aliasing=False)`` (models.py:194-198) for a float HWC image: output pixel i samples the
    input at ``factor * (i + 0.5) - 0.5`` with a separable 4-tap Catmull-Rom kernel in the image's
    own float type, reflected borders, result clipped to the input range (``clip=True``).  Pinned by
    scikit-image 0.18.3 fixtures (tests/golden/preprocess_*.npz)."""
    h, w = image.shape[:2]
    dt = image.dtype if image.dtype in (np.float32, np.float64) else np.float64
    img = image.astype(dt, copy=False)
    # scikit-image's _warp_fast works in the image's float type (float32 after ToTensor): sample
    # coordinates and the two fractional offsets are rounded to it; each cubic_interpolation call
    # evaluates in double (its literals are doubles) and returns the image's type.
    ry = ((h / out_h) * (np.arange(out_h) + 0.5) - 0.5).astype(dt)
    rx = ((w / out_w) * (np.arange(out_w) + 0.5) - 0.5).astype(dt)
    y0 = np.floor(ry).astype(np.int64)
    x0 = np.floor(rx).astype(np.int64)
    ty = (ry - y0.astype(dt)).astype(np.float64).reshape(-1, 1, 1)
    tx = (rx - x0.astype(dt)).astype(np.float64).reshape(1, -1, 1)
    cols = [_reflect(x0 + k - 1, w) for k in range(4)]
    fr = []
    for k in range(4):
        rows = img[_reflect(y0 + k - 1, h)]
        fr.append(_cubic(tx, *[rows[:, c] for c in cols], dt=dt).astype(dt))
    out = _cubic(ty, *fr, dt=dt).astype(dt)
    return np.clip(out, img.min(), img.max())


def preprocess_image(img_u8: np.ndarray, target_size: int = 1024, model=None) -> np.ndarray:
    """models.py:191-203 for one decoded RGB image: ToTensor (u8 -> float32 / 255), resize to
    ``target_size`` x ``target_size`` when either side is larger, ``trim_black`` when square,
    float -> uint8 like ``skimage.io.imsave`` does through imageio (``uint8(float64(x) * 255 + 0.499999999)``:
    the "Lossy conversion from float32 to uint8" path, which rounds exact halves down).  With ``model``
    (an ``FCNResNet50`` on a device) the resize runs there."""
    if max(img_u8.shape[:2]) > target_size and model is not None:
        # the resize on the device (nbc_resize_cubic_u8, bit-identical to the numpy form below, which
        # takes ~1.1 s for a 4096^2 image)
        import torch
        dev_img = torch.from_numpy(np.ascontiguousarray(img_u8)).to(model.device)
        image = model.resize_cubic_u8(dev_img, target_size, target_size).cpu().numpy()
    else:
        image = img_u8.astype(np.float32) / np.float32(255)
        if max(image.shape[:2]) > target_size:
            image = resize_bicubic_reflect(image, target_size, target_size)
    if image.shape[0] == image.shape[1]:
        image = trim_black(image)
    return np.clip(image.astype(np.float64) * 255.0 + 0.499999999, 0, 255).astype(np.uint8)


def preprocess_images(root: str, target_size: int = 1024, model=None) -> None:
    """models.py:173-203: decode, resize / trim, save as PNG under processed/.  Decoding and PNG
    encoding (the bulk: ~0.2 s per 1024x1024 image) run on a small thread pool; a device resize, when
    ``model`` is given, is serialised (one context, not thread-safe)."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    lock = threading.Lock()

    def one(item):
        path, name, wood = item
        with open(path, "rb") as f:
            img = np.asarray(Image.open(f).convert("RGB"))                 # dataset.py:82-90
        if model is not None and max(img.shape[:2]) > target_size:
            with lock:
                out = preprocess_image(img, target_size, model)
        else:
            out = preprocess_image(img, target_size)
        Image.fromarray(out, mode="RGB").save(os.path.join(root, "processed", "samples", wood, name))

    workers = max(1, min(32, int(os.environ.get("NBC_HOST_WORKERS", "8"))))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        list(pool.map(one, list_images(root)))


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """Images of rank ``rank``: r, r+W, r+2W, ...."""
    return list(range(rank, n, world))


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


def gather_rows(local_rows: np.ndarray, n_total: int, world: int, dist=None, device=None) -> np.ndarray:
    """all_gather of fixed-size per-rank row buffers; returns the rows sorted by global index.
    ``local_rows``: int64 [k, ROW_WIDTH] with k <= ceil(n_total / world)."""
    import torch
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


def predict_folder(root: str, model_path: str = "./best_model.pt", precision: str = "fp32",
                   exclude_nodes: bool = False, small_zones: bool = True, device_index: int = None) -> None:
    """predict.py:51-58 + models.py:230-364 with the model call on the MI355X path."""
    import torch
    from PIL import Image
    from .model import FCNResNet50
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

    model = FCNResNet50(precision)
    model.to(dev)
    if rank == 0:
        generate_folders(root)
        preprocess_images(root, model=model)         # the resize of oversize images runs on the device
    if dist is not None:
        dist.barrier()
    if rank == 0:                                    # only one rank touches the checkpoint
        model.load_state_dict(torch.load(model_path, map_location="cpu", weights_only=True))
        model.to(dev)
    if dist is not None:
        model.broadcast_weights(src=0)

    images = list_images(os.path.join(root, "processed"))
    mine = shard_indices(len(images), rank, world)
    rows = np.zeros((len(mine), ROW_WIDTH), dtype=np.int64)

    # The forward is ~1.4 ms per image and remove_small_zones ~0.14 ms on the device; decoding the PNG
    # and writing the label PNG are tens of milliseconds of host work each.  They run on a small
    # thread pool around the GPU loop (PIL and numpy release the GIL in their C code): a few images are
    # decoded ahead and every image's PNG is handed off as soon as its labels are on the host.
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor

    def load(gi):
        with open(images[gi][0], "rb") as f:
            return np.array(Image.open(f).convert("RGB"))           # own, writable, contiguous copy

    def finish(gi, lab, counts):
        _, name, wood = images[gi]
        Image.fromarray(label_png(lab), mode="L").save(os.path.join(root, "results", "outputs", wood, name))
        return (gi, lab.shape[0], lab.shape[1], counts[0], counts[1])

    workers = max(1, min(32, int(os.environ.get("NBC_HOST_WORKERS", "8"))))
    ahead = 2 * workers
    with ThreadPoolExecutor(max_workers=workers) as pool:
        loads = deque(pool.submit(load, gi) for gi in mine[:ahead])
        done = []
        for k, gi in enumerate(mine):
            img = loads.popleft().result()
            if k + ahead < len(mine):
                loads.append(pool.submit(load, mine[k + ahead]))
            x = torch.from_numpy(img)[None].to(dev)                  # uint8 NHWC; normalised on device
            labels, counts = model.predict_labels(x, exclude_nodes=exclude_nodes, labels_dtype=torch.uint8,
                                                  small_zones=small_zones)   # models.py:269-276 on the device
            lab = labels[0].cpu().numpy()
            cnt = (int(counts[0, 1]), int(counts[0, 2]))
            done.append(pool.submit(finish, gi, lab, cnt))
        for k, f in enumerate(done):
            rows[k] = f.result()

    allrows = gather_rows(rows, len(images), world, dist, dev)
    if rank == 0:
        write_stats_csv(os.path.join(root, "results", "final_stats.csv"),
                        [stats_row(images[int(r[0])][1], images[int(r[0])][2], int(r[1]), int(r[2]), int(r[3]), int(r[4]))
                         for r in allrows])
    if dist is not None:
        dist.barrier()


def main(argv=None):
    ap = argparse.ArgumentParser(description="MI355X folder prediction (mirrors bark_calculator/predict.py)")
    ap.add_argument("root_path", metavar="DIR")
    ap.add_argument("--device", default="cuda:0", help="cuda:N (the CPU path is the reference itself)")
    ap.add_argument("--exclude_nodes", action="store_true")
    ap.add_argument("--only_preprocess", action="store_true")
    ap.add_argument("--model_path", default="./best_model.pt")       # predict.py:57
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32")
    ap.add_argument("--no_small_zones", action="store_true")
    args = ap.parse_args(argv)
    if args.only_preprocess:
        generate_folders(args.root_path, True)
        preprocess_images(args.root_path)
        return
    if not args.device.startswith("cuda"):
        raise SystemExit("this package is the MI355X path; run the reference for --device=cpu")
    idx = None
    if "WORLD_SIZE" not in os.environ and ":" in args.device:
        idx = int(args.device.split(":")[1])
    predict_folder(args.root_path, args.model_path, args.precision, args.exclude_nodes,
                   not args.no_small_zones, idx)


if __name__ == "__main__":
    main()
