"""Host logic around the hot path (CPU only): folder contract, CSV/PNG formats of the reference
(models.py:252-255,321-332,349-364; dataset.py:41-68; predict.py:10-48) and the world_size-2
sharding + gather over gloo."""
import csv
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neuralbarkcalculator_amd import predict as drv


def _touch_image(path, h=8, w=8, value=200):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(np.full((h, w, 3), value, np.uint8), mode="RGB").save(path)


def test_folder_contract_and_ordering(tmp_path):
    root = str(tmp_path)
    _touch_image(os.path.join(root, "samples", "sapin", "b.png"))
    _touch_image(os.path.join(root, "samples", "sapin", "a.bmp"))
    _touch_image(os.path.join(root, "samples", "epinette_gelee", "bmp_z.bmp"))
    _touch_image(os.path.join(root, "samples", "epinette_gelee", "notes.txt.png"))
    os.makedirs(os.path.join(root, "samples", "unknown_type"))
    open(os.path.join(root, "samples", "sapin", "readme.txt"), "w").close()
    items = drv.list_images(root)
    # wood types in the fixed order of dataset.py:50, names sorted, "bmp"->"png" everywhere (dataset.py:58)
    assert [(n, w) for _, n, w in items] == [("png_z.png", "epinette_gelee"), ("notes.txt.png", "epinette_gelee"),
                                             ("a.png", "sapin"), ("b.png", "sapin")]
    drv.generate_folders(root)
    for sub in ("processed/samples/sapin", "results/outputs/epinette_gelee", "results/combined_images/sapin"):
        assert os.path.isdir(os.path.join(root, sub))
    assert not os.path.isdir(os.path.join(root, "processed/samples/epinette_non_gelee"))   # absent type
    drv.preprocess_images(root)
    assert sorted(os.listdir(os.path.join(root, "processed/samples/sapin"))) == ["a.png", "b.png"]
    with pytest.raises(IOError):
        drv.list_images(os.path.join(root, "nowhere"))


def test_trim_black():
    img = np.full((16, 16, 3), 0.5, np.float32)
    img[:3] = 0.0                       # three black rows on top
    img[-2:, :3] = 0.0                  # bottom rows: 3/16 = 18.75 % dark pixels -> trimmed too
    img[5, :2] = 0.0                    # 12.5 % dark: kept
    out = drv.trim_black(img)
    assert out.shape == (11, 16, 3)


@pytest.mark.parametrize("path", sorted(__import__("glob").glob(os.path.join(os.path.dirname(__file__), "golden", "preprocess_*.npz"))))
def test_preprocessor_matches_skimage_fixture(path):
    """models.py:191-203 (resize order=3 reflect, trim_black, imsave) against scikit-image 0.18.3."""
    g = np.load(path, allow_pickle=False)
    out = drv.preprocess_image(g["image"], int(g["target"]))
    assert out.shape == g["expected"].shape and out.dtype == np.uint8
    assert np.array_equal(out, g["expected"]), f"{int((out != g['expected']).sum())} bytes differ from scikit-image's file"
    # and in front of the float -> uint8 conversion: the float32 image itself, value for value
    img = g["image"].astype(np.float32) / np.float32(255)
    t = int(g["target"])
    if max(img.shape[:2]) > t:
        img = drv.resize_bicubic_reflect(img, t, t)
    if img.shape[0] == img.shape[1]:
        img = drv.trim_black(img)
    assert img.dtype == np.float32 and np.array_equal(img, g["float32"])


def test_preprocess_images_resizes_oversize_inputs(tmp_path):
    root = str(tmp_path)
    _touch_image(os.path.join(root, "samples", "sapin", "big.png"), 40, 1030)
    drv.generate_folders(root, only_preprocess=True)
    drv.preprocess_images(root, target_size=64)
    from PIL import Image
    assert Image.open(os.path.join(root, "processed", "samples", "sapin", "big.png")).size == (64, 64)


def test_csv_rows_match_reference_arithmetic(tmp_path):
    # the reference's own expressions (models.py:321-332) evaluated with torch on a label map
    g = torch.Generator().manual_seed(3)
    labels = torch.randint(0, 3, (1, 520, 1024), generator=g)
    want = ["x.png", "sapin"]
    for c in (1, 2):
        n_pixels = (labels == c).float().cpu()
        class_percent = n_pixels.mean()
        want.append("{:.5f}".format(class_percent * 100))
        want.append("{:.5f}".format((n_pixels.sum() * (3.6 * 3.6)).item()))
    got = drv.stats_row("x.png", "sapin", 520, 1024, int((labels == 1).sum()), int((labels == 2).sum()))
    assert got == want
    path = os.path.join(str(tmp_path), "final_stats.csv")
    drv.write_stats_csv(path, [got])
    rows = list(csv.reader(open(path), delimiter="\t"))
    assert rows[0] == ["Name", "Type", "Image Size", "Output Bark %", "Bark area (mm^2)", "Output Node %",
                       "Node area (mm^2)"]
    assert rows[1] == want and len(rows[1]) == 6          # 7 header names, 6 values: as in the reference
    png = drv.label_png(labels[0].numpy())
    assert png.dtype == np.uint8 and set(np.unique(png)) == {0, 127, 255}
    assert ((png == 127) == (labels[0].numpy() == 1)).all()


def test_shard_indices_partition():
    for n in (0, 1, 7, 8, 1000):
        for world in (1, 2, 8):
            parts = [drv.shard_indices(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _gather_worker(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = drv.shard_indices(n_total, rank, world)
        rows = np.array([[i, 100 + i, 1024, 3 * i, 5 * i + 1] for i in mine], dtype=np.int64).reshape(-1, drv.ROW_WIDTH)
        allrows = drv.gather_rows(rows, n_total, world, dist)
        # the one-off weight broadcast: rank 0's packed blob reaches every rank unchanged
        blob = torch.arange(1000, dtype=torch.int64).to(torch.uint8) if rank == 0 else torch.zeros(1000, dtype=torch.uint8)
        dist.broadcast(blob, src=0)
        np.save(os.path.join(out_dir, f"rows{rank}.npy"), allrows)
        np.save(os.path.join(out_dir, f"blob{rank}.npy"), blob.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [0, 1, 7])
def test_world_size_2_shard_and_gather_over_gloo(tmp_path, n_total):
    world = 2
    port = 29600 + n_total
    mp.spawn(_gather_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    want = np.array([[i, 100 + i, 1024, 3 * i, 5 * i + 1] for i in range(n_total)], dtype=np.int64).reshape(-1, 5)
    for r in range(world):
        np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), f"rows{r}.npy")), want)
        np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), f"blob{r}.npy")),
                                      (np.arange(1000) % 256).astype(np.uint8))


def test_shard_by_pixels_is_contiguous_and_balanced():
    rng = np.random.default_rng(3)
    for n, world in ((0, 2), (1, 4), (5, 8), (1000, 8), (333, 3)):
        px = [int(1024 * h) for h in rng.integers(520, 731, size=n)]        # height-trimmed frames (res/*.png: H 520..730)
        shards = drv.shard_by_pixels(px, world)
        assert len(shards) == world and sum(shards, []) == list(range(n))   # a partition into contiguous ranges, in order
        if n >= 100:
            loads = [sum(px[i] for i in s) for s in shards]
            assert max(loads) - min(loads) <= 2 * max(px)                    # within two images of each other
    # uneven sizes: one huge scan does not drag a rank's share of small ones along
    shards = drv.shard_by_pixels([100, 100, 100, 100, 10000, 100, 100, 100, 100], 2)
    assert shards == [[0, 1, 2, 3], [4, 5, 6, 7, 8]] or shards == [[0, 1, 2, 3, 4], [5, 6, 7, 8]]


def _pixel_gather_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        px = [1, 1, 1, 1, 1, 1, 9, 9]                                        # shards of unequal length: 6 + 2 images
        shards = drv.shard_by_pixels(px, world)
        rows = np.array([[i, px[i], 1024, i, 2 * i] for i in shards[rank]], dtype=np.int64).reshape(-1, drv.ROW_WIDTH)
        allrows = drv.gather_rows(rows, len(px), world, dist, cap=max(len(s) for s in shards))
        np.save(os.path.join(out_dir, f"prow{rank}.npy"), allrows)
    finally:
        dist.destroy_process_group()


def test_gather_of_unequal_pixel_balanced_shards_over_gloo(tmp_path):
    mp.spawn(_pixel_gather_worker, args=(2, 29650, str(tmp_path)), nprocs=2, join=True)
    px = [1, 1, 1, 1, 1, 1, 9, 9]
    want = np.array([[i, px[i], 1024, i, 2 * i] for i in range(8)], dtype=np.int64)
    for r in range(2):
        np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), f"prow{r}.npy")), want)


def test_png_writer_round_trips_through_pil():
    import io
    from PIL import Image
    from neuralbarkcalculator_amd.pngio import encode_png
    rng = np.random.default_rng(1)
    for shape in ((1, 1), (7, 5), (64, 33), (7, 5, 3), (130, 257, 3)):
        a = rng.integers(0, 256, size=shape, dtype=np.uint8)
        for level in (0, 1, 6):
            im = Image.open(io.BytesIO(encode_png(a, level)))
            assert im.mode == ("L" if a.ndim == 2 else "RGB")                 # models.py:355: mode 'L' label maps
            assert np.array_equal(np.asarray(im), a)
    lab = drv.label_png(rng.integers(0, 3, size=(40, 50)).astype(np.uint8))
    assert set(np.unique(np.asarray(Image.open(io.BytesIO(encode_png(lab)))))) <= {0, 127, 255}
    with pytest.raises(ValueError):
        encode_png(np.zeros((4, 4, 4), np.uint8))
    with pytest.raises(ValueError):
        encode_png(np.zeros((4, 4), np.float32))


def test_plan_items_follows_the_reference_listing(tmp_path):
    """What gets predicted = what processed/ holds after the preprocessor ran, in dataset.py:41-68 order."""
    root = str(tmp_path)
    _touch_image(os.path.join(root, "samples", "sapin", "x.bmp"), value=10)
    _touch_image(os.path.join(root, "samples", "sapin", "x.png"), value=20)      # same output name: the later one wins
    _touch_image(os.path.join(root, "samples", "sapin", "a.bmp"))
    _touch_image(os.path.join(root, "samples", "epinette_gelee", "z.png"))
    _touch_image(os.path.join(root, "processed", "samples", "sapin", "old.png"))  # left over from an earlier run
    items = drv.plan_items(root)
    assert [(d["wood"], d["name"]) for d in items] == [("epinette_gelee", "z.png"), ("sapin", "a.png"), ("sapin", "old.png"),
                                                       ("sapin", "x.png")]
    byname = {d["name"]: d for d in items}
    assert byname["x.png"]["src"].endswith("x.png") and byname["old.png"]["src"] is None
    assert byname["a.png"]["processed"] == os.path.join(root, "processed", "samples", "sapin", "a.png")


def test_uint8_round_trip_through_totensor_and_imsave_is_the_identity():
    """ToTensor (u8 / 255 in float32) then imsave's float -> uint8 gives every byte back, and trim_black's lit test
    on the float image equals "some byte is non-zero": the no-resize route of preprocess_image works on bytes."""
    assert drv._U8_ROUND_TRIP_IS_IDENTITY
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)
    img[:5] = 0
    img[-9:, ::8] = 0                       # 1/8 of the pixels black: below the 15 % bar, kept
    img[-4:, ::2] = 0                       # half of them black: trimmed
    img[20, :, 1:] = 0                      # one channel left: still lit
    f = img.astype(np.float32) / np.float32(255)
    want = drv._float_to_u8(drv.trim_black(f))
    assert want.shape[0] == 64 - 5 - 4
    assert np.array_equal(drv.preprocess_image(img, 64), want)
    assert np.array_equal(np.sum(f, axis=-1) > 1e-3, img.any(axis=-1))


def test_fast_bmp_decoder_equals_pil(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(2)
    for h, w in ((5, 7), (16, 16), (33, 10), (64, 1)):                 # widths whose rows need 0-3 padding bytes
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        p = str(tmp_path / f"x{h}_{w}.bmp")
        Image.fromarray(img, mode="RGB").save(p)
        assert drv._decode_bmp24(open(p, "rb").read()) is not None
        assert np.array_equal(drv._decode_rgb(p), img)
    # anything else goes through PIL: palette BMP, grey PNG (converted to RGB like pil_loader does)
    pal = str(tmp_path / "pal.bmp")
    Image.fromarray(rng.integers(0, 256, size=(9, 9), dtype=np.uint8), mode="L").save(pal)
    assert drv._decode_bmp24(open(pal, "rb").read()) is None
    assert np.array_equal(drv._decode_rgb(pal), np.asarray(Image.open(pal).convert("RGB")))
    assert drv._decode_bmp24(b"BM" + b"\0" * 10) is None


def test_gpus_flag_refuses_more_ranks_than_gpus(tmp_path):
    """`--gpus N` starts its own ranks only when the node shows N GPUs; asked for more it says so and exits non-zero
    (no silent one-rank run).  Runs anywhere: counting devices does not initialise HIP."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 3
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "--gpus %d" % n in p.stderr and not p.stdout.strip()
    _touch_image(os.path.join(str(tmp_path), "samples", "sapin", "a.png"))
    p = subprocess.run([sys.executable, "-m", "neuralbarkcalculator_amd.predict", str(tmp_path), "--gpus", str(n)], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "--gpus %d" % n in p.stderr


def test_bmp_header_fuzz_never_crashes(tmp_path):
    """The fast BMP path sees whatever a folder holds: truncated files, absurd sizes, other flavours.  Every mutation
    of a good header either decodes to an array of the declared size or is handed to PIL (None), never an exception
    or an out-of-bounds read."""
    import io
    import struct
    from PIL import Image
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=(13, 21, 3), dtype=np.uint8)
    good = io.BytesIO()
    Image.fromarray(img, mode="RGB").save(good, format="BMP")
    good = good.getvalue()
    assert np.array_equal(drv._decode_bmp24(good), img)
    for trial in range(400):
        b = bytearray(good)
        kind = trial % 5
        if kind == 0:                                   # random header bytes
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, 54))] = int(rng.integers(0, 256))
        elif kind == 1:                                 # truncated file
            b = b[: int(rng.integers(0, len(b)))]
        elif kind == 2:                                 # absurd width / height
            struct.pack_into("<ii", b, 18, int(rng.integers(-2**31, 2**31 - 1)), int(rng.integers(-2**31, 2**31 - 1)))
        elif kind == 3:                                 # pixel offset anywhere
            struct.pack_into("<I", b, 10, int(rng.integers(0, 2**32 - 1)))
        else:                                           # other bit depths / compression
            struct.pack_into("<HHI", b, 26, int(rng.integers(0, 4)), int(rng.choice([1, 4, 8, 16, 24, 32])), int(rng.integers(0, 4)))
        out = drv._decode_bmp24(bytes(b))
        lay = drv._bmp24_layout(bytes(b[:54]), len(b))
        assert (out is None) == (lay is None)
        if out is not None:
            off, w, rows, stride, _ = lay
            assert out.shape == (rows, w, 3) and off + stride * rows <= len(b)


def test_perf_marker_gate():
    """tests/conftest.py: timing tests run only when the -m expression asks for the perf marker itself."""
    from conftest import marker_expression_asks_for_perf as asks
    assert asks("perf") and asks("gpu and perf") and asks("perf and gpu")
    for expr in ("", "gpu", "not gpu", "gpu and not perf", "not perf", "perfect", "gpu or perfect"):
        assert not asks(expr), expr


def test_abandon_marker_is_a_file_every_rank_of_the_node_can_see(tmp_path):
    """the early abandon of an f16x2 folder run was rank-local.  The rank that sees the non-finite word now drops
    a marker file under results/ that every rank polls once per window (no collective: ranks have different numbers of
    windows); rank 0 clears a stale one at the start and the final one after the flag all-reduce."""
    from neuralbarkcalculator_amd.predict import AbandonMarker
    a, b = AbandonMarker(str(tmp_path)), AbandonMarker(str(tmp_path))     # two ranks, one folder
    assert not a.is_set() and not b.is_set()
    a.clear()                                                             # nothing to clear: no error
    b.set()
    assert a.is_set() and os.path.dirname(a.path) == str(tmp_path / "results")
    a.set()                                                               # twice is fine
    a.clear()
    assert not b.is_set()
