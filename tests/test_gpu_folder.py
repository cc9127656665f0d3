"""End-to-end folder prediction on the GPU (BASELINE.json configs[4]: weights loaded from a local
.pt through load_state_dict, --exclude_nodes remap, label PNG + CSV match against the CPU oracle
driven through the same post-processing)."""
import csv
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import overflowing_state_dict
from neuralbarkcalculator_amd import predict as drv
from neuralbarkcalculator_amd import synth
from neuralbarkcalculator_amd.postprocess import remove_small_zones

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["fp32", "f16x2"])
@pytest.mark.parametrize("exclude_nodes", [False, True])
def test_predict_folder_matches_oracle(tmp_path, oracle_model, sd_np, built_lib, exclude_nodes, precision):
    """configs[4]: trained-like weights loaded from a local .pt, --exclude_nodes, label PNGs and CSV rows against the
    oracle on eight held frames, in both f32-grade modes under the same allowance."""
    from oracle.fcn_resnet50_oracle import predict_labels
    root = str(tmp_path)
    layout = [("epinette_gelee", "a01.bmp", 30, 256, 256), ("epinette_gelee", "a02.png", 31, 200, 256),
              ("sapin", "s1.png", 32, 256, 256), ("sapin", "s2.bmp", 33, 136, 256),
              ("epinette_non_gelee", "n1.png", 34, 256, 256), ("sapin", "s0.png", 35, 256, 256),
              ("epinette_gelee", "a03.png", 36, 256, 256), ("epinette_non_gelee", "n0.png", 37, 264, 256)]
    frames = {}
    for wood, name, idx, h, w in layout:
        d = os.path.join(root, "samples", wood)
        os.makedirs(d, exist_ok=True)
        img = synth.make_frame(idx, h, w)
        Image.fromarray(img, mode="RGB").save(os.path.join(d, name))
        frames[(wood, name.replace("bmp", "png"))] = img
    ckpt = os.path.join(root, "best_model.pt")
    torch.save({k: torch.from_numpy(v) for k, v in sd_np.items()}, ckpt)     # predict.py:57 loads a local file

    drv.predict_folder(root, ckpt, precision=precision, exclude_nodes=exclude_nodes, device_index=0)

    rows = list(csv.reader(open(os.path.join(root, "results", "final_stats.csv")), delimiter="\t"))
    assert rows[0] == drv.CSV_HEADER and len(rows) == 1 + len(layout)
    order = [(n, w) for _, n, w in drv.list_images(os.path.join(root, "processed"))]
    assert [(r[0], r[1]) for r in rows[1:]] == order
    assert order[0][1] == "epinette_gelee" and order[-1][1] == "sapin"        # dataset.py:50 order
    total_flips = 0
    for row in rows[1:]:
        name, wood = row[0], row[1]
        img = frames[(wood, name)]
        if img.shape[0] == img.shape[1]:
            pass                                                               # non-black frames: trim_black keeps all rows
        x = torch.from_numpy(synth.normalize_frame(img))[None]
        lab = predict_labels(oracle_model, x)[0][0].numpy().astype(np.uint8)
        lab = remove_small_zones(lab)                                          # models.py:271
        if exclude_nodes:
            lab[lab == 2] = 1                                                  # models.py:273-276
        got = np.asarray(Image.open(os.path.join(root, "results", "outputs", wood, name)))
        assert got.dtype == np.uint8 and set(np.unique(got)) <= {0, 127, 255}
        flips = int((got != drv.label_png(lab)).sum())
        total_flips += flips
        if flips == 0:
            assert row == drv.stats_row(name, wood, lab.shape[0], lab.shape[1], int((lab == 1).sum()), int((lab == 2).sum()))
        if exclude_nodes:
            assert 255 not in np.unique(got) and row[4] == "0.00000"
    # fp32 parity mode: label PNGs are bit-identical except where an exact logit tie flips a pixel (0-2 per
    # 1024x1024 frame, see test_gpu_configs.py) -- none in these eight small frames
    print("folder label PNG bytes differing from the oracle's over 8 images:", total_flips)
    assert total_flips <= 4, total_flips


def _make_folder(root, sd_np, layout):
    frames = {}
    for wood, name, idx, h, w in layout:
        d = os.path.join(root, "samples", wood)
        os.makedirs(d, exist_ok=True)
        img = synth.make_frame(idx, h, w)
        Image.fromarray(img, mode="RGB").save(os.path.join(d, name))
        frames[(wood, name.replace("bmp", "png"))] = img
    ckpt = os.path.join(root, "best_model.pt")
    torch.save({k: torch.from_numpy(v) for k, v in sd_np.items()}, ckpt)
    return ckpt, frames


def test_batched_folder_equals_one_by_one(tmp_path, sd_np, built_lib):
    """Equal-sized frames ride in batches (and the label PNGs come back through the pinned double buffer): every
    file and the CSV must be what batch = 1 writes.  Two windows, three shapes, a ragged last batch."""
    layout = [("sapin", "s%02d.bmp" % i, 40 + i, 128 if i % 5 else 96, 192) for i in range(11)] + \
             [("epinette_gelee", "e%02d.png" % i, 60 + i, 160, 160) for i in range(5)]
    outs = []
    for batch in (1, 4):
        root = str(tmp_path / ("b%d" % batch))
        ckpt, _ = _make_folder(root, sd_np, layout)
        st = drv.predict_folder(root, ckpt, precision="fp32", device_index=0, batch=batch, window=8)
        assert st["images_total"] == 16 and st["images_this_rank"] == 16
        files = {}
        for wood in ("sapin", "epinette_gelee"):
            for n in sorted(os.listdir(os.path.join(root, "results", "outputs", wood))):
                files[(wood, n)] = np.asarray(Image.open(os.path.join(root, "results", "outputs", wood, n)))
            for n in sorted(os.listdir(os.path.join(root, "processed", "samples", wood))):
                files[("processed", wood, n)] = np.asarray(Image.open(os.path.join(root, "processed", "samples", wood, n)))
        outs.append((files, open(os.path.join(root, "results", "final_stats.csv")).read()))
    assert outs[0][1] == outs[1][1]
    assert outs[0][0].keys() == outs[1][0].keys() and len(outs[0][0]) == 32
    for k in outs[0][0]:
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL broadcast + all_gather with two ranks)")
def test_two_rank_folder_over_rccl(tmp_path, sd_np, built_lib):
    """`python -m neuralbarkcalculator_amd.predict DIR --gpus 2`: the ranks start themselves, rank 0 alone reads the
    checkpoint, the packed weights travel by RCCL broadcast, the rows by all_gather; same files as one GPU."""
    import subprocess
    import sys
    layout = [("sapin", "s%02d.bmp" % i, 40 + i, 128 + 8 * (i % 3), 192) for i in range(9)]
    outs = []
    for gpus in (1, 2):
        root = str(tmp_path / ("g%d" % gpus))
        ckpt, _ = _make_folder(root, sd_np, layout)
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        p = subprocess.run([sys.executable, "-m", "neuralbarkcalculator_amd.predict", root, "--model_path", ckpt, "--gpus", str(gpus)],
                           cwd=repo, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        labs = [np.asarray(Image.open(os.path.join(root, "results", "outputs", "sapin", n)))
                for n in sorted(os.listdir(os.path.join(root, "results", "outputs", "sapin")))]
        outs.append((labs, open(os.path.join(root, "results", "final_stats.csv")).read()))
    assert outs[0][1] == outs[1][1] and len(outs[0][0]) == 9
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("n_images", [10, 1])
def test_two_rank_folder_rehearsal_on_one_gpu(tmp_path, sd_np, built_lib, n_images):
    """The multi-rank folder path on a one-GPU box: two ranks (torch.distributed.run, gloo) that both use cuda:0.
    Rank 0 alone reads the checkpoint, the packed weights travel by broadcast, each rank predicts its contiguous
    pixel-balanced shard, the rows are gathered and rank 0 writes the CSV: same files as one rank.  In the f16x2 mode (the
    CLI's default), whose non-finite flag travels through one more scalar all-reduce before the gather."""
    import subprocess
    import sys
    layout = [("epinette_gelee" if i < 4 else "sapin", "s%02d.bmp" % i, 60 + i, 96 + 8 * (i % 4), 160) for i in range(n_images)]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # n_images = 1: the second rank's shard is empty
    outs = []
    for world in (1, 2):
        root = str(tmp_path / ("w%d" % world))
        ckpt, _ = _make_folder(root, sd_np, layout)
        code = ("import sys, torch.distributed as dist\n"
                "sys.path.insert(0, %r)\n"
                "from neuralbarkcalculator_amd import predict\n"
                "dist.init_process_group('gloo')\n"
                "st = predict.predict_folder(%r, %r, precision='f16x2', device_index=0)\n"
                "assert st['world'] == %d and st['images_total'] == %d\n"
                "dist.destroy_process_group()\n" % (repo, root, ckpt, world, n_images))
        script = tmp_path / ("run%d.py" % world)
        script.write_text(code)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                            "--master-addr", "127.0.0.1", "--master-port", str(29611 + world), str(script)],
                           cwd=repo, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        labs = {}
        for wood in ("epinette_gelee", "sapin"):
            d = os.path.join(root, "results", "outputs", wood)
            for n in sorted(os.listdir(d)) if os.path.isdir(d) else []:
                labs[(wood, n)] = np.asarray(Image.open(os.path.join(d, n)))
        outs.append((labs, open(os.path.join(root, "results", "final_stats.csv")).read()))
    assert outs[0][1] == outs[1][1] and len(outs[0][0]) == n_images and outs[0][0].keys() == outs[1][0].keys()
    for k in outs[0][0]:
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k


def test_only_preprocess_cli_writes_what_the_numpy_form_writes(tmp_path, built_lib):
    """`predict.py --only_preprocess` (predict.py:53-55): raw scans larger than 1024 are resized on the device; the
    processed PNGs hold the bytes of the numpy restatement (itself pinned to scikit-image's output by the fixtures)."""
    import subprocess
    import sys
    root = str(tmp_path / "raw")
    os.makedirs(os.path.join(root, "samples", "sapin"))
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, size=(1300, 1500, 3), dtype=np.uint8)
    square = rng.integers(0, 256, size=(1200, 1200, 3), dtype=np.uint8)
    square[:300] = 0                                           # cut by trim_black
    small = synth.make_frame(3, 200, 320)
    for name, img in (("big.bmp", big), ("square.bmp", square), ("small.bmp", small)):
        Image.fromarray(img, mode="RGB").save(os.path.join(root, "samples", "sapin", name))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "neuralbarkcalculator_amd.predict", root, "--only_preprocess"],
                       cwd=repo, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert not os.path.exists(os.path.join(root, "results"))
    for name, img in (("big.png", big), ("square.png", square), ("small.png", small)):
        got = np.asarray(Image.open(os.path.join(root, "processed", "samples", "sapin", name)))
        want = drv.preprocess_image(img, 1024)
        assert got.shape == want.shape and np.array_equal(got, want), name
    assert np.asarray(Image.open(os.path.join(root, "processed", "samples", "sapin", "square.png"))).shape[0] < 1024


def test_empty_and_single_image_folders(tmp_path, sd_np, built_lib):
    """Edge cases of the driver (the reference walks whatever samples/ holds): a wood-type folder with no image
    at all gives a CSV with only the header; a single image gives one row; files that are not images are ignored."""
    root = str(tmp_path / "empty")
    os.makedirs(os.path.join(root, "samples", "sapin"))
    ckpt = os.path.join(root, "best_model.pt")
    torch.save({k: torch.from_numpy(v) for k, v in sd_np.items()}, ckpt)
    open(os.path.join(root, "samples", "sapin", "notes.txt"), "w").write("not an image")
    st = drv.predict_folder(root, ckpt, device_index=0)
    assert st["images_total"] == 0 and st["batches"] == 0
    lines = open(os.path.join(root, "results", "final_stats.csv")).read().strip().splitlines()
    assert len(lines) == 1
    Image.fromarray(synth.make_frame(9, 72, 96), mode="RGB").save(os.path.join(root, "samples", "sapin", "one.bmp"))
    st = drv.predict_folder(root, ckpt, device_index=0)
    assert st["images_total"] == 1 and st["batches"] == 1
    lines = open(os.path.join(root, "results", "final_stats.csv")).read().strip().splitlines()
    assert len(lines) == 2 and lines[1].startswith("one.png\tsapin\t")
    lab = np.asarray(Image.open(os.path.join(root, "results", "outputs", "sapin", "one.png")))
    assert lab.shape == (72, 96) and set(np.unique(lab).tolist()) <= {0, 127, 255}


def test_predict_cli_matches_the_library_call(tmp_path, sd_np, built_lib):
    """`python -m neuralbarkcalculator_amd.predict DIR --model_path ... --exclude_nodes` (the reference's
    `predict.py DIR --exclude_nodes`) writes the files the library call writes."""
    import subprocess
    import sys
    layout = [("sapin", "a.bmp", 11, 88, 120), ("sapin", "b.bmp", 12, 88, 120), ("epinette_gelee", "c.png", 13, 64, 200)]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for how in ("cli", "call"):
        root = str(tmp_path / how)
        ckpt, _ = _make_folder(root, sd_np, layout)
        if how == "cli":
            p = subprocess.run([sys.executable, "-m", "neuralbarkcalculator_amd.predict", root, "--model_path", ckpt, "--exclude_nodes",
                                "--streams", "2"], cwd=repo, capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, p.stderr[-3000:]
            assert "predicted 3 images" in p.stdout
        else:
            drv.predict_folder(root, ckpt, precision="f16x2", exclude_nodes=True, device_index=0, streams=2)    # the CLI's default mode
        labs = {}
        for wood in ("sapin", "epinette_gelee"):
            d = os.path.join(root, "results", "outputs", wood)
            for n in sorted(os.listdir(d)):
                labs[(wood, n)] = np.asarray(Image.open(os.path.join(d, n)))
        outs.append((labs, open(os.path.join(root, "results", "final_stats.csv")).read()))
    assert outs[0][1] == outs[1][1] and outs[0][0].keys() == outs[1][0].keys() and len(outs[0][0]) == 3
    for k in outs[0][0]:
        assert np.array_equal(outs[0][0][k], outs[1][0][k]) and set(np.unique(outs[0][0][k]).tolist()) <= {0, 127}      # --exclude_nodes: no 255
    # the CPU device is the reference's own path
    p = subprocess.run([sys.executable, "-m", "neuralbarkcalculator_amd.predict", str(tmp_path / "cli"), "--device", "cpu"],
                       cwd=repo, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "MI355X" in (p.stderr + p.stdout)


def test_mixed_folder_is_independent_of_batching_and_streams(tmp_path, sd_np, built_lib):
    """Raw scans that need the device resize (two aspect ratios), height-trimmed 1024-wide scans and small images in
    three file formats in one folder: every processed PNG, label PNG and the CSV are the same whether the driver runs
    one image at a time on one stream or batches on several."""
    import hashlib
    import shutil
    root = str(tmp_path / "mixed")
    layout_small = [("sapin", "s%02d.%s" % (i, ("png", "jpg", "bmp")[i % 3]), 200 + i, 40 + 37 * i % 300, 64 + 53 * i % 500) for i in range(8)]
    ckpt, _ = _make_folder(root, sd_np, layout_small)
    for i in range(2):                                  # raw scans: 2048 x 2048 and 2048 x 1548
        f = synth.make_frame(i, 1024, 1024)
        f[: 100 + 40 * i] = 0
        g = np.repeat(np.repeat(f, 2, axis=0), 2, axis=1)
        if i == 1:
            g = g[:, :1548]
        Image.fromarray(np.ascontiguousarray(g), mode="RGB").save(os.path.join(root, "samples", "sapin", "raw%d.bmp" % i))
    os.makedirs(os.path.join(root, "samples", "epinette_gelee"), exist_ok=True)
    for i in range(5):                                  # 1024 x 1024 scans with black bands: trimmed to different heights
        f = synth.make_frame(100 + i, 1024, 1024).copy()
        f[: 150 + 30 * i] = 0
        f[1024 - 200:] = 0
        Image.fromarray(f, mode="RGB").save(os.path.join(root, "samples", "epinette_gelee", "t%d.bmp" % i))

    def digest():
        h = hashlib.sha256()
        for sub in ("processed", "results"):
            for d, _, files in sorted(os.walk(os.path.join(root, sub))):
                for f in sorted(files):
                    p = os.path.join(d, f)
                    h.update(os.path.relpath(p, root).encode())
                    h.update(np.asarray(Image.open(p)).tobytes() if f.endswith(".png") else open(p, "rb").read())
        return h.hexdigest()
    seen = set()
    for kw in (dict(batch=1, streams=1), dict(batch=2, streams=4), dict(batch=4, streams=3, window=4)):
        shutil.rmtree(os.path.join(root, "results"), ignore_errors=True)
        shutil.rmtree(os.path.join(root, "processed"), ignore_errors=True)
        st = drv.predict_folder(root, ckpt, precision="fp32", device_index=0, **kw)
        assert st["images_total"] == 15
        seen.add(digest())
    assert len(seen) == 1
    heights = sorted(np.asarray(Image.open(os.path.join(root, "processed", "samples", "epinette_gelee", "t%d.png" % i))).shape[0] for i in range(5))
    assert heights == [554, 584, 614, 644, 674]          # trim_black cut the bands


def test_cli_falls_back_to_fp32_when_f16x2_overflows(tmp_path, sd_np, built_lib):
    """--precision auto (the default): weights that drive an activation beyond f16's range make the f16x2 run raise
    NonFiniteLogits (the library's sticky flag), and the CLI runs the folder again on the f32 MFMA: same files as a plain
    fp32 library call; the library call in f16x2 raises."""
    import subprocess
    import sys
    big = overflowing_state_dict(sd_np)
    layout = [("sapin", "a.bmp", 11, 88, 120), ("epinette_gelee", "c.png", 13, 64, 200)]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    root_cli, root_call = str(tmp_path / "cli"), str(tmp_path / "call")
    ckpt, _ = _make_folder(root_cli, big, layout)
    ckpt2, _ = _make_folder(root_call, big, layout)
    p = subprocess.run([sys.executable, "-m", "neuralbarkcalculator_amd.predict", root_cli, "--model_path", ckpt],
                       cwd=repo, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "running the folder again on the f32 MFMA" in p.stdout and "predicted 2 images" in p.stdout
    with pytest.raises(drv.NonFiniteLogits):
        drv.predict_folder(root_call, ckpt2, precision="f16x2", device_index=0)
    # the invalid run leaves no label PNG behind (a crash before the fp32 rerun must not leave invalid masks without a
    # CSV).  (Since round 5 the calibration guard stops this checkpoint on the FIRST image, before any batch: what the
    # preprocessor wrote by then does not depend on the arithmetic and stays; the fp32 run writes the rest.)
    for wood, name in (("sapin", "a.png"), ("epinette_gelee", "c.png")):
        assert not os.path.exists(os.path.join(root_call, "results", "outputs", wood, name))
    assert not os.path.exists(os.path.join(root_call, "results", "final_stats.csv"))
    drv.predict_folder(root_call, ckpt2, precision="fp32", device_index=0)
    assert open(os.path.join(root_cli, "results", "final_stats.csv")).read() == open(os.path.join(root_call, "results", "final_stats.csv")).read()
    for wood, name in (("sapin", "a.png"), ("epinette_gelee", "c.png")):
        a = np.asarray(Image.open(os.path.join(root_cli, "results", "outputs", wood, name)))
        b = np.asarray(Image.open(os.path.join(root_call, "results", "outputs", wood, name)))
        assert np.array_equal(a, b)


def test_f16x2_overflow_abandons_the_folder_at_the_first_batches(tmp_path, sd_np, built_lib):
    """The sticky non-finite word rides back with every batch's labels (nbc_nonfinite_peek_async), so a folder whose weights
    f16x2 cannot carry is abandoned after the batches that were in flight when the first one came back -- not after the
    last image (what's weak 5) -- and no label PNG of the run stays on disk."""
    big = overflowing_state_dict(sd_np)
    layout = [("sapin", "s%02d.bmp" % i, 40 + i, 64, 96) for i in range(24)]
    root = str(tmp_path / "many")
    ckpt, _ = _make_folder(root, big, layout)
    with pytest.raises(drv.NonFiniteLogits) as e:
        # (calibrate=False: the calibration guard would stop this checkpoint on the first image, before any batch; the word that
        # rides back with every batch is the second line, for an image later in the folder)
        drv.predict_folder(root, ckpt, precision="f16x2", device_index=0, batch=1, streams=2, window=4, calibrate=False)
    assert e.value.images_this_rank == 24 and 1 <= e.value.batches_run <= 8, e.value.batches_run    # two windows at most
    out = os.path.join(root, "results", "outputs", "sapin")
    assert not os.path.isdir(out) or os.listdir(out) == []
    st = drv.predict_folder(root, ckpt, precision="fp32", device_index=0, batch=1, streams=2, window=4)
    assert st["images_total"] == 24 and len(os.listdir(out)) == 24


def test_f16x2_refuses_weights_the_packer_flags_before_any_forward(tmp_path, sd_np, built_lib):
    """a weight row beyond the reach of the f16x2 row normalisation (largest |w| below 2^-51) keeps only a few
    bits with finite logits, so no flag could ever report it from the device: nbc_pack_weights reports it
    (NBC_PACK_ROW_CLAMPED, in the blob's trailer), the folder driver leaves in f16x2 before the first forward -- the way it
    leaves on the non-finite word, so --precision auto runs the folder on the f32 MFMA -- and fp32 takes the checkpoint."""
    sd = dict(sd_np)
    sd["backbone.layer2.1.conv2.weight"] = sd_np["backbone.layer2.1.conv2.weight"] * np.float32(2.0 ** -70)
    layout = [("sapin", "a.bmp", 11, 88, 120)]
    root = str(tmp_path / "flagged")
    ckpt, _ = _make_folder(root, sd, layout)
    with pytest.raises(drv.NonFiniteLogits) as e:
        drv.predict_folder(root, ckpt, precision="f16x2", device_index=0)
    assert e.value.batches_run == 0 and "NBC_PACK flags 1" in str(e.value)
    st = drv.predict_folder(root, ckpt, precision="fp32", device_index=0)
    assert st["images_total"] == 1


def _stats_do_not_describe_the_data(sd):
    """A checkpoint whose BatchNorm PROMISES an ordinary tensor (gamma of order one) and whose data does not keep the promise:
    layer2.1's conv1 at 2^-22 of its usual size under a BatchNorm with zero mean and bias and unit variance -- the tensor behind
    it peaks around 2^-20.  nbc_pack_weights, which reads the BatchNorm, gives it no power of two; the calibration guard,
    which reads the data, sees it."""
    out = dict(sd)
    out["backbone.layer2.1.conv1.weight"] = sd["backbone.layer2.1.conv1.weight"] * np.float32(2.0 ** -22)
    for k, v in (("running_mean", 0.0), ("bias", 0.0), ("running_var", 1.0)):
        out["backbone.layer2.1.bn1." + k] = np.full_like(sd["backbone.layer2.1.bn1." + k], v)
    return out


def test_f16x2_calibration_guard(tmp_path, sd_np, built_lib):
    """The underflow counterpart of the non-finite flag (what's missing 6).  nbc_activation_peaks: the
    largest stored value of every activation tensor on a real frame.  An ordinary checkpoint keeps every tensor between 2^-8 and
    2^14 (printed); one whose data breaks its BatchNorm's promise (a tensor at 2^-20) is caught before any batch runs: the
    folder driver leaves in f16x2 the way it leaves on the non-finite word (--precision auto then takes the f32 MFMA), and
    fp32 runs the checkpoint."""
    from neuralbarkcalculator_amd.model import FCNResNet50
    x = torch.from_numpy(np.stack([synth.make_input(3, 256, 320)])).to("cuda:0")
    m = FCNResNet50("f16x2").load_state_dict(sd_np).to("cuda:0")
    peaks = m.activation_peaks(x)
    ok, bad = FCNResNet50.f16x2_range_ok(peaks)
    lo, hi = min(peaks.items(), key=lambda kv: kv[1]), max(peaks.items(), key=lambda kv: kv[1])
    print("ordinary checkpoint: stored peaks from %s %.3g to %s %.3g" % (lo[0], lo[1], hi[0], hi[1]))
    assert ok and len(peaks) == 54 and lo[1] > 2.0 ** -6 and hi[1] < 2.0 ** 10, bad
    sd = _stats_do_not_describe_the_data(sd_np)
    m2 = FCNResNet50("f16x2").load_state_dict(sd).to("cuda:0")
    assert m2.pack_flags == 0 and m2.activation_exponent("backbone.layer2.1.conv1") == 0     # the packer sees nothing
    ok2, bad2 = FCNResNet50.f16x2_range_ok(m2.activation_peaks(x))
    assert not ok2 and "backbone.layer2.1.conv1" in bad2 and bad2["backbone.layer2.1.conv1"] < 2.0 ** -12, bad2
    assert not m2.nonfinite_seen()                                                            # ... and neither does the flag
    layout = [("sapin", "a.bmp", 11, 88, 120), ("sapin", "b.bmp", 12, 88, 120)]
    root = str(tmp_path / "cal")
    ckpt, _ = _make_folder(root, sd, layout)
    with pytest.raises(drv.NonFiniteLogits) as e:
        drv.predict_folder(root, ckpt, precision="f16x2", device_index=0)
    assert e.value.batches_run == 0 and "calibration" in str(e.value) and "layer2.1.conv1" in str(e.value)
    st = drv.predict_folder(root, ckpt, precision="fp32", device_index=0)
    assert st["images_total"] == 2
