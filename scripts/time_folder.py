#!/usr/bin/env python3
"""End-to-end folder prediction on one GPU (BASELINE.json configs[3] per rank; next rows N1-N3): a synthetic
folder of 1024x1024 .bmp samples -> decode, preprocess, processed/ PNG, forward + remove_small_zones, label
PNG, CSV.  usage: python scripts/time_folder.py [n_images=1000] [precisions=bf16,fp32] [ragged]
"ragged": every scan has black bands at the top and bottom, so that trim_black leaves heights of 520-730 rows (what
the reference's real folders look like: res/*.png): hundreds of distinct image shapes in one folder.
"raw": raw scans of 4096x4096 pixels with the same black bands (48 MB per .bmp): the whole preprocessor runs (cubic
resize to 1024x1024 on the GPU, trim_black, processed/ PNG) before the forward.
NBC_HOST_WORKERS sets the host thread pool (default 16, the GPU box's CPU share per GPU); NBC_STREAMS / NBC_BATCH override
the driver's batches in flight and frames per batch."""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from concurrent.futures import ThreadPoolExecutor
from PIL import Image
from neuralbarkcalculator_amd import predict as drv, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
precisions = (sys.argv[2] if len(sys.argv) > 2 else "bf16,fp32").split(",")
raw = len(sys.argv) > 3 and sys.argv[3] == "raw"
ragged = raw or (len(sys.argv) > 3 and sys.argv[3] == "ragged")
distinct = min(n, 40)
root = tempfile.mkdtemp(prefix="nbc_folder_")
try:
    t0 = time.perf_counter()
    woods = ("epinette_gelee", "epinette_non_gelee", "sapin")
    for wood in woods:
        os.makedirs(os.path.join(root, "samples", wood))
    with ThreadPoolExecutor(16) as pool:
        frames = list(pool.map(lambda i: synth.make_frame(i, 1024, 1024), range(distinct)))

        def scan(i):
            f = frames[i % distinct]
            if not ragged:
                return f
            g = f.copy()
            band = 294 + (i * 37) % 211                  # black rows in all: 294..504 -> 520..730 rows stay
            top = (i * 13) % (band + 1)
            g[:top] = 0
            g[1024 - (band - top):] = 0
            if raw:
                g = np.repeat(np.repeat(g, 4, axis=0), 4, axis=1)
            return g
        list(pool.map(lambda i: Image.fromarray(scan(i), mode="RGB").save(
            os.path.join(root, "samples", woods[i % 3], "f%04d.bmp" % i)), range(n)))
    ckpt = os.path.join(root, "best_model.pt")
    torch.save({k: torch.from_numpy(v) for k, v in synth.make_state_dict("trained_like", seed=7).items()}, ckpt)
    print(f"folder of {n} synthetic {'4096x4096' if raw else '1024x1024'} .bmp samples ({distinct} distinct frames{', black bands: 520-730 rows after trim_black' if ragged else ''}) made in {time.perf_counter() - t0:.1f} s; "
          f"host workers {drv._host_workers()}, cores available {len(os.sched_getaffinity(0))}", flush=True)
    for prec in precisions:
        for rep in range(2):                  # second run: page cache and allocator warm
            shutil.rmtree(os.path.join(root, "results"), ignore_errors=True)
            shutil.rmtree(os.path.join(root, "processed"), ignore_errors=True)
            t0 = time.perf_counter()
            st = drv.predict_folder(root, ckpt, precision=prec, device_index=0, window=int(os.environ.get("NBC_WINDOW", "64")),
                                    streams=int(os.environ["NBC_STREAMS"]) if "NBC_STREAMS" in os.environ else None,
                                    batch=int(os.environ["NBC_BATCH"]) if "NBC_BATCH" in os.environ else None)
            dt = time.perf_counter() - t0
            rows = open(os.path.join(root, "results", "final_stats.csv")).read().count("\n") - 1
            print(f"{prec} run {rep}: {n} images end to end in {dt:.2f} s = {n / dt:.1f} images/s "
                  f"(checkpoint load + weight packing + upload {st['setup_s']:.2f} s included); steady loop {st['images_per_s_loop']:.1f} "
                  f"images/s; CSV rows {rows}; {json.dumps({k: st[k] for k in ('batch', 'batches', 'streams', 'distinct_shapes', 'autotuned_shapes', 'host_workers')})}",
                  flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
