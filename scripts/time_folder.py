#!/usr/bin/env python3
"""End-to-end folder prediction (next rows N1-N3): decode, forward, remove_small_zones, label PNG, CSV.
usage: python scripts/time_folder.py [n_images] ; NBC_HOST_WORKERS sets the host thread pool (default 8, max 32)."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from PIL import Image
from neuralbarkcalculator_amd import predict as drv, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
root = tempfile.mkdtemp(prefix="nbc_folder_")
try:
    for wood in ("epinette_gelee", "sapin"):
        os.makedirs(os.path.join(root, "samples", wood))
    for i in range(n):
        Image.fromarray(synth.make_frame(i, 1024, 1024), mode="RGB").save(
            os.path.join(root, "samples", ("epinette_gelee", "sapin")[i % 2], "f%04d.png" % i))
    ckpt = os.path.join(root, "best_model.pt")
    torch.save({k: torch.from_numpy(v) for k, v in synth.make_state_dict("trained_like", seed=7).items()}, ckpt)
    for workers in (1, 8, 16):
        os.environ["NBC_HOST_WORKERS"] = str(workers)
        shutil.rmtree(os.path.join(root, "results"), ignore_errors=True)
        t0 = time.perf_counter()
        drv.predict_folder(root, ckpt, precision="bf16", device_index=0)
        dt = time.perf_counter() - t0
        print(f"{n} images of 1024x1024, {workers} host workers: {dt:.2f} s end to end = {n / dt:.1f} images/s "
              f"(includes folder set-up, preprocessing copies, weight packing and upload)", flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
