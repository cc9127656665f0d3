#!/usr/bin/env python3
"""Whole-forward A/B of library builds in ONE process on one box: images/s of the headline configuration (f16x2, batch 1,
`--streams` forwards in flight, default tiles) for two or more builds of libnbc_hip.so, rounds interleaved, every
measurement on models of its own (scripts/ab_tiles.py measures per layer on one stream; this is the bench's region).
  gpurun -- 'python scripts/ab_forward.py --libs tools/_bin/libnbc_base.so neuralbarkcalculator_amd/libnbc_hip.so'"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from neuralbarkcalculator_amd import _lib, synth
from neuralbarkcalculator_amd.model import FCNResNet50

ap = argparse.ArgumentParser()
ap.add_argument("--libs", nargs="+", required=True)
ap.add_argument("--precision", default="f16x2")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--streams", type=int, default=2)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--frames", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda", 0)
sd = synth.make_state_dict("trained_like", seed=7)
xs = [torch.from_numpy(np.stack([synth.make_input(i * args.batch + j, 1024, 1024) for j in range(args.batch)])).to(dev) for i in range(args.frames)]


def models_on(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, argtypes
    keep = _lib._lib
    _lib._lib = lib
    try:
        m = FCNResNet50(args.precision).load_state_dict(sd).to(dev)
        ms = [m] + [m.clone_shared() for _ in range(args.streams - 1)]
    finally:
        _lib._lib = keep if keep is not None else lib
    for q in ms:
        q.reserve(args.batch, 1024, 1024)
    return ms


streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(args.streams - 1)]
res = [[] for _ in args.libs]
for rnd in range(args.rounds):
    for j in range(len(args.libs)):
        k = (j + rnd) % len(args.libs)
        ms = models_on(args.libs[k])

        def step(i):
            with torch.cuda.stream(streams[i % args.streams]):
                ms[i % args.streams].predict_labels(xs[i % len(xs)], labels_dtype=torch.uint8)
        for i in range(10):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        res[k].append(args.batch * args.steps / (time.perf_counter() - t0))
        for q in ms[1:]:
            q._destroy()
        ms[0]._destroy()
for k, p in enumerate(args.libs):
    print("%-28s images/s per round: %s  median %.1f" % (os.path.basename(p), " ".join("%.1f" % v for v in res[k]), float(np.median(res[k]))))
base = float(np.median(res[0]))
for k in range(1, len(args.libs)):
    print("%s vs %s: %+.2f %%" % (os.path.basename(args.libs[k]), os.path.basename(args.libs[0]), 100 * (float(np.median(res[k])) / base - 1)))
