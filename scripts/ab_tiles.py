#!/usr/bin/env python3
"""A/B of library builds in ONE process on one box (cdna_hip_programming.md rule 24): per-layer durations of chosen conv
tiles for two or more builds of libnbc_hip.so, rounds interleaved, every measurement on a model of its own that takes the
addresses the previous one gave back (see --resident).
  gpurun -- 'python scripts/ab_tiles.py --libs neuralbarkcalculator_amd/libnbc_hip.so tools/_bin/libnbc_x.so --tiles 1,7,14,17'
Each build gets its own model object (ctypes loads every path as its own library instance).  Timing only: an experimental
build may compute garbage."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from neuralbarkcalculator_amd import _lib, synth
from neuralbarkcalculator_amd.model import FCNResNet50

ap = argparse.ArgumentParser()
ap.add_argument("--libs", nargs="+", required=True)
ap.add_argument("--tiles", default="-1")
ap.add_argument("--precision", default="f16x2")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--height", type=int, default=1024)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--forwards", type=int, default=6)
ap.add_argument("--layers", default="", help="comma-separated substrings; default: every conv")
ap.add_argument("--out", default=None)
ap.add_argument("--resident", action="store_true",
                help="one model per build, alive side by side (the old behaviour).  Default: every measurement builds its model and "
                     "destroys it again, builds taken in an order that rotates from round to round -- side by side each model has "
                     "its own workspace addresses, and identical kernels then differ by up to 1 %% with their place in the list")
args = ap.parse_args()
dev = torch.device("cuda", 0)
sd = synth.make_state_dict("trained_like", seed=7)
x = torch.from_numpy(np.stack([synth.make_frame(i, args.height, 1024) for i in range(args.batch)])).to(dev)


def model_on(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, argtypes
    keep = _lib._lib
    _lib._lib = lib
    try:
        m = FCNResNet50(args.precision).load_state_dict(sd).to(dev)
    finally:
        _lib._lib = keep if keep is not None else lib
    m.reserve(args.batch, args.height, 1024)
    return m


models = [model_on(p) for p in args.libs] if args.resident else [None] * len(args.libs)
tiles = [int(t) for t in args.tiles.split(",")]
want = [s for s in args.layers.split(",") if s]


def records(m):
    for _ in range(2):
        m.predict_labels(x, labels_dtype=torch.uint8)
    torch.cuda.synchronize()
    m.set_profiling(True)
    for _ in range(args.forwards):
        m.predict_labels(x, labels_dtype=torch.uint8)
    torch.cuda.synchronize()
    r = m.op_records()
    m.set_profiling(False)
    return [q for q in r if q["kernel"] == "conv_dma"]


out = {}
for tile in tiles:
    per = [[] for _ in models]
    for rnd in range(args.rounds):
        for j in range(len(models)):
            k = (j + rnd) % len(models)
            m = models[k] if args.resident else model_on(args.libs[k])
            m.set_conv_tile(tile)
            per[k].append(records(m))
            if not args.resident:
                m._destroy()
                del m
    names = [q["name"] for q in per[0][0]]
    # by NAME: a build may launch its layers in another order
    med = [[float(np.median([next(q["ms"] for q in rnd if q["name"] == n) for rnd in per[k]])) * 1e3 for n in names] for k in range(len(models))]
    out[tile] = dict(names=names, us=med)
    print("tile %d: conv launches per forward, us: " % tile + "  ".join("%s %.1f" % (os.path.basename(p), sum(med[k])) for k, p in enumerate(args.libs)))
    for i, n in enumerate(names):
        if want and not any(s in n for s in want):
            continue
        print("  %-34s " % n + "  ".join("%7.1f" % med[k][i] for k in range(len(models))) +
              ("   %+5.1f %%" % (100 * (med[1][i] / med[0][i] - 1)) if len(models) > 1 else ""))
    sys.stdout.flush()
if args.out:
    json.dump(dict(libs=args.libs, tiles=out), open(args.out, "w"))
