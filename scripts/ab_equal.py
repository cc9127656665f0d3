#!/usr/bin/env python3
"""Do two builds of libnbc_hip.so compute the same bits?  Full-resolution logits and labels of a few frames, every precision.
  gpurun -- 'python scripts/ab_equal.py neuralbarkcalculator_amd/libnbc_hip.so tools/_bin/libnbc_x.so'"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from neuralbarkcalculator_amd import _lib, synth
from neuralbarkcalculator_amd.model import FCNResNet50

dev = torch.device("cuda", 0)
sd = synth.make_state_dict("trained_like", seed=7)


def model_on(path, precision):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, argtypes
    keep = _lib._lib
    _lib._lib = lib
    try:
        return FCNResNet50(precision).load_state_dict(sd).to(dev)
    finally:
        _lib._lib = keep if keep is not None else lib


bad = 0
for precision in ("f16x2", "fp32", "bf16"):
    a, b = model_on(sys.argv[1], precision), model_on(sys.argv[2], precision)
    for idx, h, w in (([0], 1024, 1024), ([3, 4], 200, 328), ([5], 520, 1024)):
        x = torch.from_numpy(np.stack([synth.make_input(i, h, w) for i in idx])).to(dev)
        la, lb = a(x), b(x)
        ya, yb = a.predict_labels(x, labels_dtype=torch.uint8)[0], b.predict_labels(x, labels_dtype=torch.uint8)[0]
        same = torch.equal(la, lb) and torch.equal(ya, yb)
        bad += not same
        print("%s %s x %dx%d: %s (max |logit difference| %.3e)" % (precision, idx, h, w, "identical" if same else "DIFFERENT", float((la - lb).abs().max())))
sys.exit(1 if bad else 0)
