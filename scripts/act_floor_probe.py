#!/usr/bin/env python3
"""f16x2's activation floor, before and after the per-tensor powers of two of nbc_pack_weights: logit error against a float64
evaluation (of the logit range) for the seed-7 network with its scale-free tensors moved by a power of two
(tests/conftest.py::rescale_activations), for one or more builds of the library in one process.
  tools/build_variant.sh noactexp "-DNBC_NO_ACT_EXP"      # the packer without the powers (round 4's behaviour)
  gpurun -- 'python scripts/act_floor_probe.py tools/_bin/libnbc_noactexp.so neuralbarkcalculator_amd/libnbc_hip.so'
(profiles/r05_small_activation_floor_before_fix.log)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
from conftest import rescale_activations
from neuralbarkcalculator_amd import _lib, synth
from neuralbarkcalculator_amd.model import FCNResNet50
from oracle.fcn_resnet50_oracle import OracleFCNResNet50, predict_labels

torch.set_num_threads(16)
sd0 = synth.make_state_dict("trained_like", seed=7)
x = torch.from_numpy(np.stack([synth.make_input(i, 256, 320) for i in (81, 82)]))


def model_on(path, prec, sd):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, argtypes
    keep = _lib._lib
    _lib._lib = lib
    try:
        m = FCNResNet50(prec).load_state_dict(sd).to("cuda:0")
    finally:
        _lib._lib = keep if keep is not None else lib
    return m


cases = [(None, 0), ("internal", -12), ("internal", -16), ("internal", -20), ("stream", -16), ("stream", -20), ("all", -20),
         ("all", -24), ("all", 12)]
print("logit error / logit range against float64, 2 frames of 256x320 (tolerance of the tests: 5e-6); 'nonfinite' = the sticky flag")
for where, l2 in cases:
    sd = sd0 if where is None else rescale_activations(sd0, 2.0 ** l2, where)
    o64 = OracleFCNResNet50()
    o64.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ref = predict_labels(o64.double(), x.double())[2]
    o32 = OracleFCNResNet50()
    o32.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    rng = float(ref.abs().max())
    line = "%-9s x 2^%-3d  CPU f32 oracle %.2e" % (where or "as is", l2, float((predict_labels(o32, x)[2].double() - ref).abs().max()) / rng)
    for path in sys.argv[1:]:
        for prec in ("f16x2", "fp32"):
            m = model_on(path, prec, sd)
            y = m(x.to("cuda:0"))
            torch.cuda.synchronize()
            bad = m.nonfinite_seen()
            err = float((y.cpu().double() - ref).abs().max()) / rng
            line += "  | %s %s %s" % (os.path.basename(path).replace("libnbc_", "").replace(".so", ""), prec, "nonfinite" if bad else "%.2e" % err)
            m._destroy()
    print(line, flush=True)
