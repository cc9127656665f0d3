#!/usr/bin/env python3
"""Per-layer duration of EVERY conv tile shape on a list of (precision, batch, height) cases: the data behind the
default-tile cost model of csrc/conv_igemm_dma.hip (choose_conv_tile).  Width is 1024 (the preprocessor's output).
  gpurun -- 'python scripts/tile_model_probe.py gpurun_out/tiles_a.json "fp32:1:1024 fp32:1:640 bf16:8:1024 ..."'
For each case: the plan's default tiles and their per-launch times, the times with each of the 13 tiles forced
wherever it fits the layer (HIP events between launches, mean of 8 forwards), and what nbc_autotune picks."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from neuralbarkcalculator_amd import synth
from neuralbarkcalculator_amd.model import FCNResNet50

N_TILES = 18
out_path = sys.argv[1]
cases = [(p, int(b), int(h)) for p, b, h in (c.split(":") for c in sys.argv[2].split())]
dev = torch.device("cuda", 0)
sd = synth.make_state_dict("trained_like", seed=7)
out = {}
for prec, batch, h in cases:
    m = FCNResNet50(prec).load_state_dict(sd).to(dev)
    x = torch.from_numpy(np.stack([synth.make_frame(i, h, 1024) for i in range(batch)])).to(dev)

    def records(n=8):
        for _ in range(2):
            m.predict_labels(x, labels_dtype=torch.uint8)
        torch.cuda.synchronize()
        m.set_profiling(True)
        for _ in range(n):
            m.predict_labels(x, labels_dtype=torch.uint8)
        torch.cuda.synchronize()
        r = m.op_records()
        m.set_profiling(False)
        return [q for q in r if q["kernel"] == "conv_dma"]

    m.reserve(batch, h, 1024)
    default_tiles = m.plan_tiles()
    per_tile = {}
    for tile in range(N_TILES):
        m.set_conv_tile(tile)            # forced wherever it fits the layer; elsewhere the planned tile runs
        per_tile[tile] = [q["ms"] for q in records()]
    m.set_conv_tile(-1)
    base = records()
    tuned = m.autotune(x, reps=5)
    out["%s_b%d_h%d" % (prec, batch, h)] = dict(
        names=[q["name"] for q in base], k=[q["k"] for q in base], cout=[q["cout"] for q in base],
        flops=[q["flops"] for q in base], default_tiles=default_tiles, default_ms=[q["ms"] for q in base],
        tuned_tiles=tuned, per_tile=per_tile)
    print(prec, batch, h, "default tiles: %.3f ms per forward (conv launches)" % sum(q["ms"] for q in base), flush=True)
json.dump(out, open(out_path, "w"))
