#!/usr/bin/env python3
"""Generate tests/golden/* from the CPU oracle (run in the build container; committed output).

The reference cannot be imported here (torchvision is absent: SURVEY.md section 8c) and ships no
fixtures, so these vectors come from ``oracle/fcn_resnet50_oracle.py`` on deterministic
synthetic weights/frames (``neuralbarkcalculator_amd/synth.py``).  They guard against drift
(torch version, GPU box vs this container, refactors); they are not an independent check of
the topology -- tests/test_oracle.py holds those pins.

Usage: python scripts/make_goldens.py            (about a minute on 8 cores)
"""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuralbarkcalculator_amd import synth  # noqa: E402
from oracle.fcn_resnet50_oracle import OracleFCNResNet50, layer_outputs, predict_labels  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# (case name, frame indices, H, W)
CASES = [
    ("c128", [3], 128, 128),
    ("b2_256", [4, 5], 256, 256),
    ("h520", [6], 520, 1024),
    ("odd_h", [7], 203, 1024),     # trimmed height not a multiple of 8
    ("full1024", [0], 1024, 1024),
]


def sample_points(n, k=8):
    """k fixed flat indices into a tensor of n elements."""
    return [(i * 2654435761 + 12345) % n for i in range(k)]


def main():
    torch.set_num_threads(os.cpu_count())
    os.makedirs(OUT, exist_ok=True)
    model = OracleFCNResNet50()
    sd = synth.make_state_dict("trained_like", seed=7)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})

    for name, frames, h, w in CASES:
        x = torch.from_numpy(np.stack([synth.make_input(i, h, w) for i in frames]))
        labels, counts, logits, lowres = predict_labels(model, x)
        top2 = torch.topk(logits, 2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1])
        pts = sample_points(logits.numel())
        np.savez_compressed(
            os.path.join(OUT, f"{name}.npz"),
            frames=np.asarray(frames), hw=np.asarray([h, w]),
            lowres=lowres.numpy(), counts=counts.numpy(),
            logits_points=np.asarray(pts), logits_values=logits.flatten()[pts].numpy(),
            margin_hist=np.histogram(margin.numpy(), bins=[0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1e9])[0])
        for b in range(len(frames)):
            lab = labels[b].numpy().astype(np.uint8)
            Image.fromarray(lab * 127 + (lab == 2), mode="L").save(
                os.path.join(OUT, f"{name}_labels{b}.png"))   # {0,127,255} like models.py:350-356
        print(name, "counts", counts.tolist())

    # per-conv-unit checksums on the smallest case
    x = torch.from_numpy(synth.make_input(3, 128, 128))[None]
    outs = layer_outputs(model, x)
    names, stats, pts_all = [], [], []
    for k, v in outs.items():
        v = v.double()
        pts = sample_points(v.numel())
        names.append(k)
        stats.append([v.sum().item(), v.abs().sum().item(), v.max().item(), v.min().item()])
        pts_all.append(v.flatten()[pts].numpy())
    np.savez_compressed(os.path.join(OUT, "c128_layers.npz"), names=np.asarray(names),
                        stats=np.asarray(stats), points=np.asarray(pts_all))
    print("layers", len(names))


if __name__ == "__main__":
    main()
