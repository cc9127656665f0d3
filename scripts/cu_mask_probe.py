#!/usr/bin/env python3
"""Two forwards in flight: do they run better side by side on the whole chip (what bench.py and the folder driver do) or each
on its own half of the CUs (hipExtStreamCreateWithCUMask)?  One process, one box, the variants interleaved.
  gpurun -- 'python scripts/cu_mask_probe.py [--precision f16x2] [--batch 1] [--steps 40]'
Masks tried: none; the low / high half of the mask bits; even / odd bits; even / odd groups of 8 bits (which of these is
"four XCDs each" depends on how the runtime numbers the CUs of an eight-XCD part -- the probe measures, it does not assume)."""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from neuralbarkcalculator_amd import synth
from neuralbarkcalculator_amd.model import FCNResNet50

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="f16x2")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int
ncu = torch.cuda.get_device_properties(dev).multi_processor_count
words = (ncu + 31) // 32


def masked_stream(bits):
    arr = (C.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), words, arr)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask: %d" % rc)
    return torch.cuda.ExternalStream(h.value, device=dev)


variants = {
    "none": None,
    "halves": ([b for b in range(ncu) if b < ncu // 2], [b for b in range(ncu) if b >= ncu // 2]),
    "even_odd": ([b for b in range(ncu) if b % 2 == 0], [b for b in range(ncu) if b % 2 == 1]),
    "groups_of_8": ([b for b in range(ncu) if (b // 8) % 2 == 0], [b for b in range(ncu) if (b // 8) % 2 == 1]),
    "low_4_of_8": ([b for b in range(ncu) if b % 8 < 4], [b for b in range(ncu) if b % 8 >= 4]),
}
sd = synth.make_state_dict("trained_like", seed=7)
x = torch.from_numpy(np.stack([synth.make_frame(i, 1024, 1024) for i in range(args.batch)])).to(dev)
models = [FCNResNet50(args.precision).load_state_dict(sd).to(dev) for _ in range(2)]
for m in models:
    m.reserve(args.batch, 1024, 1024)
streams = {}
for name, masks in variants.items():
    streams[name] = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)] if masks is None else [masked_stream(masks[0]), masked_stream(masks[1])]


def run(name, steps):
    ss = streams[name]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(ss[i % 2]):
            models[i % 2].predict_labels(x, labels_dtype=torch.uint8)
    torch.cuda.synchronize()
    return steps * args.batch / (time.perf_counter() - t0)


print("%d CUs, %s batch %d, two forwards in flight, %d forwards per measurement" % (ncu, args.precision, args.batch, args.steps))
for name in variants:
    run(name, 6)
res = {name: [] for name in variants}
for _ in range(args.rounds):
    for name in variants:
        res[name].append(run(name, args.steps))
for name in variants:
    print("  %-12s %s  median %.1f images/s" % (name, " ".join("%.1f" % v for v in res[name]), float(np.median(res[name]))))
# one masked stream alone: what half the chip gives one forward
for name in ("none", "halves"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[name][0]):
        for i in range(args.steps // 2):
            models[0].predict_labels(x, labels_dtype=torch.uint8)
    torch.cuda.synchronize()
    print("  one forward at a time on stream 0 of '%s': %.1f images/s" % (name, (args.steps // 2) * args.batch / (time.perf_counter() - t0)))
