#!/usr/bin/env python3
"""Experiment: partition the chip between streams with hipExtStreamCreateWithCUMask (two halves of
128 CUs, forwards confined to one half each) against the default of four unmasked streams."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from neuralbarkcalculator_amd import synth
from neuralbarkcalculator_amd.model import FCNResNet50

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)

def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[int(sum(1 << b for b in range(32) if bits[w * 32 + b])) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)

sd = synth.make_state_dict("trained_like", seed=7)
base = FCNResNet50("bf16").load_state_dict(sd).to(dev)
xs = [torch.from_numpy(synth.make_input(k, 1024, 1024))[None].to(dev) for k in range(4)]

def run(streams, label, objective="throughput", steps=96):
    models = [base] + [base.clone_shared() for _ in range(len(streams) - 1)]
    for m, s in zip(models, streams):
        with torch.cuda.stream(s):
            m.autotune(xs[0], objective=objective)
    torch.cuda.synchronize()
    for i in range(8):
        with torch.cuda.stream(streams[i % len(streams)]):
            models[i % len(streams)].predict_labels(xs[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % len(streams)
        with torch.cuda.stream(streams[k]):
            models[k].predict_labels(xs[i % 4])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: {steps / dt:.1f} images/s", flush=True)

plain = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(3)]
run(plain, "4 unmasked streams")
for name, lo in (("first/second 128 mask bits", [b < 128 for b in range(256)]),
                 ("even/odd mask bits", [b % 2 == 0 for b in range(256)]),
                 ("alternating groups of 32 bits", [(b // 32) % 2 == 0 for b in range(256)])):
    hi = [not v for v in lo]
    a1, a2, b1, b2 = masked_stream(lo), masked_stream(lo), masked_stream(hi), masked_stream(hi)
    run([a1], f"{name}: ONE masked stream alone (latency objective)", "latency", 48)
    run([a1, b1], f"{name}: one stream per half")
    run([a1, b1, a2, b2], f"{name}: two streams per half")
run(plain, "4 unmasked streams (again)")
