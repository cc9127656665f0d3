#!/usr/bin/env python3
"""Soak run: many overlapped forwards, every result compared bit for bit with the first one of its frame.
Catches rare ordering bugs (raw s_barrier + counted vmcnt waits, ring slot reuse, shared weight blob).
usage: python scripts/soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from neuralbarkcalculator_amd import synth
from neuralbarkcalculator_amd.model import FCNResNet50

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda:0")
sd = synth.make_state_dict("trained_like", seed=7)
t_end = time.time() + budget
total = 0
for prec in ("bf16", "fp32"):
    base = FCNResNet50(prec).load_state_dict(sd).to(dev)
    models = [base] + [base.clone_shared() for _ in range(3)]
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(3)]
    for (n, h, w), objective in (((1, 1024, 1024), "throughput"), ((2, 520, 1024), "latency"), ((1, 264, 200), "throughput"), ((3, 96, 160), "latency")):
        xs = [torch.from_numpy(np.stack([synth.make_input(100 + 7 * k + j, h, w) for j in range(n)])).to(dev) for k in range(4)]
        for m in models:
            m.autotune(xs[0], objective=objective)
        ref = [tuple(t.clone() for t in base.predict_labels(x, return_lowres=True)) for x in xs]
        torch.cuda.synchronize()
        it = 0
        t_shape = time.time() + budget / 8
        while time.time() < min(t_shape, t_end + 5):
            outs = []
            for k in range(4):
                with torch.cuda.stream(streams[k]):
                    outs.append(models[k].predict_labels(xs[(k + it) % 4], return_lowres=True))
            torch.cuda.synchronize()
            for k in range(4):
                for got, want in zip(outs[k], ref[(k + it) % 4]):
                    if not torch.equal(got, want):
                        print("MISMATCH", prec, (n, h, w), "iteration", it, "stream", k)
                        sys.exit(1)
            it += 1
            total += 4
        print(prec, (n, h, w), objective, "iterations", it, flush=True)
print("soak ok:", total, "forwards, all bit-identical to the first result of their frame")
