#!/usr/bin/env python3
"""Label parity of the two f32-grade modes on MORE frames than the tests hold: n synthetic 1024x1024 frames, each through
f16x2, the f32 MFMA mode, the CPU f32 oracle and a float64 evaluation of the same network (the adjudicator).
Per frame and mode: max logit error against float64, labels differing from the CPU oracle, and for each such pixel
which side float64 agrees with and its float64 top-2 margin.   python scripts/adjudicate_frames.py [n=6] [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neuralbarkcalculator_amd import synth
from neuralbarkcalculator_amd.model import FCNResNet50
from oracle.fcn_resnet50_oracle import OracleFCNResNet50, predict_labels      # checker

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
out_path = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0)
sd = synth.make_state_dict("trained_like", seed=7)
torch.set_num_threads(min(16, os.cpu_count() or 1))
o32 = OracleFCNResNet50(); o32.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
o64 = OracleFCNResNet50(); o64.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); o64 = o64.double()
models = {m: FCNResNet50(m).load_state_dict(sd).to(dev) for m in ("f16x2", "fp32")}
rows = []
tot = {m: dict(flips=0, hip_right=0, cpu_right=0, worst_err=0.0, worst_margin=0.0) for m in models}
cpu_vs_64 = 0
for i in range(n):
    x = torch.from_numpy(synth.make_input(200 + i, 1024, 1024))[None]
    lab32, _, log32, _ = predict_labels(o32, x)
    lab64, _, log64, _ = predict_labels(o64, x.double())
    top2 = torch.topk(log64, 2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    cpu_err = float((log32.double() - log64).abs().max())
    cpu_vs_64 += int((lab32 != lab64).sum())
    row = {"frame": 200 + i, "logit_range": float(log64.abs().max()), "cpu_f32_err_vs_f64": cpu_err,
           "labels_cpu_f32_vs_f64_differ": int((lab32 != lab64).sum())}
    for m, model in models.items():
        labels, _ = model.predict_labels(x.to(dev))
        logits = model(x.to(dev))
        torch.cuda.synchronize()
        labels = labels.cpu()
        err = float((logits.cpu().double() - log64).abs().max())
        flip = labels != lab32
        k = int(flip.sum())
        hr = int((labels[flip] == lab64[flip]).sum()); cr = int((lab32[flip] == lab64[flip]).sum())
        mm = float(margin[flip].max()) if k else 0.0
        row[m] = {"err_vs_f64": err, "labels_vs_cpu_oracle_differ": k, "f64_agrees_with_hip": hr, "f64_agrees_with_cpu": cr,
                  "max_f64_margin_at_flip": mm, "labels_vs_f64_differ": int((labels != lab64).sum())}
        t = tot[m]; t["flips"] += k; t["hip_right"] += hr; t["cpu_right"] += cr
        t["worst_err"] = max(t["worst_err"], err); t["worst_margin"] = max(t["worst_margin"], mm)
    rows.append(row)
    print("frame %d: range %.3f | CPU f32 err %.2e | " % (200 + i, row["logit_range"], cpu_err) +
          " | ".join("%s err %.2e, %d flips vs CPU (f64 with HIP %d / CPU %d, margin <= %.1e)" %
                     (m, row[m]["err_vs_f64"], row[m]["labels_vs_cpu_oracle_differ"], row[m]["f64_agrees_with_hip"],
                      row[m]["f64_agrees_with_cpu"], row[m]["max_f64_margin_at_flip"]) for m in models), flush=True)
print("over %d frames (%d pixels): CPU f32 oracle vs float64 labels differ at %d pixels" % (n, n * 1024 * 1024, cpu_vs_64))
for m, t in tot.items():
    print("  %s: %d labels differ from the CPU oracle (float64 sides with HIP at %d, with the CPU at %d; largest float64 margin %.2e); "
          "worst logit error vs float64 %.2e" % (m, t["flips"], t["hip_right"], t["cpu_right"], t["worst_margin"], t["worst_err"]))
if out_path:
    json.dump({"frames": rows, "totals": tot}, open(out_path, "w"), indent=1)
