"""One-off parity check on a 2048x1536 frame (3x the pixels of the bench frame) against the CPU oracle, both modes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from neuralbarkcalculator_amd import synth
from neuralbarkcalculator_amd.model import FCNResNet50
from oracle.fcn_resnet50_oracle import OracleFCNResNet50, predict_labels
torch.set_num_threads(16)
sd = synth.make_state_dict("trained_like", seed=7)
om = OracleFCNResNet50(); om.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); om.eval()
h, w = 2048, 1536
x = torch.from_numpy(synth.make_input(77, h, w))[None]
t0 = time.time(); labels_ref, counts_ref, logits_ref, lowres_ref = predict_labels(om, x); print("oracle s", time.time() - t0)
for prec, rtol, min_agree in (("fp32", 2e-5, 0.9999), ("bf16", 4e-2, 0.98)):
    m = FCNResNet50(prec).load_state_dict(sd).to("cuda:0")
    labels, counts, lowres = m.predict_labels(x.to("cuda:0"), return_lowres=True)
    err = float((lowres.cpu() - lowres_ref).abs().max()); scale = float(lowres_ref.abs().max())
    agree = float((labels.cpu() == labels_ref).float().mean())
    print(prec, "lowres rel err %.2e" % (err / scale), "label agreement %.6f" % agree, "counts ok", int(counts.sum()) == h * w)
    assert err <= rtol * scale and agree >= min_agree
    del m
print("big image ok")
