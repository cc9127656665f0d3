import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, ctypes as C
from neuralbarkcalculator_amd import synth, _lib
from neuralbarkcalculator_amd.model import FCNResNet50
from oracle.fcn_resnet50_oracle import OracleFCNResNet50, predict_labels
sd = synth.make_state_dict("trained_like", seed=7)
torch.set_num_threads(16)
o64 = OracleFCNResNet50(); o64.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); o64 = o64.double()
o32 = OracleFCNResNet50(); o32.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
xs = [torch.from_numpy(synth.make_input(i, 1024, 1024))[None] for i in (3, 201)]
refs = [predict_labels(o64, x.double())[2] for x in xs]
r32 = [predict_labels(o32, x)[2] for x in xs]
print("cpu f32 oracle:", ["%.2e" % float((a.double() - b).abs().max()) for a, b in zip(r32, refs)])
def model_on(path, prec):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, argtypes
    keep = _lib._lib; _lib._lib = lib
    try: m = FCNResNet50(prec).load_state_dict(sd).to("cuda:0")
    finally: _lib._lib = keep if keep is not None else lib
    return m
for path in sys.argv[1:]:
    for prec in (("f16x2", "fp32") if path == sys.argv[1] else ("f16x2",)):
        m = model_on(path, prec)
        errs = []
        for x, ref in zip(xs, refs):
            y = m(x.to("cuda:0")); torch.cuda.synchronize()
            errs.append(float((y.cpu().double() - ref).abs().max()))
        print(os.path.basename(path), prec, ["%.2e" % e for e in errs])
