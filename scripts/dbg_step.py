import sys; sys.path.insert(0, '.')
import numpy as np, torch
from neuralbarkcalculator_amd import synth
from neuralbarkcalculator_amd.model import FCNResNet50
from torch.profiler import profile, ProfilerActivity
sd = synth.make_state_dict("trained_like", seed=7)
m = FCNResNet50("bf16").load_state_dict(sd).to("cuda:0")
b = np.stack([synth.make_input(0, 256, 256)])
x = torch.from_numpy(b).to("cuda:0")
print(x.is_contiguous(), x.stride())
m.predict_labels(x, labels_dtype=torch.uint8); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    m.predict_labels(x, labels_dtype=torch.uint8); torch.cuda.synchronize()
for e in prof.key_averages():
    if 'aten' in e.key or 'copy' in e.key.lower() or 'fill' in e.key.lower() or 'Memset' in e.key or 'Memcpy' in e.key:
        print(e.key, e.count)
