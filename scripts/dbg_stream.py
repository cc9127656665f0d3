import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from neuralbarkcalculator_amd import synth, topology
from neuralbarkcalculator_amd.model import FCNResNet50
sd = synth.make_state_dict("trained_like", seed=7)
x = torch.from_numpy(np.stack([synth.make_input(40, 1024, 1024)])).to("cuda:0")
m = FCNResNet50("f16x2").load_state_dict(sd).to("cuda:0")
m.set_keep_activations(True)
names = [u.name for u in topology.conv_units() if u.bn is not None]
acts = {}
for tile in (17, 20):
    m.set_conv_tile(tile)
    m.lowres_logits(x); torch.cuda.synchronize()
    print("tile", tile, "plan", m.plan_tiles())
    acts[tile] = {}
    for n in names:
        u = [q for q in topology.conv_units() if q.name == n][0]
        acts[tile][n] = m.read_activation(n, 1 * u.cout * 512 * 512)
for n in names:
    a, b = acts[17][n], acts[20][n]
    if not np.array_equal(a, b):
        d = np.argwhere(a != b)
        print(n, a.shape, "differs at", len(d), "elements; channels", np.unique(d[:, 1])[:8], "rows", np.unique(d[:, 2])[:8], np.unique(d[:,2])[-4:], "cols", np.unique(d[:, 3])[:16], np.unique(d[:, 3])[-8:])
        print("  nan in 20:", int(np.isnan(b).sum()), " max abs diff", float(np.nanmax(np.abs(a - b))))
        break
else:
    print("all layers equal")
