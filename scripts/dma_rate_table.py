#!/usr/bin/env python3
"""Per conv launch of the f16x2 forward: bytes the block's LDS-DMAs move into LDS per CU, the rate that is over the launch's
duration, and the time a model 'MFMAs at peak + DMAs at the DMA-only rate, one after the other' predicts.
  python scripts/dma_rate_table.py [profiles/r05_per_forward_ops_f16x2_b1.json] [dma_only_GB_per_s_per_CU]
Reads only the committed per-forward table (scripts/per_forward_table.py); no GPU.  The DMA-only rate (89 GB/s per CU) is the head
conv's K loop with its MFMAs and fragment reads removed (profiles/r05_f16x2_kloop_ablations_in_network.log: 423 us)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_per_forward_ops_f16x2_b1.json")
dma_only = float(sys.argv[2]) if len(sys.argv) > 2 else 89.0
d = json.load(open(path))
peak = d["peak_tflops"] * 1e12
print("%-32s %-8s %5s %6s | %7s %7s %7s | %7s %6s | %6s" % ("launch", "tile", "grid", "K", "MiB/CU", "us", "GB/s/CU", "model", "ratio", "util"))
tot = [0.0, 0.0]
for o in d["ops"]:
    k = o["kernel"]
    m = re.match(r"conv_dma_kernel<2, (\d+), (\d+), (\d+), (\d+), (\d+), (\w+), (\d+), (\w+)>", k)
    r = re.match(r"conv3x3_rowstep_kernel<([\d, ]+)>", k)          # <WN, NT, SB[, OR]>: OR output rows per block
    if m and m.group(6) == "false":
        wm, wn, mt, nt = (int(m.group(i)) for i in range(1, 5))
        bm, bn, orows = wm * mt * 32, wn * nt * 32, 0
    elif r:
        a = [int(v) for v in r.group(1).split(",")]
        orows = a[3] if len(a) > 3 else 1
        bm, bn = 128 * orows, a[0] * 32 * a[1]
    else:
        continue
    co = o["cout"]
    pixels = o["grid"] * bm * bn / co
    K = o["flops"] / (2 * pixels * co)
    if orows == 0:
        per_block = K / 32 * (bm + bn) * 128                      # a K-step: BM pixel rows + BN weight rows of 128 bytes
    else:                                                         # a channel block: orows + 2 row slots of 144 pixels + nine taps' weight rows
        per_block = K / 32 / 9 * ((orows + 2) * 144 * 128 + 9 * bn * 128)
    per_cu = per_block * o["grid"] / 256
    model = o["flops"] / peak * 1e6 + per_cu / (dma_only * 1e3)
    tot[0] += o["median_us"]
    tot[1] += model
    print("%-32s %-8s %5d %6d | %7.2f %7.1f %7.1f | %7.1f %6.2f | %6.2f" % (
        o["op"].replace("backbone.", ""), "%dx%d" % (bm, bn), o["grid"], K, per_cu / 2**20, o["median_us"], per_cu / o["median_us"] / 1e3,
        model, o["median_us"] / model, o.get("mfma_util") or 0))
print("sum of these launches: measured %.0f us, model %.0f us" % tuple(tot))
