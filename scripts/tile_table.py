#!/usr/bin/env python3
"""Print a tile_model_probe.py result as a per-layer table: us per launch for every tile, the best tile, the default.
  python scripts/tile_table.py gpurun_out/tiles.json [case]"""
import json
import sys

data = json.load(open(sys.argv[1]))
case = sys.argv[2] if len(sys.argv) > 2 else sorted(data)[0]
d = data[case]
names, per = d["names"], d["per_tile"]
tiles = sorted(int(t) for t in per)
print(case)
print("%-32s" % "layer" + " ".join("%6d" % t for t in tiles) + "   best      default")
best_tot = 0.0
for i, n in enumerate(names):
    row = [per[str(t)][i] * 1e3 for t in tiles]
    b = min(row)
    best_tot += b
    print("%-32s" % n.replace("backbone.", "") + " ".join("%6.1f" % v for v in row) +
          "  %6.1f t%-2d  t%-2d %6.1f" % (b, tiles[row.index(b)], d["default_tiles"][i], d["default_ms"][i] * 1e3))
print("conv launches per forward: default tiles %.1f us, per-layer best %.1f us" % (sum(d["default_ms"]) * 1e3, best_tot))
print("autotune picked", d["tuned_tiles"])
