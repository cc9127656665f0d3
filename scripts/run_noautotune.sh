#!/bin/bash
source scripts/bench_matrix.sh true
run lat_tuned --steps 40 --warmup 5 --streams 1
run lat_default --steps 40 --warmup 5 --streams 1 --no-autotune
run s4_default --steps 80 --warmup 8 --streams 4 --no-autotune --no-op-events
python3 - <<PY
import json
a=json.load(open("gpurun_out/ops_lat_tuned.json")); b=json.load(open("gpurun_out/ops_lat_default.json"))
for x,y in zip(a,b):
    if y["ms"]>1.15*x["ms"] and y["ms"]-x["ms"]>0.002: print("  %-34s tuned %6.1f us  default %6.1f us"%(x["name"],x["ms"]*1e3,y["ms"]*1e3))
PY
