#!/bin/bash
source scripts/bench_matrix.sh true
for r in 8; do
  export NBC_UP_ROWS=$r
  run up$r --steps 40 --warmup 5 --streams 1
  python3 - <<PY
import json
for y in json.load(open("gpurun_out/ops_up$r.json")):
    if y["name"] in ("upsample_argmax","backbone.maxpool","ingest","classifier.4","backbone.conv1"): print("   ", y["name"], round(y["ms"]*1e3,1), "us")
PY
done
