#!/bin/bash
# A/B matrix of bench.py configurations on one GPU box (writes gpurun_out/<tag>.json).
set -e
mkdir -p gpurun_out
run() { tag=$1; shift; timeout -k 10 240 python bench.py --no-cpu-baseline --no-parity "$@" --dump-ops gpurun_out/ops_$tag.json > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; python - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
r=d.get("roofline",{})
print("$tag", "img/s %.1f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "instr %.3f" % r.get("instrumented_ms_per_step",0), "conv TF %.0f" % r.get("achieved",0), "3x3 TF %.0f" % r.get("conv3x3_tflops",0), "sumk %.3f" % r.get("sum_kernel_ms_per_step",0))
PY
}
"$@"
