#!/bin/bash
# One-line summaries of several bench.py configurations on ONE box (A/B runs must share a box: +-4 % between boxes).
#   gpurun -- 'bash scripts/bench_matrix.sh "<flags 1>" "<flags 2>" ...'   ->  gpurun_out/matrix/
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/matrix; mkdir -p $out
i=0
for flags in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python $root/bench.py --no-cpu-baseline --no-parity $flags > $out/run$i.json 2> $out/run$i.err || { echo "run $i failed"; tail -3 $out/run$i.err; continue; }
  python3 - "$out/run$i.json" "$flags" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d.get("roofline",{}); b=d.get("bf16_batch8")
print("%-50s %8.1f img/s  %6.2f ms/step  frac %.3f  3x3 %.1f TF" % (sys.argv[2], d["value"], d["ms_per_step"], r.get("frac",0), r.get("conv3x3_tflops",0)),
      ("| b8 %.1f img/s frac %.3f" % (b["value"], b.get("roofline",{}).get("frac",0))) if b else "")
PY
done
