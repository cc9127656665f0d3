#!/bin/bash
# End-of-round evidence on one GPU box: GPU tests, smoke, the default bench line, the folder timing,
# then scripts/profile_round.sh (rocprofv3 kernel trace + PMC passes -> per-forward tables).
#   gpurun --timeout 1150 -- 'bash scripts/round_evidence.sh'   ->  gpurun_out/evidence/ and gpurun_out/profile_round/
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/evidence
rm -rf $out; mkdir -p $out
cd $root
timeout -k 10 500 python -m pytest tests -m gpu -q -s > $out/gpu_tests.log 2>&1 || { tail -20 $out/gpu_tests.log; exit 1; }
tail -1 $out/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -5 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -5 $out/bench_default.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --steps 20 --warmup 3 --no-bf16-leg > $out/bench_2ranks_one_gpu_gloo.json 2> $out/bench_2ranks.err || { tail -5 $out/bench_2ranks.err; exit 1; }
timeout -k 10 300 python scripts/time_folder.py 1000 bf16,fp32 > $out/time_folder.log 2>&1 || { tail -5 $out/time_folder.log; exit 1; }
grep -E "run 1" $out/time_folder.log
python3 - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print("f32 b1: %.1f img/s frac %.3f | parity %s" % (d["value"], d["roofline"]["frac"], d["parity"]["label_mismatches"]))
b=d["bf16_batch8"]; print("bf16 b8: %.1f img/s frac %.3f | match %.5f" % (b["value"], b["roofline"]["frac"], b["parity"]["label_match"]))
print("cpu baseline %.3f img/s on %d cores" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
PY
timeout -k 10 600 bash scripts/profile_round.sh > $out/profile_round.log 2>&1 || { tail -5 $out/profile_round.log; exit 1; }
echo evidence done
