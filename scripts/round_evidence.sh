#!/bin/bash
# End-of-round evidence on one GPU box: GPU tests, smoke, the default bench line, the two-rank rehearsal, the folder
# timings, the host-side ceiling of eight ranks; scripts/profile_round.sh (rocprofv3 kernel trace + PMC passes ->
# per-forward tables) runs in its own call (the two together exceed one call's time limit).
#   gpurun --timeout 1150 -- 'bash scripts/round_evidence.sh'   ->  gpurun_out/evidence/
#   gpurun --timeout 1150 -- 'bash scripts/profile_round.sh'    ->  gpurun_out/profile_round/
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/evidence
rm -rf $out; mkdir -p $out
cd $root
timeout -k 10 600 python -m pytest tests -m gpu -q -s > $out/gpu_tests.log 2>&1 || { tail -20 $out/gpu_tests.log; exit 1; }
tail -1 $out/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -5 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -5 $out/bench_default.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --steps 20 --warmup 3 --no-bf16-leg > $out/bench_2ranks_one_gpu_gloo.json 2> $out/bench_2ranks.err || { tail -5 $out/bench_2ranks.err; exit 1; }
timeout -k 10 300 python scripts/time_folder.py 1000 f16x2,bf16,fp32 > $out/time_folder.log 2>&1 || { tail -5 $out/time_folder.log; exit 1; }
grep -E "run 1" $out/time_folder.log
timeout -k 10 200 python scripts/time_folder.py 1000 f16x2,fp32 ragged > $out/time_folder_ragged.log 2>&1 || { tail -5 $out/time_folder_ragged.log; exit 1; }
grep -E "run 1" $out/time_folder_ragged.log
timeout -k 10 200 python scripts/host_ceiling.py 8 16 125 290 2 > $out/host_ceiling_f16x2.log 2>&1 || { tail -5 $out/host_ceiling_f16x2.log; exit 1; }
timeout -k 10 200 python scripts/host_ceiling.py 8 16 125 840 8 > $out/host_ceiling_bf16.log 2>&1 || { tail -5 $out/host_ceiling_bf16.log; exit 1; }
tail -n 3 $out/host_ceiling_f16x2.log; tail -n 3 $out/host_ceiling_bf16.log
python3 - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print("f16x2 b1: %.1f img/s frac %.3f | label mismatches %s, logit err %.2e of range" % (d["value"], d["roofline"]["frac"], d["parity"]["label_mismatches"], d["parity"]["max_lowres_logit_err_over_oracle_range"]))
f=d["f32_mfma_batch1"]; print("f32 MFMA b1: %.1f img/s frac %.3f | label mismatches %s, logit err %.2e" % (f["value"], f["roofline"]["frac"], f["parity"]["label_mismatches"], f["parity"]["max_lowres_logit_err_over_oracle_range"]))
b=d["bf16_batch8"]; print("bf16 b8: %.1f img/s frac %.3f | match %.5f" % (b["value"], b["roofline"]["frac"], b["parity"]["label_match"]))
print("cpu baseline %.3f img/s on %d cores" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
PY
echo evidence done
