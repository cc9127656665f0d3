#!/bin/bash
# End-of-round evidence on one GPU box: tests, bench lines, rocprofv3 kernel stats and PMC traffic.
#   gpurun --timeout 1100 -- 'bash scripts/refresh_profiles.sh'   ->  gpurun_out/refresh/
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/refresh
rm -rf $out; mkdir -p $out
cd $root
timeout -k 10 400 python -m pytest tests -m gpu -q > $out/gpu_tests.log 2>&1 || { tail -20 $out/gpu_tests.log; exit 1; }
tail -1 $out/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -5 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout -k 10 300 python bench.py --batch 8 --streams 2 --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-leg > $out/bench_batch8.json 2> $out/bench_batch8.err
timeout -k 10 300 python bench.py --precision fp32 --streams 1 --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_fp32.json 2> $out/bench_fp32.err
echo benches done
for prec in bf16 fp32; do
  bash scripts/profile_pmc.sh $prec --precision $prec --streams 1 --steps 5 --warmup 2 > $out/pmc_$prec.log 2>&1
  python3 scripts/pmc_summary.py $root/gpurun_out/prof_$prec $root/gpurun_out/prof_$prec/ops.json $out/pmc_traffic_$prec.json >> $out/pmc_$prec.log 2>&1
  cp $(find $root/gpurun_out/prof_$prec/stats -name "*kernel_stats.csv" | head -1) $out/rocprof_kernel_stats_${prec}_streams1.csv
done
echo profiles done
python3 - <<PY
import json
for n in ("default","batch8","fp32"):
    d=json.load(open("$out/bench_%s.json"%n))
    print(n, "img/s %.1f"%d["value"], "roofline", {k:d["roofline"].get(k) for k in ("achieved","frac","traffic")}, d.get("single_stream",{}).get("value"), d.get("cpu_baseline",{}).get("value"), d.get("parity",{}))
PY
