#!/bin/bash
# rocprofv3 evidence for the two bench configurations, on one GPU box:
#   gpurun --timeout 1100 -- 'bash scripts/profile_round.sh'   ->  gpurun_out/profile_round/
# Per configuration (f16x2 batch 1 = the headline; f32 MFMA batch 1; bf16 batch 8 = configs[2]):
#   1. a plain bench run measures the per-layer tiles once and saves them (tiles.json);
#   2. rocprofv3 --kernel-trace --stats of the SAME bench command with those tiles installed, so that the
#      trace and the --stats summary hold nothing but warm-up and timed forwards (no autotune launches);
#   3. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (they do not fit one pass: MI355X_MICROARCH.md), and
#      SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE in a third (matrix-pipe utilisation and shader clock per launch);
#   4. scripts/per_forward_table.py -> per_forward_ops_<dtype>_b<batch>.json (copy it into profiles/).
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profile_round
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
K=${K:-12}; W=${W:-3}
python3 $root/bench.py --frames 8 --streams 1 --steps $K --warmup $W --bf16-steps $K --bf16-streams 1 --f32-steps $K --f32-streams 1 --no-cpu-baseline --no-parity --save-tiles $out/tiles.json > $out/bench_tiles.json 2> $out/bench_tiles.err
cfgs=("f16x2 1 f16x2" "fp32 1 f32" "bf16 8 bf16")
for cfg in "${cfgs[@]}"; do
  set -- $cfg; prec=$1; batch=$2; dt=$3
  d=$out/${dt}_b${batch}; mkdir -p $d
  common="--frames 8 --precision $prec --batch $batch --streams 1 --steps $K --warmup $W --no-bf16-leg --no-f32-leg --no-cpu-baseline --no-parity --tiles-file $out/tiles.json"
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 $root/bench.py $common --dump-ops $d/ops.json > $d/trace.log 2>&1
  echo "trace $cfg done"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 $root/bench.py $common --no-op-events > $d/fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $d/write -- python3 $root/bench.py $common --no-op-events > $d/write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $d/mfma -- python3 $root/bench.py $common --no-op-events > $d/mfma.log 2>&1
  echo "pmc $cfg done"
  python3 $root/scripts/per_forward_table.py $d $out/per_forward_ops_${dt}_b${batch}.json --precision $prec --batch $batch --steps $K --warmup $W --no-autotune > $out/table_${dt}_b${batch}.log 2>&1 || { tail -5 $out/table_${dt}_b${batch}.log; exit 1; }
  cp $(find $d/trace -name "*kernel_stats.csv" | head -1) $out/rocprof_kernel_stats_${dt}_b${batch}.csv
  grep -h '"metric"' $d/trace.log > $out/bench_under_rocprof_${dt}_b${batch}.json || true
  rm -rf $d/fetch $d/write $d/mfma  # counter CSVs are large; the table keeps what is needed
  find $d/trace -name "*kernel_trace.csv" -exec gzip -9 {} \;
done
head -40 $out/table_f16x2_b1.log
