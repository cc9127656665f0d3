#!/bin/bash
# Copy what scripts/round_evidence.sh left under gpurun_out/ into profiles/ (tracked), named for the round.
#   bash scripts/collect_evidence.sh r02
set -e
r=${1:-r02}
E=gpurun_out/evidence; P=gpurun_out/profile_round
cp $E/gpu_tests.log profiles/${r}_gpu_tests.log
cp $E/smoke.log profiles/${r}_smoke.log
cp $E/bench_default.json profiles/${r}_bench_default.json
cp $E/bench_2ranks_one_gpu_gloo.json profiles/${r}_bench_2ranks_one_gpu_gloo.json
cp $E/time_folder.log profiles/${r}_time_folder.log
cp gpurun_out/fp64_adjudication.json profiles/${r}_fp64_adjudication.json
for c in f32_b1 bf16_b8; do
  cp $P/per_forward_ops_$c.json profiles/per_forward_ops_$c.json          # what bench.py reads for roofline.traffic
  cp $P/rocprof_kernel_stats_$c.csv profiles/${r}_rocprof_kernel_stats_$c.csv
  cp $P/bench_under_rocprof_$c.json profiles/${r}_bench_under_rocprof_$c.json
done
cp $P/per_forward_ops_f32_b1.json profiles/${r}_per_forward_ops_f32.json
cp $P/per_forward_ops_bf16_b8.json profiles/${r}_per_forward_ops_bf16.json
