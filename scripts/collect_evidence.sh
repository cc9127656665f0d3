#!/bin/bash
# Copy what scripts/round_evidence.sh and scripts/profile_round.sh left under gpurun_out/ into profiles/ (tracked),
# named for the round.   bash scripts/collect_evidence.sh r03
set -e
r=${1:-r03}
E=gpurun_out/evidence; P=gpurun_out/profile_round
if [ -d $E ]; then
  cp $E/gpu_tests.log profiles/${r}_gpu_tests.log
  cp $E/smoke.log profiles/${r}_smoke.log
  cp $E/bench_default.json profiles/${r}_bench_default.json
  cp $E/bench_2ranks_one_gpu_gloo.json profiles/${r}_bench_2ranks_one_gpu_gloo.json
  cp $E/time_folder.log profiles/${r}_time_folder.log
  cp $E/time_folder_ragged.log profiles/${r}_time_folder_ragged.log
  cp $E/host_ceiling_f16x2.log profiles/${r}_host_ceiling_8ranks_f16x2.log
  cp $E/host_ceiling_bf16.log profiles/${r}_host_ceiling_8ranks_bf16.log
  for m in fp32 f16x2; do cp gpurun_out/fp64_adjudication_$m.json profiles/${r}_fp64_adjudication_$m.json; done
fi
if [ -d $P ]; then
  for c in f16x2_b1 f32_b1 bf16_b8; do
    cp $P/per_forward_ops_$c.json profiles/per_forward_ops_$c.json          # what bench.py reads for roofline.traffic
    cp $P/per_forward_ops_$c.json profiles/${r}_per_forward_ops_$c.json
    cp $P/rocprof_kernel_stats_$c.csv profiles/${r}_rocprof_kernel_stats_$c.csv
    cp $P/bench_under_rocprof_$c.json profiles/${r}_bench_under_rocprof_$c.json
  done
fi
