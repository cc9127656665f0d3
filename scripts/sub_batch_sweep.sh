#!/bin/bash
# bf16 batch 8 with the tail of the plan run in sub-batches (nbc_set_sub_batch): first op x images per sub-batch x streams.
#   gpurun -- 'bash scripts/sub_batch_sweep.sh'   ->  gpurun_out/matrix/
root=${GRAFT_REPO_ROOT:-$(pwd)}
runs=()
for st in 1 2; do
  for sb in off backbone.layer3.0.conv1:1 backbone.layer3.0.conv1:2 backbone.layer3.0.conv1:4 backbone.layer4.0.conv1:1 \
            backbone.layer4.0.conv1:2 backbone.layer4.0.conv1:4 backbone.layer2.0.conv1:2 backbone.layer2.0.conv1:4 backbone.layer1.0.conv1:4; do
    runs+=("--precision bf16 --batch 8 --streams $st --no-bf16-leg --steps 30 --sub-batch $sb")
  done
done
bash $root/scripts/bench_matrix.sh "${runs[@]}"
