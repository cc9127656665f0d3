#!/bin/bash
# Block timelines of f16x2 layers at batch 1 (128x128 feature maps): where a block's time goes per tile shape.
#   gpurun -- 'bash tools/run_f16x2_layers.sh'  ->  gpurun_out/f16x2_layers.log
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/f16x2_layers.log; : > $out
export NBC_WARM=${NBC_WARM:-300}
run() { timeout -k 5 60 $root/tools/_bin/conv_timeline "$@" >> $out 2>&1 || echo "failed: $*" >> $out; }
for tile in ${TILES:-17 14 8}; do
  run 128 128 1024 256 1 1 0 $tile 1 2      # layer3 conv1
  run 128 128 256 1024 1 1 1 $tile 1 2      # layer3 conv3 + identity
  run 128 128 256 256 3 2 0 $tile 1 2       # layer3 conv2 (d 2)
  run 128 128 512 2048 1 1 1 $tile 1 2      # layer4 conv3 + identity
  run 128 128 2048 512 1 1 0 $tile 1 2      # layer4 conv1
done
run 128 128 2048 512 3 1 0 17 1 2           # head conv
run 128 128 2048 512 3 1 0 5 1 2
run 256 256 64 64 3 1 0 10 1 2              # layer1 conv2
run 256 256 64 256 1 1 1 17 1 2             # layer1 conv3
