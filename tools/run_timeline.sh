#!/bin/bash
# Per-phase block timelines of representative layers (bf16, 1024x1024 input).  gpurun -- 'bash tools/run_timeline.sh'
T=tools/_bin/conv_timeline
run() { timeout -k 5 60 $T "$@" || exit 1; }
echo "== layer1 conv2 (3x3 64->64 @256x256)";   for t in 0 7 6; do run 256 256 64 64 3 1 0 $t; done
echo "== layer1 conv3 (1x1 64->256 +res)";      for t in 0 1 7 8 5; do run 256 256 64 256 1 1 1 $t; done
echo "== layer1 conv1 (1x1 256->64)";           for t in 0 7 6; do run 256 256 256 64 1 1 0 $t; done
echo "== layer2 conv2 (3x3 128->128 @128x128)"; for t in 0 1 7 8; do run 128 128 128 128 3 1 0 $t; done
echo "== layer2 conv3 (1x1 128->512 +res)";     for t in 0 1 7 8 5; do run 128 128 128 512 1 1 1 $t; done
echo "== layer3 conv1 (1x1 1024->256)";         for t in 0 1 7 8; do run 128 128 1024 256 1 1 0 $t; done
echo "== layer3 conv2 (3x3 256->256 d2)";       for t in 0 1 8 4; do run 128 128 256 256 3 2 0 $t; done
echo "== layer3 conv3 (1x1 256->1024 +res)";    for t in 1 5 3 2 0; do run 128 128 256 1024 1 1 1 $t; done
echo "== layer4 conv1 (1x1 2048->512)";         for t in 1 2 5 4; do run 128 128 2048 512 1 1 0 $t; done
echo "== layer4 conv2 (3x3 512->512 d4)";       for t in 1 2 5 3; do run 128 128 512 512 3 4 0 $t; done
echo "== layer4 conv3 (1x1 512->2048 +res)";    for t in 3 5 2 1; do run 128 128 512 2048 1 1 1 $t; done
echo "== head (3x3 2048->512)";                 for t in 5 3 2; do run 128 128 2048 512 3 1 0 $t; done
