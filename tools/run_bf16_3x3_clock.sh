#!/bin/bash
# bf16 3x3 layers: TFLOP/s next to the shader clock the chip holds in the K loop and the MFMA-busy share of
# its cycles (diagnostic build with in-kernel stamps; random data, back-to-back launches).
# Batch 1 shapes (M = 16384 / 65536) and batch 8 shapes (M x 8), on the tiles the autotuner picks.
T=tools/_bin/conv_timeline
run() { timeout -k 5 90 $T "$@" | grep -E "^shape|K loop" | cut -c1-200 || exit 1; }
echo "== batch 1"
run 256 256 64 64 3 1 0 10      # layer1 conv2
run 128 128 128 128 3 1 0 9     # layer2 conv2
run 128 128 256 256 3 2 0 9     # layer3 conv2 (dilation 2)
run 128 128 512 512 3 4 0 5     # layer4 conv2 (dilation 4)
run 128 128 512 512 3 4 0 2
run 128 128 2048 512 3 1 0 5    # classifier.0
run 128 128 2048 512 3 1 0 2
echo "== batch 8"
run 2048 256 64 64 3 1 0 10
run 1024 128 128 128 3 1 0 9
run 1024 128 256 256 3 2 0 3
run 1024 128 512 512 3 4 0 3
run 1024 128 2048 512 3 1 0 3
