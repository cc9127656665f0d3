#!/bin/bash
# f32 parity mode: shader clock and MFMA-busy share of the K loop on the layers that carry the FLOPs
T=tools/_bin/conv_timeline
run() { timeout -k 5 90 $T "$@" | grep -E "^shape|K loop|per wave" || exit 1; }
run 128 128 2048 512 3 1 0 2 1 0
run 128 128 2048 512 3 1 0 5 1 0
run 128 128 2048 512 3 1 0 9 1 0
run 128 128 2048 512 3 1 0 1 1 0
run 128 128 2048 512 3 1 0 4 1 0
run 128 128 512 512 3 4 0 2 1 0
run 128 128 512 512 3 4 0 9 1 0
run 128 128 1024 2048 1 1 0 2 1 0
run 128 128 512 2048 1 1 1 2 1 0
run 128 128 256 1024 1 1 1 9 1 0
run 128 128 256 256 3 2 0 9 1 0
run 128 128 256 256 3 2 0 10 1 0
run 256 256 64 64 3 1 0 10 1 0
run 256 256 64 64 3 1 0 0 1 0
