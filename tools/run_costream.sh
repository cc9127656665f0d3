#!/bin/bash
T=tools/_bin/conv_timeline
run() { timeout -k 5 60 $T "$@" | grep -E "^shape|streams:" | cut -c1-200 || exit 1; }
echo "== head"; for t in 5 2 1 4 9 0; do run 128 128 2048 512 3 1 0 $t 2; done
echo "== layer4 conv2"; for t in 5 1 9; do run 128 128 512 512 3 4 0 $t 2; done
echo "== layer4 conv3"; for t in 3 5 1 9; do run 128 128 512 2048 1 1 1 $t 2; done
echo "== layer3 conv2"; for t in 8 10 1; do run 128 128 256 256 3 2 0 $t 2; done
echo "== layer3 conv3"; for t in 9 1 3; do run 128 128 256 1024 1 1 1 $t 2; done
