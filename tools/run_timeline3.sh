#!/bin/bash
T=tools/_bin/conv_timeline
run() { timeout -k 5 60 $T "$@" | head -1 || exit 1; }
echo "== layer1 conv2";  for t in 7 10 9 11; do run 256 256 64 64 3 1 0 $t; done
echo "== layer1 conv3";  for t in 8 9 10 11; do run 256 256 64 256 1 1 1 $t; done
echo "== layer1 conv1";  for t in 7 10 11; do run 256 256 256 64 1 1 0 $t; done
echo "== layer2 conv2";  for t in 8 9 10; do run 128 128 128 128 3 1 0 $t; done
echo "== layer2 conv3";  for t in 1 9 10 11; do run 128 128 128 512 1 1 1 $t; done
echo "== layer2 conv1";  for t in 7 9 10 11; do run 128 128 512 128 1 1 0 $t; done
echo "== layer3 conv1";  for t in 8 9 10 11; do run 128 128 1024 256 1 1 0 $t; done
echo "== layer3 conv2";  for t in 8 9 10 11; do run 128 128 256 256 3 2 0 $t; done
echo "== layer3 conv3";  for t in 3 1 9 10 11; do run 128 128 256 1024 1 1 1 $t; done
echo "== layer4 conv1";  for t in 5 9 11; do run 128 128 2048 512 1 1 0 $t; done
echo "== layer4 conv3";  for t in 3 9 11; do run 128 128 512 2048 1 1 1 $t; done
