#!/bin/bash
# Batch-8 bf16 1x1 layers (layer4 conv3 + identity, layer4 conv1, layer3 conv3 + identity): tiles and ablations.
T=tools/_bin/conv_timeline
run() { timeout -k 5 90 $T "$@" | grep -E "^shape|first K-step|last K-step|MFMAs done|stores issued|stores acked|K loop|CUs used" | cut -c1-190 || exit 1; }
for tile in 3 12 5 2 1 9; do run 1024 128 512 2048 1 1 1 $tile; done
for ab in 1 2 3 4; do echo "== NBC_CONV_ABLATE=$ab (1 no MFMA, 2 no refill DMA, 3 DMA+barriers only, 4 MFMA only)"; NBC_CONV_ABLATE=$ab run 1024 128 512 2048 1 1 1 3; NBC_CONV_ABLATE=$ab run 1024 128 512 2048 1 1 1 5; done
echo "== without the identity"
run 1024 128 512 2048 1 1 0 3
echo "== layer4 conv1"
for tile in 3 12 5; do run 1024 128 2048 512 1 1 0 $tile; done
echo "== layer3 conv3"
for tile in 3 12 5 1; do run 1024 128 256 1024 1 1 1 $tile; done
