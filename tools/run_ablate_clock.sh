#!/bin/bash
# Shader clock / MFMA-busy share / wait shares of the head conv under the timing-only ablations.
T=tools/_bin/conv_timeline
for a in 0 2 4 1 3; do
  echo "== NBC_CONV_ABLATE=$a (0 full, 2 no refill DMA, 4 MFMA on constants, 1 no MFMA, 3 DMA+barriers)"
  NBC_CONV_ABLATE=$a timeout -k 5 60 $T 128 128 2048 512 3 1 0 5 | grep -E "^shape|K loop|per wave" | cut -c1-250 || exit 1
done
