#!/bin/bash
# batch-8 bf16: head conv (compute-bound) next to layer4 conv3 + identity (memory-bound), tiles that can / cannot share a CU
P=tools/_bin/conv_pair
H="1024 128 2048 512 3 1 0"; C="1024 128 512 2048 1 1 1"
for ta in 3 1 9 0; do for tb in 3 12 1 9 7; do timeout -k 5 120 $P "$H $ta" "$C $tb" 20 || exit 1; done; done
echo "== layer4 conv2 next to layer4 conv1"
A="1024 128 512 512 3 4 0"; B="1024 128 2048 512 1 1 0"
for ta in 3 1 9; do for tb in 3 1 9; do timeout -k 5 120 $P "$A $ta" "$B $tb" 20 || exit 1; done; done
