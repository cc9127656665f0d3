#!/bin/bash
source scripts/bench_matrix.sh true
for k in 1 2; do
  unset NBC_CONV_MFMA32
  run s4_var0_$k --steps 96 --warmup 8 --streams 4 --no-op-events
  export NBC_CONV_MFMA32=1
  run s4_var1_$k --steps 96 --warmup 8 --streams 4 --no-op-events
done
unset NBC_CONV_MFMA32
run lat_var0 --steps 40 --warmup 5 --streams 1 --no-op-events
export NBC_CONV_MFMA32=1
run lat_var1 --steps 40 --warmup 5 --streams 1 --no-op-events
