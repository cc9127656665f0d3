#!/bin/bash
# Stream-count and forced-tile sweep of the default workload.
source scripts/bench_matrix.sh true
for s in 2 3 4 6 8; do run bf16_s$s --steps 96 --warmup 8 --streams $s; done
run bf16_s4_t9 --steps 96 --warmup 8 --streams 4 --conv-tile 9
run bf16_s4_t1 --steps 96 --warmup 8 --streams 4 --conv-tile 1
run bf16_s4_t5 --steps 96 --warmup 8 --streams 4 --conv-tile 5
run bf16_s4_b2 --steps 48 --warmup 4 --streams 4 --batch 2
run bf16_s2_b4 --steps 24 --warmup 4 --streams 2 --batch 4
