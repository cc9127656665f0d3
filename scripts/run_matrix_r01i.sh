source scripts/bench_matrix.sh true
run bf16_at_s4 --steps 80 --warmup 8 --streams 4 --no-op-events
run bf16_t1_s4 --steps 80 --warmup 8 --streams 4 --conv-tile 1 --no-op-events
run bf16_t0_s4 --steps 80 --warmup 8 --streams 4 --conv-tile 0 --no-op-events
run bf16_t3_s4 --steps 80 --warmup 8 --streams 4 --conv-tile 3 --no-op-events
run bf16_t3_s6 --steps 90 --warmup 12 --streams 6 --conv-tile 3 --no-op-events
run bf16_at_s6 --steps 90 --warmup 12 --streams 6 --no-op-events
run bf16_at_s8 --steps 96 --warmup 16 --streams 8 --no-op-events
