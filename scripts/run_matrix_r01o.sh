source scripts/bench_matrix.sh true
run bf16_s4 --steps 80 --warmup 8 --streams 4
run bf16_lat --steps 40 --warmup 5 --streams 1
run bf16_t3 --steps 20 --warmup 3 --streams 1 --conv-tile 3
run bf16_b8 --steps 10 --warmup 2 --streams 1 --batch 8
run bf16_b8s2 --steps 12 --warmup 2 --streams 2 --batch 8
