source scripts/bench_matrix.sh true
run bf16_auto --steps 50 --warmup 5
run bf16_t0 --steps 30 --warmup 3 --conv-tile 0
run bf16_t1 --steps 30 --warmup 3 --conv-tile 1
run bf16_t2 --steps 30 --warmup 3 --conv-tile 2
run bf16_t3 --steps 30 --warmup 3 --conv-tile 3
run bf16_v1 --steps 30 --warmup 3 --conv-impl 0
run fp32_auto --steps 20 --warmup 3 --precision fp32
run fp32_v1 --steps 20 --warmup 3 --precision fp32 --conv-impl 0
run bf16_b8 --steps 10 --warmup 2 --batch 8
run bf16_b8_t3 --steps 10 --warmup 2 --batch 8 --conv-tile 3
