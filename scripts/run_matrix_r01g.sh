source scripts/bench_matrix.sh true
run bf16_at_s3 --steps 60 --warmup 6 --streams 3
run bf16_t3_s3 --steps 60 --warmup 6 --streams 3 --conv-tile 3
run bf16_t3_s4 --steps 60 --warmup 8 --streams 4 --conv-tile 3
run bf16_t2_s3 --steps 60 --warmup 6 --streams 3 --conv-tile 2
run bf16_t5_s3 --steps 60 --warmup 6 --streams 3 --conv-tile 5
run bf16_b2_s3 --steps 30 --warmup 6 --streams 3 --batch 2
run bf16_b2_t3_s3 --steps 30 --warmup 6 --streams 3 --batch 2 --conv-tile 3
