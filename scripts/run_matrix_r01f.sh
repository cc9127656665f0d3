source scripts/bench_matrix.sh true
run bf16_at --steps 50 --warmup 5
run bf16_at_s3 --steps 60 --warmup 6 --streams 3
run fp32_at --steps 20 --warmup 3 --precision fp32
run bf16_b8_at --steps 10 --warmup 2 --batch 8
