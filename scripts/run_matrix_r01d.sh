source scripts/bench_matrix.sh true
run bf16_auto --steps 50 --warmup 5
run bf16_s2 --steps 60 --warmup 6 --streams 2
run bf16_s3 --steps 60 --warmup 6 --streams 3
run bf16_s4 --steps 60 --warmup 8 --streams 4
run fp32_auto --steps 20 --warmup 3 --precision fp32
run fp32_s2 --steps 20 --warmup 4 --precision fp32 --streams 2
run bf16_b8 --steps 10 --warmup 2 --batch 8
run bf16_b8_s2 --steps 12 --warmup 2 --batch 8 --streams 2
run bf16_b4_s2 --steps 12 --warmup 2 --batch 4 --streams 2
