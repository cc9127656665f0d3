source scripts/bench_matrix.sh true
run bf16_auto --steps 50 --warmup 5
NBC_CONV_VARIANT=1 run bf16_auto_prio --steps 50 --warmup 5
run bf16_t1 --steps 30 --warmup 3 --conv-tile 1
run bf16_t2 --steps 30 --warmup 3 --conv-tile 2
run fp32_auto --steps 20 --warmup 3 --precision fp32
NBC_CONV_VARIANT=1 run fp32_auto_prio --steps 20 --warmup 3 --precision fp32
run bf16_b8 --steps 10 --warmup 2 --batch 8
NBC_CONV_VARIANT=1 run bf16_b8_prio --steps 10 --warmup 2 --batch 8
