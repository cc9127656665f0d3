source scripts/bench_matrix.sh true
run bf16_s4_full --steps 80 --warmup 8 --streams 4 --no-op-events
NBC_CONV_ABLATE=1 run bf16_s4_nomfma --steps 80 --warmup 8 --streams 4 --no-op-events
NBC_CONV_ABLATE=2 run bf16_s4_nodma --steps 80 --warmup 8 --streams 4 --no-op-events
run bf16_b8s2_full --steps 12 --warmup 2 --streams 2 --batch 8 --no-op-events
NBC_CONV_ABLATE=1 run bf16_b8s2_nomfma --steps 12 --warmup 2 --streams 2 --batch 8 --no-op-events
NBC_CONV_ABLATE=2 run bf16_b8s2_nodma --steps 12 --warmup 2 --streams 2 --batch 8 --no-op-events
