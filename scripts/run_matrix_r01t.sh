source scripts/bench_matrix.sh true
run bf16_lat --steps 40 --warmup 5 --streams 1
NBC_CONV_MFMA32=1 run bf16_lat_m32 --steps 40 --warmup 5 --streams 1
run bf16_s4 --steps 80 --warmup 8 --streams 4
NBC_CONV_MFMA32=1 run bf16_s4_m32 --steps 80 --warmup 8 --streams 4
run bf16_b8 --steps 10 --warmup 2 --streams 1 --batch 8
NBC_CONV_MFMA32=1 run bf16_b8_m32 --steps 10 --warmup 2 --streams 1 --batch 8
