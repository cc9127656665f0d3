source scripts/bench_matrix.sh true
run bf16_lat --steps 50 --warmup 5 --streams 1
run bf16_t9 --steps 30 --warmup 5 --streams 1 --conv-tile 9
run bf16_t5 --steps 30 --warmup 5 --streams 1 --conv-tile 5
run bf16_t8 --steps 30 --warmup 5 --streams 1 --conv-tile 8
run bf16_t3 --steps 30 --warmup 5 --streams 1 --conv-tile 3
run bf16_s4 --steps 80 --warmup 8 --streams 4
run bf16_b8 --steps 10 --warmup 2 --batch 8 --streams 1
run fp32_lat --steps 20 --warmup 3 --precision fp32 --streams 1
python - <<'PY'
import json
for t in ['bf16_lat','bf16_s4','bf16_b8','fp32_lat']:
    print(t, json.load(open(f'gpurun_out/bench_{t}.json'))['config']['autotuned_tiles'])
PY
