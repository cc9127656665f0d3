source scripts/bench_matrix.sh true
run bf16_noat --steps 50 --warmup 5 --no-autotune
run bf16_at --steps 50 --warmup 5
run bf16_at_s3 --steps 60 --warmup 6 --streams 3
run fp32_at --steps 20 --warmup 3 --precision fp32
run bf16_b8_at --steps 10 --warmup 2 --batch 8
run bf16_b8_at_s2 --steps 12 --warmup 2 --batch 8 --streams 2
python - <<'PY'
import json
for t in ['bf16_at','fp32_at','bf16_b8_at']:
    print(t, json.load(open(f'gpurun_out/bench_{t}.json'))['config']['autotuned_tiles'])
PY
