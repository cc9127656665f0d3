source scripts/bench_matrix.sh true
run bf16_at_s3 --steps 60 --warmup 6 --streams 3
run bf16_t3_s3 --steps 60 --warmup 6 --streams 3 --conv-tile 3
run bf16_at_s4 --steps 60 --warmup 8 --streams 4
run bf16_at_s2 --steps 60 --warmup 8 --streams 2
run bf16_b2_s3 --steps 30 --warmup 6 --streams 3 --batch 2
python - <<'PY'
import json
for t in ['bf16_at_s3']:
    print(t, json.load(open(f'gpurun_out/bench_{t}.json'))['config']['autotuned_tiles'])
PY
