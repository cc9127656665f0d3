source scripts/bench_matrix.sh true
run bf16_lat --steps 40 --warmup 5 --streams 1
run bf16_s4 --steps 80 --warmup 8 --streams 4
run bf16_b8 --steps 10 --warmup 2 --streams 1 --batch 8
python3 scripts/ops_report.py gpurun_out/ops_bf16_lat.json
python3 -c "
import json
for t in ['bf16_lat','bf16_s4','bf16_b8']: print(t, json.load(open(f'gpurun_out/bench_{t}.json'))['config']['autotuned_tiles'])"
