#!/bin/bash
# Example A/B matrix on one GPU box:  gpurun -- 'bash scripts/run_matrix.sh'
source scripts/bench_matrix.sh true
run bf16_lat --steps 40 --warmup 5 --streams 1
run bf16_s4 --steps 80 --warmup 8 --streams 4
run bf16_b8 --steps 10 --warmup 2 --streams 1 --batch 8
run fp32_lat --steps 20 --warmup 3 --streams 1 --precision fp32
python3 scripts/ops_report.py gpurun_out/ops_bf16_lat.json
