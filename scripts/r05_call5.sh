#!/bin/bash
mkdir -p gpurun_out
for s in 1 2 3; do
  echo "streams $s" >> gpurun_out/r05_streams.log
  timeout -k 10 200 python scripts/ab_forward.py --libs neuralbarkcalculator_amd/libnbc_hip.so --streams $s --rounds 3 2>&1 | tail -1 >> gpurun_out/r05_streams.log
done
cat gpurun_out/r05_streams.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m perf -x -q -k "default_tiles" > gpurun_out/r05_perf_tests.log 2>&1; tail -5 gpurun_out/r05_perf_tests.log
