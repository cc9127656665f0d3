#!/bin/bash
# round 5, third GPU call: the row-resident 3x3 kernel: parity first, then per-layer timings against the generic kernel
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_resident or full1024 or tile_shape" > gpurun_out/r05_rows_tests.log 2>&1
rc=$?
echo "rows tests rc $rc"
tail -15 gpurun_out/r05_rows_tests.log
if [ $rc -eq 0 ]; then
  timeout -k 10 300 python scripts/ab_tiles.py --libs tools/_bin/libnbc_base.so neuralbarkcalculator_amd/libnbc_hip.so --tiles=-1 --rounds 3 > gpurun_out/r05_rows_ab.log 2>&1
  echo "ab rc $?"
  grep -E "tile|conv2|classifier.0" gpurun_out/r05_rows_ab.log
fi
