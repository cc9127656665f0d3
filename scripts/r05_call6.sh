#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_resident or full1024 or tile_shape" > gpurun_out/r05_rowstep_tests.log 2>&1
rc=$?; echo "tests rc $rc"; tail -12 gpurun_out/r05_rowstep_tests.log
if [ $rc -eq 0 ]; then
  timeout -k 10 300 python scripts/ab_tiles.py --libs tools/_bin/libnbc_rows.so neuralbarkcalculator_amd/libnbc_hip.so --tiles=-1 --rounds 3 --layers conv2 > gpurun_out/r05_rowstep_ab.log 2>&1
  grep -E "tile|layer1|layer2" gpurun_out/r05_rowstep_ab.log
  timeout -k 10 300 python scripts/ab_forward.py --libs tools/_bin/libnbc_rows.so neuralbarkcalculator_amd/libnbc_hip.so > gpurun_out/r05_rowstep_forward.log 2>&1; tail -3 gpurun_out/r05_rowstep_forward.log
fi
