#!/bin/bash
# round 5, first GPU call: the activation floor before / after, K-loop ablations of the f16x2 kernel in the network, GPU tests
mkdir -p gpurun_out
python scripts/act_floor_probe.py tools/_bin/libnbc_noactexp.so neuralbarkcalculator_amd/libnbc_hip.so > gpurun_out/r05_act_floor.log 2>&1
echo "act_floor rc $?"
python scripts/ab_tiles.py --libs neuralbarkcalculator_amd/libnbc_hip.so tools/_bin/libnbc_abl1.so tools/_bin/libnbc_abl2.so tools/_bin/libnbc_abl3.so tools/_bin/libnbc_abl4.so --tiles -1 --rounds 2 > gpurun_out/r05_ablations.log 2>&1
echo "ablations rc $?"
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_tests_1.log 2>&1
echo "gpu tests rc $?"
tail -5 gpurun_out/r05_gpu_tests_1.log
