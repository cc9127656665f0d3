#!/bin/bash
# Diagnostic tools (not part of libnbc_hip.so).  Output: tools/_bin/ (git-ignored, travels with gpurun).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DNBC_STAMPS -DNBC_DIAG -Ineuralbarkcalculator_amd/csrc \
  tools/conv_timeline.hip neuralbarkcalculator_amd/csrc/conv_igemm_dma.hip neuralbarkcalculator_amd/csrc/conv3x3_rows.hip -o tools/_bin/conv_timeline
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_f32_probe.hip -o tools/_bin/mfma_f32_probe
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/split_probe.hip -o tools/_bin/split_probe
