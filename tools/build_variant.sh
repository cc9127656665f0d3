#!/bin/bash
# An experimental build of the library for A/B runs (scripts/ab_tiles.py): tools/build_variant.sh NAME "-DFLAG ..."
#   -> tools/_bin/libnbc_NAME.so (git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; srcdir=${3:-neuralbarkcalculator_amd/csrc}     # third argument: another source directory (e.g. an older revision)
obj=tools/_bin/obj_$name; mkdir -p $obj
for src in nbc_net.cpp conv_igemm_dma.hip conv3x3_rows.hip pointwise.hip small_zones.hip nbc_api.hip; do
  [ -f $srcdir/$src ] || continue          # (an older revision has no conv3x3_rows.hip / conv1x1_stream.hip)
  extra="-ffp-contract=off"; case $src in *.hip) extra="-x hip";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags $extra -Iinclude -c $srcdir/$src -o $obj/$src.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_bin/libnbc_$name.so $obj/*.o
echo tools/_bin/libnbc_$name.so
