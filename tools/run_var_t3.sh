#!/bin/bash
T=tools/_bin/conv_timeline
run() { timeout -k 5 60 $T "$@" | grep -E "^shape|K loop" | cut -c1-230 || exit 1; }
for v in 0 1 0 1; do
  echo "== NBC_CONV_MFMA32=$v"
  NBC_CONV_MFMA32=$v run 128 256 2048 512 3 1 0 3
  NBC_CONV_MFMA32=$v run 128 256 512 512 3 4 0 3
  NBC_CONV_MFMA32=$v run 128 256 2048 512 1 1 0 3
done
