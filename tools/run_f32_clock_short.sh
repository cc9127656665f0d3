#!/bin/bash
T=tools/_bin/conv_timeline
run() { timeout -k 5 90 $T "$@" | grep -E "^shape|K loop|per wave" || exit 1; }
run 128 128 2048 512 3 1 0 2 1 0
run 128 128 2048 512 3 1 0 5 1 0
run 128 128 2048 512 3 1 0 9 1 0
run 128 128 2048 512 3 1 0 4 1 0
run 128 128 512 2048 1 1 1 2 1 0
run 128 128 256 256 3 2 0 9 1 0
