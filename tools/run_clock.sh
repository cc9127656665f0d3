#!/bin/bash
T=tools/_bin/conv_timeline
run() { timeout -k 5 60 $T "$@" | grep -E "^shape|K loop|per wave" || exit 1; }
run 128 128 2048 512 3 1 0 5
run 128 128 2048 512 3 1 0 2
run 128 128 512 512 3 4 0 5
run 128 128 2048 512 1 1 0 5
run 128 128 1024 2048 1 1 0 3
run 128 128 512 2048 1 1 1 3
run 128 128 256 256 3 2 0 8
