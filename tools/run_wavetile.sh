#!/bin/bash
T=tools/_bin/conv_timeline
run() { timeout -k 5 60 $T "$@" | grep -E "^shape|K loop|per wave" | cut -c1-260 || exit 1; }
run 128 256 2048 512 3 1 0 5
run 128 256 2048 512 3 1 0 3
run 128 256 2048 512 3 1 0 2
run 128 256 512 512 3 4 0 5
run 128 256 512 512 3 4 0 3
