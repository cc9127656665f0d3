#!/bin/bash
T=tools/_bin/conv_timeline
run() { timeout -k 5 90 $T "$@" | grep -E "^shape|first K-step|MFMAs done|stores issued|stores acked" | cut -c1-170 || exit 1; }
run 512 256 512 2048 1 1 0 3 1 1
run 512 256 512 2048 1 1 1 3 1 1
run 512 256 512 2048 1 1 0 9 1 1
run 512 256 512 2048 1 1 1 9 1 1
run 512 256 64 2048 1 1 0 3 1 1
run 512 256 64 2048 1 1 1 3 1 1
