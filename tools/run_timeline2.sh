#!/bin/bash
T=tools/_bin/conv_timeline
run() { timeout -k 5 60 $T "$@" || exit 1; }
run 256 256 64 64 3 1 0 7
run 256 256 256 64 1 1 0 7
run 128 128 128 512 1 1 1 0
run 128 128 128 512 1 1 1 1
run 128 128 256 1024 1 1 1 1
run 128 128 256 256 3 2 0 8
run 128 128 512 2048 1 1 1 3
