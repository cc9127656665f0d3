#!/bin/bash
# round 5, second GPU call: what would less pixel-row traffic buy (abl5: rows fetched for one tap column in three; abl6: none), four ring slots on tile 14
mkdir -p gpurun_out
python scripts/ab_tiles.py --libs tools/_bin/libnbc_base.so tools/_bin/libnbc_abl5.so tools/_bin/libnbc_abl6.so tools/_bin/libnbc_t14s4.so --tiles=-1,14,5 --rounds 2 > gpurun_out/r05_ablations2.log 2>&1
echo "ablations2 rc $?"
