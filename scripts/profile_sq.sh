#!/bin/bash
# SQ/LDS/MFMA counters per dispatch (own PMC pass, kernel-trace only).  Usage: profile_sq.sh <tag> [bench args]
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/sq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/a -- python3 $root/bench.py --no-cpu-baseline --no-parity --no-op-events "$@" > $out/a.log 2>&1 || tail -5 $out/a.log
cd $root
find $out -name "*counter_collection.csv" | head
