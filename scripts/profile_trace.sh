#!/bin/bash
# kernel-trace + stats of the event-free bench loop.  Usage: profile_trace.sh <tag> [bench args]
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/trace_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py --no-cpu-baseline --no-parity --no-op-events "$@" > $out/run.log 2>&1
cd $root
python3 scripts/trace_gaps.py $(find $out -name "*kernel_trace.csv" | head -1)
