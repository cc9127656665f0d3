#!/bin/bash
# kernel trace of the 4-stream default bench, per-op durations under overlap.  gpurun -- 'bash scripts/profile_trace4.sh'
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/trace4
rm -rf $out; mkdir -p $out
python3 $root/bench.py --no-cpu-baseline --no-parity --streams 4 --steps 20 --warmup 3 --keep-tiles --dump-ops $out/ops.json > $out/ops_run.json 2> $out/ops_run.err || true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 $root/bench.py --no-cpu-baseline --no-parity --no-op-events --streams 4 --steps 64 --warmup 8 > $out/run.log 2>&1
cd $root
python3 scripts/trace_ops.py $(find $out/t -name "*kernel_trace.csv" | head -1) $out/ops.json
