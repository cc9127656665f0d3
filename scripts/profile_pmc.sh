#!/bin/bash
# rocprofv3 passes for the bench: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate
# PMC passes (MI355X_MICROARCH.md: they do not fit one pass).  Usage: profile_pmc.sh <tag> [bench args]
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --no-cpu-baseline --no-parity --dump-ops $out/ops.json "$@" > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py --no-cpu-baseline --no-parity --no-op-events "$@" > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py --no-cpu-baseline --no-parity --no-op-events "$@" > $out/write.log 2>&1
find $out -name "*.csv" | head -20
