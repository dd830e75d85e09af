#!/bin/bash
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py "$@" > $OUT/bench_trace.log 2>&1 || { tail -5 $OUT/bench_trace.log; exit 1; }
python3 $ROOT/tools/kstats.py $OUT/trace 8
