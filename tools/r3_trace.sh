#!/bin/bash
# kernel trace of one bench command: usage r3_trace.sh <tag> <bench args...>
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --cpu-budget 0 --only --no-pure-gather "$@" > $OUT/bench_trace.log 2>&1 || { tail -5 $OUT/bench_trace.log; exit 1; }
grep -o '"ms_per_step": [0-9.]*' $OUT/bench_trace.log
