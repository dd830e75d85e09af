#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + stats, then one PMC pass per counter.
# usage: bash tools/profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r1}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-budget 0 --only --no-pure-gather $@"   # e.g. "--steps 50 --warmup 5", "--workload hbm --steps 24 --warmup 8", "--eval-only lfm2b"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -20
