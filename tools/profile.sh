#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + stats, then one PMC pass per counter group.
# usage: bash tools/profile.sh <tag> [bench args...]            FETCH_SIZE and WRITE_SIZE (HBM-side traffic per launch)
#        PMC_EXTRA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" bash tools/profile.sh <tag> ...
#                                                               + one more pass with those counters (matrix-pipe busy share)
# Counters are collected in runs of their own, with --kernel-trace / --stats nowhere near them (MI355X guide).
set -o pipefail
TAG=${1:-r1}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
case "$*" in
  *--eval-only*) ARGS="$@" ;;
  *) ARGS="--cpu-budget 0 --only --no-pure-gather $@" ;;   # e.g. "--steps 20 --warmup 5", "--workload hbm --steps 24 --warmup 8"
esac
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1 || exit 1
if [ -n "$PMC_EXTRA" ]; then
  rocprofv3 --pmc $PMC_EXTRA --output-format csv -d $OUT/pmc_extra -- python3 $ROOT/bench.py $ARGS > $OUT/bench_extra.log 2>&1 || exit 1
fi
python3 $ROOT/tools/pmc_summary.py $OUT $ROOT/gpurun_out/profiles_$TAG > /dev/null
python3 $ROOT/tools/kstats.py $OUT/trace 12
cat $ROOT/gpurun_out/profiles_$TAG/pmc_summary.json | head -60
